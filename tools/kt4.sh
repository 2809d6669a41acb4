set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_four && mkdir -p $R/gpurun_out/kt_four
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_four -- python3 $R/bench.py --workload glass-of-water-1080p-16spp --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-pmc > $R/gpurun_out/kt_four/bench.json 2> $R/gpurun_out/kt_four/err.log
