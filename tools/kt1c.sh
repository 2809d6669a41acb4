set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_one && mkdir -p $R/gpurun_out/kt_one
BDPT_WF_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_one -- python3 $R/bench.py --workload cornell-512 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > $R/gpurun_out/kt_one/bench.json 2> $R/gpurun_out/kt_one/err.log
