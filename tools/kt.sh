# usage: bash tools/kt.sh <tag> <bench args...>   -> gpurun_out/kt_<tag>/  (rocprofv3 kernel trace + stats)
set -e
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/kt_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$tag -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extras --no-pmc > $R/gpurun_out/kt_$tag/bench.json 2> $R/gpurun_out/kt_$tag/err.log
cat $R/gpurun_out/kt_$tag/*/*_kernel_stats.csv | cut -c1-200
