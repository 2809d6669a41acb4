# kernel statistics (one batch in flight) of a workload for the default library and for another build: bash tools/kt_ab.sh <workload> <other.so>
R=$GRAFT_REPO_ROOT
W=$1; OTHER=$2
cd /tmp && export TMPDIR=/tmp
export BDPT_WF_STREAMS=1
for tag in new other; do
  rm -rf $R/gpurun_out/ktab_$tag && mkdir -p $R/gpurun_out/ktab_$tag
  if [ $tag = other ]; then export BDPT_LIB=$R/$OTHER; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktab_$tag -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > $R/gpurun_out/ktab_$tag/bench.json 2> $R/gpurun_out/ktab_$tag/err.log
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/ktab_$tag/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $tag  sum %.1f ms" % (tot/1e6))
for r in rows[:9]:
    n=r["Name"]; n=n[n.find("k_"):n.find("(bdpt")] if "k_" in n else n[:30]
    print("  %-34s calls %5s total %8.1f ms avg %8.1f us"%(n,r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3))
PY
  rm -f $R/gpurun_out/ktab_$tag/*/*_kernel_trace.csv
done
