# kernel statistics of one frame of a workload with one batch in flight: bash tools/kt_one.sh <workload>  -> gpurun_out/kt1_<workload>/
R=$GRAFT_REPO_ROOT
W=$1
cd /tmp && export TMPDIR=/tmp
export BDPT_WF_STREAMS=1
rm -rf $R/gpurun_out/kt1_$W && mkdir -p $R/gpurun_out/kt1_$W
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt1_$W -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > $R/gpurun_out/kt1_$W/bench.json 2> $R/gpurun_out/kt1_$W/err.log
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/kt1_$W/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $W  sum %.1f ms" % (tot/1e6))
for r in rows[:10]:
    n=r["Name"]; n=n[n.find("k_"):n.find("(bdpt")] if "k_" in n else n[:30]
    print("  %-34s calls %5s total %8.1f ms avg %8.1f us"%(n,r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3))
PY
rm -f $R/gpurun_out/kt1_$W/*/*_kernel_trace.csv
