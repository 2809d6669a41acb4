# kernel statistics of the small composite scenes with one batch in flight: bash tools/kt_cx.sh <scene name part>  (cornell_textures, cornell_smoke, cornell_instances)
R=$GRAFT_REPO_ROOT
W=$1
cd /tmp && export TMPDIR=/tmp
export BDPT_WF_STREAMS=1
rm -rf $R/gpurun_out/ktcx_$W && mkdir -p $R/gpurun_out/ktcx_$W
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktcx_$W -- python3 $R/tools/cx_speed.py 512 64 $W > $R/gpurun_out/ktcx_$W/out.txt 2> $R/gpurun_out/ktcx_$W/err.log
cat $R/gpurun_out/ktcx_$W/out.txt
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/ktcx_$W/*/*_kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $W  sum %.1f ms (two frames)" % (tot/1e6))
for r in rows[:9]:
    n=r["Name"]; n=n[n.find("k_"):n.find("(bdpt")] if "k_" in n else n[:30]
    print("  %-44s calls %5s total %8.1f ms avg %8.1f us"%(n,r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3))
PY
rm -f $R/gpurun_out/ktcx_$W/*/*_kernel_trace.csv
