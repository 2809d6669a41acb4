# the launches of the connection stage of one batch, one by one (one batch in flight): bash tools/kt_conn.sh <workload>
R=$GRAFT_REPO_ROOT
W=$1
cd /tmp && export TMPDIR=/tmp
export BDPT_WF_STREAMS=1
rm -rf $R/gpurun_out/ktc_$W && mkdir -p $R/gpurun_out/ktc_$W
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktc_$W -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > $R/gpurun_out/ktc_$W/bench.json 2> $R/gpurun_out/ktc_$W/err.log
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/ktc_$W/*/*_kernel_trace.csv"))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_conn_plan" in r["Kernel_Name"]]
a=idx[1]; b=[i for i,r in enumerate(rows) if "k_resolve" in r["Kernel_Name"] and i>a][0]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b+1]:
    n=r["Kernel_Name"]; n=n[n.find("k_"):n.find("(")] if "k_" in n else n[:30]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  %-40s grid %s wg %s lds %s" % ((s-t0)/1e3,(e-s)/1e3,n[:40],r.get("Grid_Size_X",r.get("Grid_Size","?")),r.get("Workgroup_Size_X",r.get("Workgroup_Size","?")),r.get("LDS_Block_Size","?")))
PY
rm -f $R/gpurun_out/ktc_$W/*/*_kernel_trace.csv
