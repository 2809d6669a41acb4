# per-launch durations of the traversal and shading kernels of ONE batch, in launch order: bash tools/kt_seq.sh <workload>
R=$GRAFT_REPO_ROOT
W=$1
cd /tmp && export TMPDIR=/tmp
export BDPT_WF_STREAMS=1
rm -rf $R/gpurun_out/ktseq_$W && mkdir -p $R/gpurun_out/ktseq_$W
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktseq_$W -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > /dev/null 2> $R/gpurun_out/ktseq_$W/err.log
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob("$R/gpurun_out/ktseq_$W/*/*_kernel_trace.csv"))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# per launch index within a batch (a batch starts at k_gen_eye), summed over the batches of the frame
acc=collections.OrderedDict()
pos=collections.Counter()
for r in rows:
    n=r["Kernel_Name"]
    if "k_" not in n: continue
    n=n[n.find("k_"):n.find("(bdpt")]
    if n.startswith("k_gen_eye"): pos=collections.Counter()
    key=(n,pos[n]); pos[n]+=1
    acc[key]=acc.get(key,0.0)+(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for (n,k),v in acc.items():
    if v>=0.5: print("%-46s launch %3d of its batch: %8.1f ms over the frame" % (n,k,v))
PY
rm -rf $R/gpurun_out/ktseq_$W/*/
