# rocprofv3 passes over the default bench (wavefront pipeline): kernel trace + stats, then one PMC pass per counter
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_wf
cd /tmp && export TMPDIR=/tmp
rm -rf $O && mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > $O/kt_bench.json 2> $O/kt.err
echo kt done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err
echo fetch done
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > $O/pmc_write_bench.json 2> $O/pmc_write.err
echo write done
python3 - <<PY
import csv, glob, collections, json
def short(n):
    return n[n.find('k_'):n.find('(bdpt')] if 'k_' in n and '(bdpt' in n else n[:40]
out = {}
for tag in ('fetch', 'write'):
    f = sorted(glob.glob('$O/pmc_%s/*/*_counter_collection.csv' % tag))[-1]
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        acc[short(r['Kernel_Name'])] += float(r['Counter_Value']); cnt[short(r['Kernel_Name'])] += 1
    out[tag] = {k: {'sum': v, 'dispatches': cnt[k]} for k, v in acc.items()}
json.dump(out, open('$O/pmc_by_kernel.json', 'w'), indent=1)
# the kernel trace itself is large: keep the stats only
import os
for f in glob.glob('$O/kt/*/*_kernel_trace.csv'): os.remove(f)
for t in ('fetch', 'write'):
    for f in glob.glob('$O/pmc_%s/*/*_counter_collection.csv' % t): os.remove(f)
PY
ls -R $O | head -40
