# usage: bash tools/pmc.sh <tag> "<counters>" <bench args...>  -> gpurun_out/pmc_<tag>/
set -e
R=$GRAFT_REPO_ROOT
tag=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$tag
timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py "$@" --no-cpu-baseline --no-roofline --no-extras --no-pmc > /dev/null 2> $R/gpurun_out/pmc_$tag/err.log
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob('$R/gpurun_out/pmc_$tag/*/*_counter_collection.csv'))[-1]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']; n=n[n.find('k_'):n.find('(bdpt')] if 'k_' in n else n[:30]
    acc[n][r['Counter_Name']]+=float(r['Counter_Value'])
for n,d in acc.items():
    print(n, {k:'%.4g'%v for k,v in d.items()})
PY
