cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python -u bench.py --workload glass-of-water-1080p --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Msamples/s'%d['value'])"; }
run X=1
run BDPT_WF_BLOCKS_PER_CU=4
run BDPT_WF_STREAMS=2
