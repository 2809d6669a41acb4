# A/B timing of library builds: each line of stdin is "ENV=... [ENV=...]"; WORKLOAD selects the bench workload
cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-glass-of-water-1080p}
STEPS=${STEPS:-2}
run() { echo -n "$W $* : "; env "$@" timeout -k 10 150 python -u bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-pmc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Msamples/s  %.1f ms'%(d['value'],d['ms_per_step']))"; }
while read -r line; do [ -n "$line" ] && run $line; done
exit 0
