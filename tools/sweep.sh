cd $GRAFT_REPO_ROOT
W=${WORKLOAD:-glass-of-water-1080p}
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python -u bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --no-pmc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f Msamples/s'%d['value'])"; }
while read -r line; do [ -n "$line" ] && run $line; done
exit 0
