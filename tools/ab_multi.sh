# A/B timing over several workloads: WORKLOADS="a b c" tools/ab_multi.sh < settings.txt  (each line: ENV=... settings, "X=0" for none)
cd $GRAFT_REPO_ROOT
mapfile -t LINES
for W in ${WORKLOADS:-glass-of-water-1080p coffee-machine-1080p cornell-512}; do
  printf '%s\n' "${LINES[@]}" | WORKLOAD=$W STEPS=${STEPS:-2} bash tools/ab.sh || exit 1
done
