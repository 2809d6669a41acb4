# Evidence of one round: bash tools/profile_round.sh <tag> [lines|stats|checks]  ->  gpurun_out/<tag>/ (copy what is to be kept into profiles/<tag>/)
#  lines:  bench_line_default.json     the default bench.py run (all legs: roofline with its counter passes, extras, CPU baseline)
#          bench_line_<workload>.json  bench lines of the other workloads, each with its own counter passes (roofline.traffic)
#          pmc_by_kernel_<workload>.json  per-kernel counter sums of the counter passes each of those runs made
#  stats:  kernel_stats_bench_default.csv   rocprofv3 --kernel-trace --stats of the default workload (4 batches in flight)
#          kernel_stats_one_<workload>.csv  the same with one batch in flight (kernels back to back), per workload
#  checks: tie_check_full.json (tools/tie_check.py --full with the cross-check build), composite_speed.txt (tools/cx_speed.py)
R=$GRAFT_REPO_ROOT
T=$1
PART=${2:-all}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
if [ "$PART" = all ] || [ "$PART" = lines ]; then
  timeout -k 10 900 python3 bench.py --pmc-json $O/pmc_by_kernel_glass-of-water-1080p.json > $O/bench_line_default.json 2> $O/bench_default.err || exit 1
  echo default done
  for W in cornell-512 coffee-machine-1080p sphere-lattice-1m-1024 sphere-lattice-10m-4096 sphere-lattice-instanced-10m-4096; do
    timeout -k 10 600 python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pmc-json $O/pmc_by_kernel_$W.json > $O/bench_line_$W.json 2> $O/bench_$W.err || exit 1
    echo $W done
  done
fi
if [ "$PART" = all ] || [ "$PART" = stats ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > $O/bench_line_under_rocprof.json 2> $O/kt.err || exit 1
  cp $O/kt/*/*_kernel_stats.csv $O/kernel_stats_bench_default.csv
  rm -rf $O/kt
  echo kt done
  export BDPT_WF_STREAMS=1
  for W in glass-of-water-1080p cornell-512 coffee-machine-1080p sphere-lattice-1m-1024 sphere-lattice-10m-4096 sphere-lattice-instanced-10m-4096; do
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > /dev/null 2> $O/kt1.err || exit 1
    cp $O/kt1/*/*_kernel_stats.csv $O/kernel_stats_one_$W.csv
    rm -rf $O/kt1
  done
fi
if [ "$PART" = all ] || [ "$PART" = checks ]; then
  cd $R
  unset BDPT_WF_STREAMS
  # the tie cross-check at the larger sizes, and the small composite scenes beside the plain Cornell box
  BDPT_LIB=$R/bidirectional-path-tracer_amd/libbdpt_amd_tiecheck.so timeout -k 10 600 python3 tools/tie_check.py --full > $O/tie_check_full.json 2> $O/tie_check.err || exit 1
  timeout -k 10 300 python3 tools/cx_speed.py 512 64 > $O/composite_speed.txt 2> $O/cx_speed.err || exit 1
fi
echo all done
