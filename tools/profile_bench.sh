set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > $R/gpurun_out/prof/kt_bench.json 2> $R/gpurun_out/prof/kt.err
echo kt done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > /dev/null 2> $R/gpurun_out/prof/pmc_fetch.err
echo fetch done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > /dev/null 2> $R/gpurun_out/prof/pmc_write.err
echo write done
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/prof/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > /dev/null 2> $R/gpurun_out/prof/pmc_sq.err
echo sq done
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $R/gpurun_out/prof/pmc_cache -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extras --no-pmc > /dev/null 2> $R/gpurun_out/prof/pmc_cache.err
echo cache done
find $R/gpurun_out/prof -name "*.csv" | head -30
