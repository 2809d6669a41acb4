# usage: bash tools/pmc_issue.sh <workload> [steps]  -> per-kernel SQ issue counters and TCP/TA counters, one batch in flight
cd $GRAFT_REPO_ROOT
W=$1
export BDPT_WF_STREAMS=1
bash tools/pmc.sh issue_$W "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" --workload $W --steps 1 --warmup 0 || exit 1
bash tools/pmc.sh mem_$W "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" --workload $W --steps 1 --warmup 0 || exit 1
bash tools/pmc.sh fetch_$W "FETCH_SIZE GRBM_GUI_ACTIVE" --workload $W --steps 1 --warmup 0 || exit 1
bash tools/pmc.sh write_$W "WRITE_SIZE" --workload $W --steps 1 --warmup 0 || exit 1
