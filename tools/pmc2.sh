# usage: bash tools/pmc2.sh  (SQ wait + TA/TCP passes on the staged single-stream profiling workload)
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh sqw "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" --workload glass-of-water-1080p-16spp --steps 1 --warmup 0 | grep k_traverse
bash tools/pmc.sh ta "TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" --workload glass-of-water-1080p-16spp --steps 1 --warmup 0 | grep k_traverse
