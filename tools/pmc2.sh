# SQ wait + TA/TCP passes on the single-stream profiling workload (traversal kernels only)
cd $GRAFT_REPO_ROOT
bash tools/pmc.sh sqw "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" --workload glass-of-water-1080p-16spp --steps 1 --warmup 0 | grep -E "k_traverse|k_shade"
bash tools/pmc.sh ta "TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" --workload glass-of-water-1080p-16spp --steps 1 --warmup 0 | grep -E "k_traverse|k_shade"
