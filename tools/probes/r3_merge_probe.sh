# tools/r3_merge_probe.sh -- one GPU call: what stalls the merge kernel (TLB, instruction cache, texture path)
o=gpurun_out/r3n; mkdir -p $o
bash tools/merge_pmc.sh p1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum > $o/p1.log 2>&1
bash tools/merge_pmc.sh p2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL > $o/p2.log 2>&1
bash tools/merge_pmc.sh p3 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum > $o/p3.log 2>&1
bash tools/merge_pmc.sh p4 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT > $o/p4.log 2>&1
bash tools/merge_pmc.sh p5 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES > $o/p5.log 2>&1
grep -h "seed_merge\|rror" $o/p*.log
