# tools/r3_merge_probe.sh -- one GPU call: merge parity tests, phase profile, occupancy sweep, counters
o=gpurun_out/r3g; mkdir -p $o
timeout 600 python -m pytest tests/test_seed_merge_gpu.py -x -q > $o/t1.log 2>&1
(FGA_LIBRARY=$PWD/fastga_amd/variants/lib_prof.so python tools/merge_bench.py --reps 3 --check) > $o/prof.log 2>&1
python tools/merge_bench.py --reps 4 --check > $o/plain.log 2>&1
for w in 8 10 12; do FGA_MERGE_WAVES=$w python tools/merge_bench.py --reps 3 2>&1 | tail -1 | sed "s/^/waves $w: /"; done > $o/waves.log 2>&1
(cd /tmp; rocprofv3 -L > /root/repo/$o/counters.txt 2>&1)
bash tools/merge_pmc.sh a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES > $o/pmc_a.log 2>&1
bash tools/merge_pmc.sh b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA > $o/pmc_b.log 2>&1
bash tools/merge_pmc.sh c FETCH_SIZE > $o/pmc_c.log 2>&1
bash tools/merge_pmc.sh d WRITE_SIZE > $o/pmc_d.log 2>&1
bash tools/merge_pmc.sh e SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT > $o/pmc_e.log 2>&1
bash tools/merge_pmc.sh f GRBM_GUI_ACTIVE GRBM_COUNT > $o/pmc_f.log 2>&1
bash tools/merge_pmc.sh g TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum > $o/pmc_g.log 2>&1
bash tools/merge_pmc.sh h TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum > $o/pmc_h.log 2>&1
python tools/merge_bench.py --reps 3 --self --mbp 150 > $o/self150.log 2>&1
tail -n 14 $o/*.log
