#!/bin/bash
# Developer: SQ / TCC counters of the tiled vs streaming variant on the HBM-bound shapes.
OUT=/root/repo/gpurun_out/pmc_cmp
rm -rf $OUT; mkdir -p $OUT; cd /root/repo; export TMPDIR=/tmp
for v in 0 1; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq_v$v -- python3 tools/kbench.py --shape a1one,hd5,big --reps 2 --opts "prefer_stream=$v" > $OUT/k_sq_v$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2_v$v -- python3 tools/kbench.py --shape a1one,hd5,big --reps 2 --opts "prefer_stream=$v" > $OUT/k_sq2_v$v.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_v$v -- python3 tools/kbench.py --shape a1one,hd5,big --reps 2 --opts "prefer_stream=$v" > $OUT/k_f_v$v.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc_v$v -- python3 tools/kbench.py --shape a1one,hd5,big --reps 2 --opts "prefer_stream=$v" > $OUT/k_tcc_v$v.log 2>&1
done
echo done; ls $OUT
