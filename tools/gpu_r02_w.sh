#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02w; mkdir -p $o
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $o/pass$i -- python bench.py --quick --steps 2 --warmup 1 > $o/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py $o > $o/pmc_lds.txt 2>&1; grep -E "k_inter_pool|k_deblock" $o/pmc_lds.txt
