#!/bin/bash
# PMC passes for the hot path (each in its own run, kernel-trace only; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc_passes.sh <outdir> <bench args...>
out=$1; shift
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -- python bench.py "$@" --no-cpu-baseline > $out.pass$i.log 2>&1
  echo "pass $i rc=$?"
done
