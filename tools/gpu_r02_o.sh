#!/bin/bash
# round-2 evidence run: rocprofv3 kernel stats of the bench command, PMC passes (each its own run, kernel-trace only), phase profile
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02o; mkdir -p $o
(rocprofv3 -L 2>/dev/null || rocprofv3-avail list 2>/dev/null) | grep -oE "\b(SQC?_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+)\b" | sort -u > $o/counters_avail.txt; wc -l $o/counters_avail.txt; grep -i "icache\|ifetch" $o/counters_avail.txt | tr '\n' ' '
timeout 300 rocprofv3 --kernel-trace --stats -d $o/prof -- python bench.py --quick --steps 6 --warmup 2 > $o/prof_bench.json 2> $o/prof.err; echo "rocprof rc=$?"
db=$(find $o/prof -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > $o/kernel_stats.csv 2>$o/rocpd.err; find $o/prof -name "*stats*.csv" | head -3; head -12 $o/kernel_stats.csv
out=$o/pmc
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_IFETCH"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -- python bench.py --quick --steps 2 --warmup 1 > $out.pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py $out > $o/pmc_summary.txt 2>&1; grep -E "k_inter_pool|k_deblock" $o/pmc_summary.txt | head -60
