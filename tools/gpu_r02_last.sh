#!/bin/bash
# Round 2, last minutes of the GPU budget: GOM-level rate control inside the kernel with 32 concurrent sessions; FETCH_SIZE / WRITE_SIZE
# of the bench command at HEAD (separate --pmc passes, kernel trace only).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/last; rm -rf $o; mkdir -p $o
WELS_HIP_GOM=2 timeout 60 python tools/config5_sessions.py 32 40 gom > $o/gom32.jsonl 2> $o/gom.err; cut -c1-500 $o/gom32.jsonl
out=$o/pmc; i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 60 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -- python bench.py --quick --steps 2 --warmup 1 > $out.pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python tools/pmc_summary.py $out > $o/pmc_summary.txt 2>&1; grep -i "inter_pool" $o/pmc_summary.txt | head -4
