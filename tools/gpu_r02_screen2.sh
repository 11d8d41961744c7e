#!/bin/bash
# Round 2, second screen-content run on the MI355X: the GPU tier at HEAD (slice threads, TRY_REENCODING in the binding), concurrent
# screen-content sessions in one process (frame API batching), the all-IDR 720p leg with two workgroups per CU.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/screen2; rm -rf $o; mkdir -p $o
timeout 240 python -m pytest tests -m gpu -q -x -n 4 > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
for n in 1 4 16; do timeout 120 python tools/config5_sessions.py $n 50 screen >> $o/screen_sessions.jsonl 2>> $o/screen_sessions.err; tail -1 $o/screen_sessions.jsonl | cut -c1-600; done
for s in 256 512; do timeout 150 python bench.py --quick --workload intra --width 1280 --height 720 --sessions $s --steps 10 --warmup 2 > $o/bench_intra720_s$s.json 2> $o/bench_intra_s$s.err; cut -c1-300 $o/bench_intra720_s$s.json; done
