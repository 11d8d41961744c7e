#!/bin/bash
# End-of-round measurement run on the MI355X box: tests, smoke, bench lines, rocprofv3 kernel trace, PMC passes.
# Everything lands in gpurun_out/final/ (copy what should be judged into profiles/).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/final; mkdir -p $o
timeout 200 python -m pytest tests -m gpu -q > $o/pytest_gpu.txt 2>&1; tail -2 $o/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
timeout 300 python bench.py --e2e > $o/bench_default.json 2> $o/bench_default.err; tail -c 600 $o/bench_default.json
for s in 8 64; do timeout 120 python bench.py --sessions $s --no-cpu-baseline > $o/bench_s$s.json 2>/dev/null; done
timeout 120 python bench.py --workload intra --sessions 64 --steps 3 --warmup 1 --no-cpu-baseline > $o/bench_intra_s64.json 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats -d $o/prof -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $o/prof_bench.json 2> $o/prof.err; echo "rocprof rc=$?"
timeout 120 python tools/phase_profile.py 128 > $o/phase_p128.txt 2>&1
tools/pmc_passes.sh $o/pmc --steps 2 --warmup 1
# knob experiments (same workload as the bench default, no CPU baseline): recorded for the next round
x=$o/experiments.txt; : > $x
run() { echo "== $*" >> $x; ( env "$@" timeout 60 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" ) >> $x 2>&1; }
run WELSHIP_QUEUES=2 WELSHIP_P_WAVES=6
run WELSHIP_QUEUES=4 WELSHIP_P_WAVES=6
run WELSHIP_P_LOOKAHEAD=1
run WELSHIP_MB_BAND=6
run WELSHIP_MB_BAND=8
run WELSHIP_MB_BAND=12 WELSHIP_P_WAVES=12
echo "== --sessions 192" >> $x; timeout 60 python bench.py --no-cpu-baseline --sessions 192 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
echo "== --deblock-idc 2" >> $x; timeout 60 python bench.py --no-cpu-baseline --deblock-idc 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
# candidate deblocking edge filters: second library, parity on the device, then the same bench
cand=$(python -c "from openh264_amd import build as B; print(B.build_hip(verbose=False, defines=('WH_DB_PER_EDGE',), tag='wh_db_per_edge'))")
echo "== candidate $cand: parity" >> $x; timeout 600 python tools/fuzz_parity.py --lib $cand --cases 40 --seed 7 2>&1 | tail -1 >> $x
run WELSHIP_LIB=$cand
cat $x
ls $o
