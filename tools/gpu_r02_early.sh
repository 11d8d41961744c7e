#!/bin/bash
# Round 2, the last GPU call: A/B of claiming + fetching a wave's next macroblock as soon as the prediction of the one in hand is final
# (WH_EARLY_CLAIM, hip_backend.hip / inter_mb.h) -- the same bench command on the two builds, then the GPU tier on the default build.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/early; rm -rf $o; mkdir -p $o
WELSHIP_LIB=$PWD/openh264_amd/libwelship_noearly.so timeout 40 python bench.py --quick --steps 60 > $o/bench_noearly.json 2> $o/bench_noearly.err
echo "noearly: $(cut -c1-330 $o/bench_noearly.json)"
timeout 50 python bench.py --no-cpu-baseline --no-extra --steps 60 > $o/bench_early.json 2> $o/bench_early.err
echo "early:   $(cut -c1-330 $o/bench_early.json)"; grep -o '"verified": [a-z]*' $o/bench_early.json
timeout 100 python -m pytest tests -m gpu -q -n 8 > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest_gpu.txt
timeout 30 python tools/phase_profile.py 256 > $o/phase_cycles.txt 2>&1; head -22 $o/phase_cycles.txt | cut -c1-150
