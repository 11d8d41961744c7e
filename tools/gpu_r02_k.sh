#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02k; mkdir -p $o
timeout 900 python -m pytest tests/test_downsample_gpu.py tests/test_hooks_simulcast.py -m gpu -q > $o/pytest_ds_simulcast.txt 2>&1; tail -4 $o/pytest_ds_simulcast.txt
timeout 300 python tools/downsample_bench.py > $o/downsample_bench.jsonl 2>&1; cat $o/downsample_bench.jsonl
