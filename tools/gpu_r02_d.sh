#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02d; mkdir -p $o
timeout 900 python -m pytest tests/test_reference_content.py -m gpu -q -x > $o/pytest_refcontent.txt 2>&1; tail -3 $o/pytest_refcontent.txt
timeout 120 python tools/phase_profile.py 128 > $o/phase_p128.txt 2>&1; cat $o/phase_p128.txt
