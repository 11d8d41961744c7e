#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02f; mkdir -p $o
timeout 200 python tools/phase_profile.py 128 res > $o/phase_res_p128.txt 2>&1; cat $o/phase_res_p128.txt
timeout 200 python tools/phase_profile.py 128 > $o/phase_syn_p128.txt 2>&1; head -20 $o/phase_syn_p128.txt
