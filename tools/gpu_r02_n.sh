#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02n; mkdir -p $o
WELSHIP_PROF_GROUPS=1 timeout 300 python tools/phase_profile.py 128 > $o/phase_syn_p128.txt 2>&1; head -3 $o/phase_syn_p128.txt; grep -i "workgroup end" $o/phase_syn_p128.txt
WELSHIP_PROF_GROUPS=1 WELSHIP_MD_ASSIGN=0 timeout 300 python tools/phase_profile.py 128 > $o/phase_syn_p128_noassign.txt 2>&1; head -3 $o/phase_syn_p128_noassign.txt; grep -i "workgroup end" $o/phase_syn_p128_noassign.txt
WELSHIP_PROF_GROUPS=1 timeout 300 python tools/phase_profile.py 128 res > $o/phase_res_p128.txt 2>&1; head -3 $o/phase_res_p128.txt; grep -i "workgroup end" $o/phase_res_p128.txt
WELSHIP_PROF_GROUPS=1 timeout 300 python tools/phase_profile.py 256 > $o/phase_syn_p256.txt 2>&1; head -3 $o/phase_syn_p256.txt; grep -i "workgroup end" $o/phase_syn_p256.txt
