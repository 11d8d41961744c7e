#!/bin/bash
# static ISA statistics of one kernel in a hipcc -S listing: tools/isa_stats.sh <file.s> <kernel-substring>
f=$1; k=$2
start=$(grep -n "^_Z.*$k.*:" $f | head -1 | cut -d: -f1)
end=$(awk -v s=$start 'NR>s && /s_endpgm/{print NR; exit}' $f)
sed -n "${start},${end}p" $f > /tmp/_k.s
for p in "s_barrier" "s_waitcnt vmcnt(0)" "s_waitcnt" "scratch_" "flat_load" "flat_store" "global_load" "global_store" "ds_" "s_load" "s_cbranch" "v_readfirstlane" "v_readlane" "_dpp" "ds_bpermute"; do printf "%-22s %s\n" "$p" $(grep -c -- "$p" /tmp/_k.s); done
echo "instructions $(grep -v '^\s*;' /tmp/_k.s | grep -c '^\s[a-z]')"
grep -A12 "amdhsa_kernel.*$k" $f | grep -E "private_segment_fixed_size|group_segment" 
grep "$k.*num_vgpr\|$k.*numbered_sgpr" $f | head -2
