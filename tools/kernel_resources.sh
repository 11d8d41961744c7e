#!/bin/bash
# registers / LDS / scratch of every kernel in a built library: tools/kernel_resources.sh [openh264_amd/libwelship.so]
lib=${1:-openh264_amd/libwelship.so}; d=$(mktemp -d); cp $lib $d/lib.so
( cd $d && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1
  for f in lib.so.*.hipv4-*; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null | grep -E "\.name:|vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill" | paste - - - - - - ; done ) |
  sed -e 's/_ZN12_GLOBAL__N_1[0-9]*//' -e 's/E11WhSeqParams.*\t/\t/' | awk '{print}' | sort -k4
rm -rf $d
