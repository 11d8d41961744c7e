#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02ac; mkdir -p $o
yuv=/tmp/c5.yuv; oracle/_ref/ref_dec oracle/_ref/res/VID_1280x720_cavlc_temporal_direct.264 $yuv > /dev/null 2>&1
for mode in "-slcmd 0" "-slcmd 1 -slcnum 4"; do
  for hip in 0 1; do
    echo "== $mode WELS_HIP=$hip"
    WELS_HIP=$hip WELSHIP_LIB=$PWD/openh264_amd/libwelship.so WELS_HIP_TRACE=2 oracle/_ref/ref_enc_hip -i $yuv -w 1280 -h 720 -o /tmp/g$hip.264 -frames 40 -fps 30 -rc 1 -bitrate 1500000 $mode -threads 1 -iper 0 -quiet 2>&1 | grep -v "hooks: did" | tail -2
  done
  cmp /tmp/g0.264 /tmp/g1.264 && echo same
done 2>&1 | tee $o/gom_speed.txt
