#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02am; mkdir -p $o
yuv=/tmp/c4.yuv; oracle/_ref/ref_dec oracle/_ref/res/VID_1920x1080_cavlc_temporal_direct.264 $yuv > /dev/null 2>&1
for n in 4 8; do
  echo "== $n sessions"
  WELSHIP_FRAME_STATS=1 WELSHIP_LIB=$PWD/openh264_amd/libwelship.so WELS_HIP_TRACE=2 oracle/_ref/ref_enc_hip -parallel $n -i $yuv -w 1920 -h 1080 -o /tmp/c4.264 -frames 54 -fps 30 -rc 1 -bitrate 1500000 -threads 1 -iper 0 -quiet -slcmd 1 -slcnum 4 -simulcast 240 135 -simulcast 480 270 -simulcast 960 540 2>&1 | grep -v "hooks: did\|launch\|installed" | tail -$((n+14)) | cut -c1-220
done 2>&1 | tee $o/trace.txt
