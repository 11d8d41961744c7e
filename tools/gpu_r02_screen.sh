#!/bin/bash
# Round 2, screen content + CABAC on the MI355X: the GPU tier, then where a screen-content picture's time goes (one row of the
# Adobe table through the hooks with WELS_HIP_TRACE=2, against the reference's C path on one host core).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/screen; rm -rf $o; mkdir -p $o
timeout 280 python -m pytest tests -m gpu -q -x -n 4 > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
R=oracle/_ref
d=/tmp/scr; mkdir -p $d; $R/ref_dec $R/res/Adobe_PDF_sample_a_1024x768_50Frms.264 $d/adobe.yuv > /dev/null 2>&1
for mode in "-slcmd 1 -slcnum 4" "-slcmd 0"; do
  for lib in hip c; do
    if [ $lib = c ]; then export WELS_HIP=0; else unset WELS_HIP; fi
    WELSHIP_LIB=openh264_amd/libwelship.so WELS_HIP_GOM=1 WELS_HIP_TRACE=2 timeout 120 $R/ref_enc_hip -i $d/adobe.yuv -w 1024 -h 768 -fps 30 -usage 1 -rc 1 -bitrate 2400000 $mode -scene 1 -denoise 1 -frameskip 1 -o $d/$lib.264 2> $d/err.txt | tail -1 > $d/out.txt
    echo "screen 1024x768 $mode [$lib]: $(cat $d/out.txt) | $(grep 'per picture' $d/err.txt | sed 's/welship hooks: //')" | tee -a $o/screen_timing.txt
  done
  cmp $d/hip.264 $d/c.264 && echo "  same bytes" | tee -a $o/screen_timing.txt
done
