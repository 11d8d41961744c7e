#!/bin/bash
# Round 2: screen-content pictures with a scroll vector in the chained 2:1 order (WH_SEQ_CHAIN) instead of plain coding order.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/chain; rm -rf $o; mkdir -p $o
timeout 200 python -m pytest tests/test_hooks_screen.py tests/test_hooks_cabac_threads.py -m gpu -q -x -n 4 > $o/pytest_hooks_gpu.txt 2>&1; tail -2 $o/pytest_hooks_gpu.txt
R=oracle/_ref
d=/tmp/scr; mkdir -p $d; $R/ref_dec $R/res/Adobe_PDF_sample_a_1024x768_50Frms.264 $d/adobe.yuv > /dev/null 2>&1
for serial in 0 1; do
  WELSHIP_SCC_SERIAL=$serial WELSHIP_LIB=openh264_amd/libwelship.so WELS_HIP_TRACE=2 timeout 120 $R/ref_enc_hip -i $d/adobe.yuv -w 1024 -h 768 -fps 30 -usage 1 -rc 1 -bitrate 2400000 -slcmd 1 -slcnum 4 -scene 1 -denoise 1 -frameskip 1 -o $d/hip$serial.264 2> $d/err.txt | tail -1 > $d/out.txt
  echo "screen 1024x768 four slices, WELSHIP_SCC_SERIAL=$serial: $(cat $d/out.txt) | $(grep 'per picture' $d/err.txt | sed 's/welship hooks: //')" | tee -a $o/screen_timing.txt
done
cmp $d/hip0.264 $d/hip1.264 && echo "  same bytes" | tee -a $o/screen_timing.txt
for n in 1 16 32; do WELSHIP_FRAME_STATS=1 timeout 120 python tools/config5_sessions.py $n 50 screen >> $o/screen_sessions.jsonl 2> $o/frame_stats_screen$n.txt; tail -1 $o/screen_sessions.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('screen', d['config'][:24], 'C', round(d['reference_c_path']['sum_of_session_encode_fps']), 'device', round(d['hooks_on_device']['sum_of_session_encode_fps']), d['same_bitstreams'])"; done
grep "welship:" $o/frame_stats_screen16.txt | head -12
