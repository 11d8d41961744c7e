#!/bin/bash
# Round 2: concurrent sessions through the binding (frame API: two launch sets per key in flight, page-locked small transfers).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/lanes; rm -rf $o; mkdir -p $o
for n in 16 32; do WELSHIP_FRAME_STATS=1 timeout 120 python tools/config5_sessions.py $n 50 screen >> $o/screen_sessions.jsonl 2> $o/frame_stats_screen$n.txt; tail -1 $o/screen_sessions.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('screen', d['config'][:24], 'C', round(d['reference_c_path']['sum_of_session_encode_fps']), 'device', round(d['hooks_on_device']['sum_of_session_encode_fps']), d['same_bitstreams'])"; done
grep "welship:" $o/frame_stats_screen16.txt | head -14
for n in 8 32; do timeout 120 python tools/config5_sessions.py $n 90 >> $o/config5_sessions.jsonl 2>> $o/err.txt; tail -1 $o/config5_sessions.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('720p', d['config'][:24], 'C', round(d['reference_c_path']['sum_of_session_encode_fps']), 'device', round(d['hooks_on_device']['sum_of_session_encode_fps']), d['same_bitstreams'])"; done
