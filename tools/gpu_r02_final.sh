#!/bin/bash
# Round 2, last run on the MI355X at HEAD: the GPU tier, smoke, rocprofv3 kernel trace of the bench command, the GOM-level rate
# control variants, the screen-content timing, then the complete default bench line.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/final2; rm -rf $o; mkdir -p $o
timeout 70 python -m pytest tests -m gpu -q -n 4 > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
timeout 60 rocprofv3 --kernel-trace --stats -d $o/prof -- python bench.py --quick --steps 6 --warmup 2 > $o/prof_bench.json 2> $o/prof.err; echo "rocprof rc=$?"
db=$(find $o/prof -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > $o/kernel_stats.csv 2>$o/rocpd.err; head -5 $o/kernel_stats.csv | cut -c1-150
for g in 1 2; do WELS_HIP_GOM=$g timeout 40 python tools/config5_sessions.py 1 40 gom >> $o/gom_sessions.jsonl 2>> $o/gom.err; done
WELS_HIP_GOM=2 timeout 40 python tools/config5_sessions.py 8 40 gom >> $o/gom_sessions.jsonl 2>> $o/gom.err
WELS_HIP_GOM=1 timeout 60 python tools/config5_sessions.py 8 40 gom >> $o/gom_sessions.jsonl 2>> $o/gom.err
python - <<'PY'
import json
for l in open("gpurun_out/final2/gom_sessions.jsonl"):
    d=json.loads(l); print(d["config"][:22], d["config"].split("one slice")[1][:34], "| C", round(d["reference_c_path"]["sum_of_session_encode_fps"]), "| device", round(d["hooks_on_device"]["sum_of_session_encode_fps"]), d["same_bitstreams"])
PY
R=oracle/_ref; d=/tmp/scr; mkdir -p $d; $R/ref_dec $R/res/Adobe_PDF_sample_a_1024x768_50Frms.264 $d/adobe.yuv > /dev/null 2>&1
WELSHIP_LIB=openh264_amd/libwelship.so WELS_HIP_TRACE=2 timeout 60 $R/ref_enc_hip -i $d/adobe.yuv -w 1024 -h 768 -fps 30 -usage 1 -rc 1 -bitrate 2400000 -slcmd 1 -slcnum 4 -scene 1 -denoise 1 -frameskip 1 -o $d/hip.264 2> $d/err.txt | tail -1 > $d/out.txt
echo "screen 1024x768 four slices: $(cat $d/out.txt) | $(grep 'per picture' $d/err.txt | sed 's/welship hooks: //')" | tee $o/screen_timing.txt
( time timeout 170 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; tail -c 300 $o/bench_default.json; tail -3 $o/bench_default.time
