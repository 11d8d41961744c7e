#!/bin/bash
# End-of-round evidence run on the MI355X box (round 2): tests, smoke, bench lines, rocprofv3 kernel trace, PMC passes (each in
# its own run, kernel-trace only), phase profiles, on-device fuzz, config-5 share.  Everything lands in gpurun_out/final/;
# what should be judged is copied into profiles/ afterwards (see profiles/README.md).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/${OUT:-final}; rm -rf $o; mkdir -p $o
[ -n "$SKIP_SLOW" ] || timeout 900 python -m pytest tests -m gpu -q > $o/pytest_gpu.txt 2>&1; tail -2 $o/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
( time timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; tail -c 400 $o/bench_default.json; cat $o/bench_default.time | tail -3
for s in 8 64 128; do timeout 200 python bench.py --quick --sessions $s --steps 40 > $o/bench_s$s.json 2>/dev/null; done
timeout 300 rocprofv3 --kernel-trace --stats -d $o/prof -- python bench.py --quick --steps 6 --warmup 2 > $o/prof_bench.json 2> $o/prof.err; echo "rocprof rc=$?"
db=$(find $o/prof -name "*.db" | head -1); python tools/rocpd_stats.py "$db" > $o/kernel_stats.csv 2>$o/rocpd.err; head -6 $o/kernel_stats.csv | cut -c1-160
out=$o/pmc; i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -- python bench.py --quick --steps 2 --warmup 1 > $out.pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python tools/pmc_summary.py $out > $o/pmc_summary.txt 2>&1
WH_O=$o python - <<'PY'
import json, re
t = open(__import__("os").environ.get("WH_O", "gpurun_out/final") + "/pmc_summary.txt").read()
def per(k, c):
    m = re.search(r"%s\S*\s+%s\s+total\s+\S+\s+dispatches\s+\d+\s+per_dispatch\s+(\S+)" % (k, c), t)
    return float(m.group(1)) if m else None
f, w = per("k_inter_pool", "FETCH_SIZE"), per("k_inter_pool", "WRITE_SIZE")
if f and w:
    json.dump({"workload": "p", "sessions": 256, "width": 1920, "height": 1080, "kernel": "k_inter_pool<768>",
               "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --quick --steps 2 --warmup 1",
               "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
               "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
               "hbm_bytes_per_launch": (2 * f + w) * 1024.0, "algorithmic_bytes_per_launch": 2144 * 8160 * 256}, open(__import__("os").environ.get("WH_O", "gpurun_out/final") + "/pmc_traffic.json", "w"), indent=1)
    print("traffic", (2 * f + w) * 1024.0 / 1e9, "GB per launch")
PY
timeout 300 python tools/phase_profile.py 256 > $o/phase_syn_p256.txt 2>&1; grep "total cycles" $o/phase_syn_p256.txt
timeout 300 python tools/phase_profile.py 256 res > $o/phase_res_p256.txt 2>&1; grep "total cycles" $o/phase_res_p256.txt
[ -n "$SKIP_SLOW" ] || timeout 900 python tools/fuzz_parity.py --lib openh264_amd/libwelship.so --cases 500 --seed 11 > $o/fuzz_500.txt 2>&1; tail -1 $o/fuzz_500.txt
timeout 600 python tools/config5_sessions.py 8 120 > $o/config5_8sessions.json 2>$o/config5.err; cat $o/config5_8sessions.json
yuv=/tmp/c5.yuv; oracle/_ref/ref_dec oracle/_ref/res/VID_1280x720_cavlc_temporal_direct.264 $yuv > /dev/null 2>&1
WELSHIP_LIB=openh264_amd/libwelship.so WELS_HIP_TRACE=2 oracle/_ref/ref_enc_hip -i $yuv -w 1280 -h 720 -o /tmp/c5.264 -frames 120 -fps 30 -rc 1 -bitrate 1500000 -slcmd 2 -slcmbnum 900 -threads 1 -iper 0 -quiet 2>&1 | grep -v "welship hooks: did" | tail -3 > $o/config5_one_session_timing.txt; cat $o/config5_one_session_timing.txt
ls $o
