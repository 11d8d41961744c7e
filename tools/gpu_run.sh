#!/bin/bash
# The one script that runs on the GPU box:   gpurun -- 'bash tools/gpu_run.sh <tag> <stage> [<stage> ...]'
# Every stage writes under gpurun_out/<tag>/ and prints a short summary; what is worth keeping is copied into profiles/ by hand.
#
# stages
#   tier            the GPU tier exactly as the driver runs it (serial, -x) -> pytest_gpu.txt
#   tier_all        the same without -x (every failure of the tier in one run)
#   bench           the default bench line -> bench_default.json (+ a digest)
#   quick           bench.py --quick (the timed configuration only) -> bench_quick.json
#   stats           rocprofv3 --kernel-trace --stats of `bench.py --quick --steps 20 --warmup 5` (the driver's run shape) -> kernel_stats.csv
#   pmc             FETCH_SIZE / WRITE_SIZE passes of `bench.py --quick` (separate runs, --kernel-trace only beside --pmc) -> pmc_traffic.txt
#   counters:<a>+<b>,<c>   one rocprofv3 --pmc pass per comma-separated group of `bench.py --quick` (counters of a group joined by +) -> counters.txt;
#                   "counters:list" writes the counter names the box offers (rocprofv3 --list-avail) -> counters_avail.txt
#   phase           in-kernel phase cycles of the P macroblock body (tools/phase_profile.py 256) -> phase_cycles.txt
#   iwaves          IDR step of 256 four-slice 1080p pictures with 16 / 12 / 10 / 8 / 6 waves per intra workgroup (WELSHIP_I_WAVES) -> intra_waves.txt
#   c5trace         config 5's shape (1 and 8 sessions 1080p, rate control, raster slices) through the binding with WELS_HIP_TRACE=2 -> config5_trace.txt
#   c5ab:<v>,<v>..  the same, device frames/s against the C path, per variant (library tag[:ENV=VALUE...]) -> config5_ab.txt
#   c4ab:<v>,<v>..  config 4's shape (four simulcast layers, 1 and 8 sessions) per variant -> config4_ab.txt
#   trace1          kernel + copy timeline of ONE 1080p session through the dispatch-table binding (config 5's shape) -> trace1_timeline.txt
#   iphase          the same for the I macroblock body (the IDR step of 256 pictures) -> phase_cycles_intra.txt
#   rphase          the same on the reference's own 1080p clip -> phase_cycles_res_clip.txt
#   tables          every device row of both SHA1 tables incl. the size-limited rows -> *_rows.txt (the tool prints at the end: the timeouts are generous;
#                   tables:camera / tables:screen / tables:dyn run one part)
#   repro:<seeds>   size-limited-slice sessions <seeds> (comma separated; s = screen content, q = low QP: e.g. s21001,q11012,1003) with slice threads,
#                   RUNS times each (default 6), for every library of LIBS (default: the product library) -> repro.txt
#   ab:<v>,<v>..    `bench.py --quick` alternating between variants, three rounds; a variant = a library tag (openh264_amd/libwelship_<tag>.so, "-" = the product
#                   library) optionally followed by :ENV=VALUE settings (e.g. -:WELSHIP_P_WAVES=14); AB_ARGS = further bench.py arguments
#   l2              LDS-DMA loads and the L2: tools/micro/lds_dma_l2 under the TCC counters; the P kernel's TCC counters with 32 / 64 / 128 / 256 pictures in flight -> l2_micro.txt, l2_by_sessions.txt
#   l2ab:<v>,<v>..  the P kernel's TCC counters and FETCH_SIZE / WRITE_SIZE per variant (as ab:) -> l2_ab.txt
#   detail:<tag>    sub-phase cycles of the claim / neighbour-load / P_Skip phases: tools/phase_profile.py with a library built with -DWH_PROF_DETAIL
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:?tag}; shift
o=gpurun_out/$tag; mkdir -p $o
T0=$SECONDS; lap() { echo "[$((SECONDS - T0)) s] $1"; }
digest() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 2), "frac", round(r["frac"], 4), "avg_launch_ms", r.get("avg_launch_ms"), "traffic x", r.get("traffic_over_algorithmic"), "verified", d.get("verified"))
for k in ("e2e", "e2e_pipelined"):
    if k in d: print(k, {x: d[k][x] for x in d[k] if x in ("frames_per_s", "frames_per_s_second_half", "steps_ahead")}, d[k].get("bitstream_vs_reference", {}).get("match"))
for k in d:
    if k.startswith("config") and k != "config": print(k, d[k].get("device_frames_per_s"), d[k].get("c_path_frames_per_s"), d[k].get("same_bitstreams"))
print("res_clip", d.get("res_clip", {}).get("value"), "intra_720p", d.get("intra_720p", {}).get("value"), "latency", d.get("latency"), "events_ms", d.get("events_ms"))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"))
PY
}
for stage in "$@"; do
  case $stage in
  tier)     WELSHIP_REQUIRE_ORACLE=1 timeout 1800 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -5 $o/pytest_gpu.txt | cut -c1-300; lap "gpu tier (serial, -x, as the driver runs it)";;
  tier_all) WELSHIP_REQUIRE_ORACLE=1 timeout 1800 python -m pytest tests -m gpu -q > $o/pytest_gpu_all.txt 2>&1; tail -15 $o/pytest_gpu_all.txt | cut -c1-300; lap "gpu tier (serial, every failure)";;
  bench)    timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; digest $o/bench_default.json; lap "bench default";;
  quick)    timeout 300 python bench.py --quick > $o/bench_quick.json 2> $o/bench_quick.err; digest $o/bench_quick.json; lap "bench --quick";;
  stats)    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$o/stats -- python $OLDPWD/bench.py --quick --steps 20 --warmup 5 > $OLDPWD/$o/prof_bench.json 2> $OLDPWD/$o/prof_bench.err )
            f=$(find $o/stats -name "*kernel_stats.csv" | head -1); cp $f $o/kernel_stats.csv 2>/dev/null; head -14 $o/kernel_stats.csv | cut -c1-200
            python -c "import json; d=json.loads(open('$o/prof_bench.json').read().strip().splitlines()[-1]); print('HIP events of the same run: avg MD launch ms', d['roofline']['avg_launch_ms'])"
            rm -rf $o/stats; lap "kernel stats";;
  pmc)      for c in FETCH_SIZE WRITE_SIZE; do
              ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OLDPWD/$o/pmc_$c -- python $OLDPWD/bench.py --quick --steps 20 --warmup 5 > $OLDPWD/$o/pmc_$c.log 2>&1 )
              python tools/pmc_summary.py $o/pmc_$c | grep -E "inter_|intra_|deblock|k_tile|k_expand|src_tile"
              rm -rf $o/pmc_$c
            done > $o/pmc_traffic.txt 2>&1; cat $o/pmc_traffic.txt; lap "pmc";;
  counters:list) ( cd /tmp && timeout 120 rocprofv3 --list-avail > $OLDPWD/$o/counters_avail.txt 2>&1 ); grep -c . $o/counters_avail.txt; grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INSTS_[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*" $o/counters_avail.txt | sort -u | tr '\n' ' '; lap "counter list";;
  counters:*) for grp in $(echo "${stage#counters:}" | tr , ' '); do
              ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $(echo $grp | tr + ' ') --output-format csv -d $OLDPWD/$o/pmc_$grp -- python $OLDPWD/bench.py --quick --steps 6 --warmup 2 > $OLDPWD/$o/pmc_$grp.log 2>&1 )
              python tools/pmc_summary.py $o/pmc_$grp | grep -E "inter_|intra_|deblock"
              rm -rf $o/pmc_$grp
            done > $o/counters.txt 2>&1; cat $o/counters.txt; lap "counters";;
  phase)    timeout 200 python tools/phase_profile.py 256 > $o/phase_cycles.txt 2>&1; head -30 $o/phase_cycles.txt; lap "phase cycles";;
  rphase)   timeout 300 python tools/phase_profile.py 256 res > $o/phase_cycles_res_clip.txt 2>&1; head -20 $o/phase_cycles_res_clip.txt; lap "phase cycles (the reference's 1080p clip)";;
  tables|tables:*)   W=${WORKERS:-48}; part=${stage#tables}; part=${part#:}
            [ -z "$part" -o "$part" = camera ] && timeout 700 python tools/sha1_table_rows.py --gom ${GOM:-2} --workers $W > $o/camera_table_1792_rows.txt 2>&1; tail -2 $o/camera_table_1792_rows.txt | cut -c1-220
            [ -z "$part" -o "$part" = screen ] && timeout 600 python tools/sha1_table_rows.py --table adobe --workers $W > $o/screen_table_896_rows.txt 2>&1; tail -2 $o/screen_table_896_rows.txt | cut -c1-220
            [ -z "$part" -o "$part" = dyn ] && timeout 300 python tools/sha1_table_rows.py --dynslice --workers $W > $o/camera_table_size_limited_512_rows.txt 2>&1; tail -2 $o/camera_table_size_limited_512_rows.txt | cut -c1-220
            [ -z "$part" -o "$part" = dyn ] && timeout 300 python tools/sha1_table_rows.py --table adobe --dynslice --workers $W > $o/screen_table_size_limited_256_rows.txt 2>&1; tail -2 $o/screen_table_size_limited_256_rows.txt | cut -c1-220
            lap "SHA1 tables";;
  repro:*)  FUZZ_DYNSLICE_KEEP=$o/streams timeout ${REPRO_TIMEOUT:-600} python tools/repro_dynslice.py "${stage#repro:}" > $o/repro.txt 2>&1; grep -E "^==|DIFF|FAILED|VARIES|summary" $o/repro.txt | cut -c1-260 | head -80; lap "repro";;
  ab:*)     # variants: a library tag ("-" = the product library), optionally followed by :ENV=VALUE settings; comma separated
            for rep in $(seq 1 ${REPS:-3}); do for v in $(echo "${stage#ab:}" | tr , ' '); do
              t=${v%%:*}; lib=openh264_amd/libwelship.so; [ "$t" != "-" ] && lib=openh264_amd/libwelship_$t.so
              envs=$(echo "${v#*:}" | tr ':' ' '); [ "$envs" = "$v" ] && envs=""
              n=$(echo "$v" | tr ':=' '__')
              env $envs WELSHIP_LIB=$PWD/$lib timeout 200 python bench.py --quick ${AB_ARGS:-} > $o/ab_${n}_$rep.json 2> $o/ab_${n}_$rep.err
              python -c "import json; d=json.loads(open('$o/ab_${n}_$rep.json').read().strip().splitlines()[-1]); print('$v', $rep, 'value', round(d['value']), 'md_ms', round(d['roofline']['avg_launch_ms'], 3), 'res_clip', round(d.get('res_clip', {}).get('value', 0)), 'verified', d.get('verified'))"
            done; done | tee $o/ab.txt; lap "A/B";;
  iwaves)   for w in 16 12 10 8 6 0; do echo "WELSHIP_I_WAVES=$w: $(WELSHIP_I_WAVES=$w timeout 120 python tools/phase_profile.py 256 synthetic intra 2>&1 | grep -E "IDR step|dependency wait|total cycles" | tr '\n' ' ')"; done > $o/intra_waves.txt 2>&1
            cat $o/intra_waves.txt; lap "IDR step by waves per intra workgroup";;
  c5trace)  for n in 1 8; do echo "== $n session(s)"; WELS_HIP_TRACE=2 WELSHIP_FRAME_STATS=1 timeout 200 python tools/config5_sessions.py $n 40 x 1080p 2>&1 | cut -c1-600; done > $o/config5_trace.txt 2>&1
            cat $o/config5_trace.txt; lap "config 5: where the binding path's time goes";;
  c5ab:*)   # variants: a library tag ("-" = the product library), optionally followed by :ENV=VALUE settings; comma separated
            for rep in 1 2; do for v in $(echo "${stage#c5ab:}" | tr , ' '); do for n in 1 8; do
              t=${v%%:*}; lib=openh264_amd/libwelship.so; [ "$t" != "-" ] && lib=openh264_amd/libwelship_$t.so
              envs=$(echo "${v#*:}" | tr ':' ' '); [ "$envs" = "$v" ] && envs=""
              r=$(env $envs WELSHIP_LIB=$PWD/$lib timeout 200 python tools/config5_sessions.py $n 40 x 1080p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['hooks_on_device']['sum_of_session_encode_fps'], d['reference_c_path']['sum_of_session_encode_fps'], d['same_bitstreams'])")
              echo "$v rep $rep sessions $n: device fps, C path fps, same bitstreams: $r"
            done; done; done > $o/config5_ab.txt 2>&1; cat $o/config5_ab.txt; lap "config 5 A/B";;
  c4ab:*)   # config 4's shape (four simulcast layers of the 1080p clip, 1 and 8 sessions) per variant, as c5ab
            for rep in 1 2; do for v in $(echo "${stage#c4ab:}" | tr , ' '); do for n in 1 8; do
              t=${v%%:*}; lib=openh264_amd/libwelship.so; [ "$t" != "-" ] && lib=openh264_amd/libwelship_$t.so
              envs=$(echo "${v#*:}" | tr ':' ' '); [ "$envs" = "$v" ] && envs=""
              r=$(env $envs WELSHIP_LIB=$PWD/$lib timeout 200 python tools/config5_sessions.py $n 54 simulcast 1080p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['hooks_on_device']['sum_of_session_encode_fps'], d['reference_c_path']['sum_of_session_encode_fps'], d['same_bitstreams'], d['hooks_on_device']['device_pictures'], d['hooks_on_device']['host_pictures'])")
              echo "$v rep $rep sessions $n: device fps, C path fps, same bitstreams, pictures on the device / left to the host: $r"
            done; done; done > $o/config4_ab.txt 2>&1; cat $o/config4_ab.txt; lap "config 4 A/B";;
  trace1)   ( cd /tmp && WELS_HIP_TRACE=2 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OLDPWD/$o/trace1 -- python $OLDPWD/tools/config5_sessions.py 1 12 x 1080p > $OLDPWD/$o/trace1.log 2>&1 )
            python tools/trace_timeline.py $o/trace1 2 > $o/trace1_timeline.txt 2>&1; tail -60 $o/trace1_timeline.txt; rm -rf $o/trace1; lap "timeline of one 1080p session through the binding";;
  iphase)   timeout 200 python tools/phase_profile.py 256 synthetic intra > $o/phase_cycles_intra.txt 2>&1; head -16 $o/phase_cycles_intra.txt; lap "phase cycles (IDR step)";;
  detail:*) WELSHIP_LIB=$PWD/openh264_amd/libwelship_${stage#detail:}.so WELSHIP_PROF_DETAIL=1 timeout 200 python tools/phase_profile.py 256 > $o/phase_cycles_detail.txt 2>&1; head -22 $o/phase_cycles_detail.txt; lap "phase cycles (detail)";;
  l2)       # does an LDS-DMA load allocate in the L2 (tools/micro/lds_dma_l2.hip), and the P kernel's L2 hit counters by pictures in flight (capacity?)
            ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OLDPWD/$o/l2_micro -- $OLDPWD/tools/micro/lds_dma_l2 > $OLDPWD/$o/l2_micro.txt 2>&1 )
            python tools/pmc_summary.py $o/l2_micro >> $o/l2_micro.txt 2>&1; rm -rf $o/l2_micro
            ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$o/l2_micro -- $OLDPWD/tools/micro/lds_dma_l2 > /dev/null 2>&1 )
            python tools/pmc_summary.py $o/l2_micro >> $o/l2_micro.txt 2>&1; rm -rf $o/l2_micro
            ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OLDPWD/$o/l2_micro -- $OLDPWD/tools/micro/lds_dma_l2 > /dev/null 2>&1 )
            python tools/pmc_summary.py $o/l2_micro >> $o/l2_micro.txt 2>&1; rm -rf $o/l2_micro
            cat $o/l2_micro.txt
            for n in 32 64 128 256; do
              ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OLDPWD/$o/l2_s$n -- python $OLDPWD/bench.py --quick --steps 6 --warmup 2 --sessions $n > $OLDPWD/$o/l2_s$n.log 2>&1 )
              echo "== $n pictures in flight"; python tools/pmc_summary.py $o/l2_s$n | grep -E "inter_"; rm -rf $o/l2_s$n
            done > $o/l2_by_sessions.txt 2>&1; cat $o/l2_by_sessions.txt; lap "L2";;
  l2ab:*)   # TCC counters of the P kernel per variant (library tag[:ENV=VALUE...], as ab:) -> l2_ab.txt
            for v in $(echo "${stage#l2ab:}" | tr , ' '); do
              t=${v%%:*}; lib=openh264_amd/libwelship.so; [ "$t" != "-" ] && lib=openh264_amd/libwelship_$t.so
              envs=$(echo "${v#*:}" | tr ':' ' '); [ "$envs" = "$v" ] && envs=""
              n=$(echo "$v" | tr ':=' '__')
              for grp in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
                ( cd /tmp && env $envs WELSHIP_LIB=$OLDPWD/$lib timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OLDPWD/$o/l2ab_$n -- python $OLDPWD/bench.py --quick --steps 6 --warmup 2 ${AB_ARGS:-} > $OLDPWD/$o/l2ab_$n.log 2>&1 )
                echo "== $v"; python tools/pmc_summary.py $o/l2ab_$n | grep -E "inter_"; rm -rf $o/l2ab_$n
              done
            done > $o/l2_ab.txt 2>&1; cat $o/l2_ab.txt; lap "L2 counters per variant";;
  *)        echo "unknown stage $stage";;
  esac
done
