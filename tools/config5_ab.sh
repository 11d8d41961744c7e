#!/bin/bash
# config 5's shape (N sessions 1080p, rate control, raster slices) through the binding: tools/config5_ab.sh <variant> ...
# a variant is a library tag (openh264_amd/libwelship_<tag>.so; "-" = the product library), optionally followed by :ENV=VALUE settings
cd "$(dirname "$0")/.."
for rep in 1 2; do for v in "$@"; do for n in 1 8; do
  t=${v%%:*}; lib=openh264_amd/libwelship.so; [ "$t" != "-" ] && lib=openh264_amd/libwelship_$t.so
  envs=$(echo "${v#*:}" | tr ':' ' '); [ "$envs" = "$v" ] && envs=""
  r=$(env $envs WELSHIP_LIB=$PWD/$lib timeout 200 python tools/config5_sessions.py $n 40 x 1080p 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['hooks_on_device']['sum_of_session_encode_fps'], d['reference_c_path']['sum_of_session_encode_fps'], d['same_bitstreams'])")
  echo "$v rep $rep sessions $n: device fps, C path fps, same bitstreams: $r"
done; done; done
