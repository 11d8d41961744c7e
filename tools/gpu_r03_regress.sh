#!/bin/bash
# Round 3, regression at the round's last kernel code: every device row of both SHA1 tables (the size-limited rows included) on the MI355X
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_regress; rm -rf $o; mkdir -p $o
T0=$SECONDS; lap() { echo "[$((SECONDS - T0)) s] $1"; }
W=${WORKERS:-48}
timeout 400 python tools/sha1_table_rows.py --workers $W > $o/camera_table_1792_rows.txt 2>&1; tail -3 $o/camera_table_1792_rows.txt | cut -c1-220; lap "camera table"
timeout 300 python tools/sha1_table_rows.py --table adobe --workers $W > $o/screen_table_896_rows.txt 2>&1; tail -3 $o/screen_table_896_rows.txt | cut -c1-220; lap "screen table"
timeout 300 python tools/sha1_table_rows.py --dynslice --workers $W > $o/camera_table_size_limited_512_rows.txt 2>&1; tail -3 $o/camera_table_size_limited_512_rows.txt | cut -c1-220; lap "camera table, size-limited rows"
timeout 300 python tools/sha1_table_rows.py --table adobe --dynslice --workers $W > $o/screen_table_size_limited_256_rows.txt 2>&1; tail -3 $o/screen_table_size_limited_256_rows.txt | cut -c1-220; lap "screen table, size-limited rows"
