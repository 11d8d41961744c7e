#!/bin/bash
# Round 3: which HIP streams share a hardware queue (tools/micro/stream_alias.hip)
cd "$(dirname "$0")/.."
o=gpurun_out/r03_y; rm -rf $o; mkdir -p $o
timeout 60 tools/micro/stream_alias | tee $o/stream_alias.txt
GPU_MAX_HW_QUEUES=8 timeout 60 tools/micro/stream_alias | head -3 | tee $o/stream_alias_hwq8.txt
