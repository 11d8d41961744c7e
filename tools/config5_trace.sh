#!/bin/bash
# where the binding path's time goes at BASELINE config 5's size: 1 / 8 sessions 1080p, rate control, raster slices (WELS_HIP_TRACE=2, WELSHIP_FRAME_STATS=1)
cd "$(dirname "$0")/.."
for n in 1 8; do
  echo "== $n session(s)"
  WELS_HIP_TRACE=2 WELSHIP_FRAME_STATS=1 timeout 200 python tools/config5_sessions.py $n 40 x 1080p 2>&1 | cut -c1-600
done
