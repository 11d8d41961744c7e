#!/bin/bash
# IDR step of 256 four-slice 1080p pictures with 16 / 12 / 8 / 6 waves per intra workgroup (WELSHIP_I_WAVES)
cd "$(dirname "$0")/.."
for w in 16 12 10 8 6 0; do echo "WELSHIP_I_WAVES=$w: $(WELSHIP_I_WAVES=$w timeout 120 python tools/phase_profile.py 256 synthetic intra 2>&1 | grep -E "IDR step|dependency wait|total cycles" | tr '\n' ' ')"; done
