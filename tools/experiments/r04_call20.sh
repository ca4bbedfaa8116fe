#!/bin/bash
export TMPDIR=/tmp FLAKE_AUX_ONLY=1
S=tools/experiments/r04_gmap_pos_flake2.py
echo "== launch_bounds(256,2): no AGPRs"; ETP_LIB=etpnav_amd/lib_dbg_lb2.so python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== default"; python $S 2>&1 | grep "0.weight" | cut -c1-150
