#!/bin/bash
export TMPDIR=/tmp FLAKE_AUX_ONLY=1
S=tools/experiments/r04_gmap_pos_flake2.py
echo "== ETP_STREAM_PRIO=0"; ETP_STREAM_PRIO=0 python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== ETP_STREAM_PRIO=0 again"; ETP_STREAM_PRIO=0 python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== default"; python $S 2>&1 | grep "0.weight" | cut -c1-150
