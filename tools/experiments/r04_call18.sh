#!/bin/bash
export TMPDIR=/tmp FLAKE_AUX_ONLY=1
S=tools/experiments/r04_gmap_pos_flake2.py
echo "== default"; python $S 2>&1 | grep "0.weight"
echo "== ETP_SKIP_LN=red"; ETP_SKIP_LN=red python $S 2>&1 | grep "0.weight"
echo "== ETP_SKIP_WGRAD=1"; ETP_SKIP_WGRAD=1 python $S 2>&1 | grep "0.weight"
echo "== ETP_WGRAD_GROUP=0"; ETP_WGRAD_GROUP=0 python $S 2>&1 | grep "0.weight"
echo "== ETP_MM32=0"; ETP_MM32=0 python $S 2>&1 | grep "0.weight"
echo "== ETP_LNBWD_TWO_STAGE=0"; ETP_LNBWD_TWO_STAGE=0 python $S 2>&1 | grep "0.weight"
echo "== ETP_DTXT_STREAM=0"; ETP_DTXT_STREAM=0 python $S 2>&1 | grep "0.weight"
