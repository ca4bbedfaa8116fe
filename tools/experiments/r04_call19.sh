#!/bin/bash
export TMPDIR=/tmp FLAKE_AUX_ONLY=1
S=tools/experiments/r04_gmap_pos_flake2.py
echo "== default"; python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== system-scope atomics in block_flush"; ETP_LIB=etpnav_amd/lib_dbg_sc1.so python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== gmap_embed_bwd grid 1"; ETP_DBG_GMAP_GRID=1 python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== gmap_embed_bwd grid 8"; ETP_DBG_GMAP_GRID=8 python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== no fused colsum in grouped wgrad"; ETP_DBG_NOCOLSUM=1 python $S 2>&1 | grep "0.weight" | cut -c1-150
