#!/bin/bash
export TMPDIR=/tmp FLAKE_AUX_ONLY=1
S=tools/experiments/r04_gmap_pos_flake2.py
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== HSA_ENABLE_DEBUG... default"; python $S 2>&1 | grep "0.weight" | cut -c1-150
echo "== AMD_SERIALIZE_KERNEL=3 (serialised launches)"; AMD_SERIALIZE_KERNEL=3 python $S 2>&1 | grep "0.weight" | cut -c1-150
