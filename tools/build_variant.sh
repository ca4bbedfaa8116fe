#!/bin/bash
# Build a variant of libetpnav_hip.so with extra -D flags on ONE source (same-box A/B through ETP_LIB):
#   tools/build_variant.sh <out.so> <source.hip> <flags...>
set -e
OUT=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
D=etpnav_amd/build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -munsafe-fp-atomics -Wno-return-type-c-linkage"
base=$(basename $SRC .hip)
EXTRA=$(python -c "from etpnav_amd.build import PER_SOURCE_FLAGS; print(' '.join(PER_SOURCE_FLAGS.get('$SRC', [])))")
/opt/rocm/bin/hipcc $F $EXTRA "$@" -c etpnav_amd/csrc/$SRC -o /tmp/${base}_variant_$$.o
OBJS=""
for o in $(python -c "from etpnav_amd.build import SOURCES; print(' '.join('etpnav_amd/build/' + s.replace('.hip', '.o') for s in SOURCES))"); do
  if [ "$(basename $o .o)" == "$base" ]; then OBJS="$OBJS /tmp/${base}_variant_$$.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS
rm -f /tmp/${base}_variant_$$.o
echo built $OUT
