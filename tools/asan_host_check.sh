#!/bin/bash
# Host-side AddressSanitizer run of libetpnav_hip.so (SURVEY.md §5 "sanitizer run"): the library's host code (layout builder,
# bump allocators, recorder, argument checks, dropout generator) instrumented with -fsanitize=address (device code is not
# instrumented: -fno-gpu-sanitize), exercised WITHOUT a GPU through the C ABI.   bash tools/asan_host_check.sh
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/etp_asan; mkdir -p $OUT
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
for f in planner capi graphrec comm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fno-gpu-rdc -fsanitize=address -fno-gpu-sanitize -shared-libsan \
      -Wno-unused-result -munsafe-fp-atomics -Wno-return-type-c-linkage -c etpnav_amd/csrc/$f.hip -o $OUT/$f.o &
done
wait
# kernels' translation units un-instrumented (their host side is launch stubs only)
# (object list from etpnav_amd/build.py SOURCES, so a new translation unit cannot be forgotten here)
REST=$(python -c "from etpnav_amd.build import SOURCES; print(' '.join('etpnav_amd/build/' + s.replace('.hip', '.o') for s in SOURCES if s.replace('.hip', '') not in ('planner', 'capi', 'graphrec', 'comm')))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libsan -o $OUT/libetpnav_hip_asan.so \
    $OUT/planner.o $OUT/capi.o $OUT/graphrec.o $OUT/comm.o $REST
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 ETP_ASAN_LIB=$OUT/libetpnav_hip_asan.so python tools/asan_host_driver.py
