#!/bin/bash
# CPU-only: builds libcrt_host.so with AddressSanitizer + UBSan, runs the CPU test suite against it, restores the normal
# library. (GPU sanitizers are not available on the pool; the device code is covered by the parity tests.)
#   tools/sanitize_host.sh [pytest args]      default: -m "not gpu" without the brute-force test (it only exercises the oracle)
set -e
cd "$(dirname "$0")/.."
H=clraytracer_amd/host
cp $H/libcrt_host.so /tmp/libcrt_host.normal.so
trap 'cp /tmp/libcrt_host.normal.so '$H'/libcrt_host.so' EXIT
g++ -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $H/libcrt_host.so \
    $H/AssetManager.cpp $H/MeshCache.cpp $H/JpegDecode.cpp $H/BVH.cpp $H/CPURayTrace.cpp $H/Renderer.cpp $H/ResourceManager.cpp $H/crt_host_c.cpp \
    -Lclraytracer_amd/csrc -lcrt_hip -Wl,-rpath,"$PWD/clraytracer_amd/csrc"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
if [ $# -gt 0 ]; then python -m pytest "$@"; else python -m pytest tests -q -m "not gpu" --deselect tests/test_brute_force.py; fi
