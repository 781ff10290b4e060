#!/bin/bash
# GPU-box side: the larger round-6 sweeps on the FINAL library (default kernel): pathological scenes (2 cameras + 4 lattice frames + 4096 query rays each), the instance cull,
# soak, device BVH builds. A failing leg stops the script.
set -e -o pipefail
out=gpurun_out
timeout -k 10 900 python tools/fuzz_many.py 110000 6000 > $out/r06_fuzz_b.txt 2>&1; tail -1 $out/r06_fuzz_b.txt | cut -c1-300
timeout -k 10 200 python tools/fuzz_cull.py 3000 300 > $out/r06_fuzz_cull.txt 2>&1; tail -1 $out/r06_fuzz_cull.txt | cut -c1-300
timeout -k 10 150 python tools/soak.py > $out/r06_soak.txt 2>&1; tail -1 $out/r06_soak.txt | cut -c1-300
timeout -k 10 200 python tools/fuzz_bvh_build.py 120000 3000 > $out/r06_fuzz_build.txt 2>&1; tail -1 $out/r06_fuzz_build.txt | cut -c1-300
