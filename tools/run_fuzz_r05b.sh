#!/bin/bash
# GPU-box side: the larger round-5 sweeps (default kernel, final library): pathological scenes, the instance cull, soak, device BVH builds.
out=gpurun_out
timeout -k 10 700 python tools/fuzz_many.py 100000 6000 > $out/r05_fuzz_b.txt 2>&1; tail -1 $out/r05_fuzz_b.txt
timeout -k 10 200 python tools/fuzz_cull.py 2000 200 > $out/r05_fuzz_cull.txt 2>&1; tail -1 $out/r05_fuzz_cull.txt
timeout -k 10 150 python tools/soak.py > $out/r05_soak.txt 2>&1; tail -1 $out/r05_soak.txt
timeout -k 10 200 python tools/fuzz_bvh_build.py 97000 3000 > $out/r05_fuzz_build.txt 2>&1; tail -1 $out/r05_fuzz_build.txt
