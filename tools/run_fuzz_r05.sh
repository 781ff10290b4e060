#!/bin/bash
# GPU-box side: round 5's differential sweeps against the oracle -- the default kernel on the round-5 library, and the three opt-in compaction
# kernels (new or rewritten this round) on pathological geometry and on seeded viewpoints of the full-size scenes.
# ROUND-6 NOTE: as run in round 5 the wavefront legs crashed (shadow flag on odd seeds -> CRT_E_UNSUPPORTED) and the odd seeds of the refill / block legs were
# rendered by the default kernel (silent fall-back, removed since). Kept as the record of what produced profiles/r05_fuzz.txt; use tools/run_fuzz_r06.sh.
out=gpurun_out
{ echo "# default kernel"; timeout -k 10 500 python tools/fuzz_many.py 90000 1500; } > $out/r05_fuzz.txt 2>&1
for k in wavefront refill block; do { echo "# CRT_KERNEL=$k"; CRT_KERNEL=$k timeout -k 10 300 python tools/fuzz_many.py 91000 600; } >> $out/r05_fuzz.txt 2>&1; done
{ echo "# default kernel"; timeout -k 10 400 python tools/random_views.py 30; } > $out/r05_random_views.txt 2>&1
for k in wavefront refill block; do { echo "# CRT_KERNEL=$k"; CRT_KERNEL=$k timeout -k 10 300 python tools/random_views.py 12; } >> $out/r05_random_views.txt 2>&1; done
tail -3 $out/r05_fuzz.txt; tail -3 $out/r05_random_views.txt
