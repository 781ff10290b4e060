#!/bin/bash
# GPU-box side: round 6's differential sweeps against the oracle: the default kernel and the three opt-in compaction forms, each on
# pathological geometry (tools/fuzz_many.py: 2 cameras + 4 lattice frames per scene; the default kernel also through crt_query_hits and with shadow rays)
# and on seeded viewpoints of the full-size scenes (tools/random_views.py). Under CRT_KERNEL the tools use only what a form renders and check every
# frame's kernel name (crt_debug_last_kernel), so a leg cannot test another kernel than the one it names. A failing leg stops the script (set -e).
set -e -o pipefail
out=gpurun_out
: > $out/r06_fuzz.txt; : > $out/r06_random_views.txt
{ echo "# default kernel"; timeout -k 10 400 python tools/fuzz_many.py 100000 1500; } >> $out/r06_fuzz.txt 2>&1
for k in wavefront refill block ldstop; do { echo "# CRT_KERNEL=$k"; CRT_KERNEL=$k timeout -k 10 300 python tools/fuzz_many.py 101000 600; } >> $out/r06_fuzz.txt 2>&1; done
{ echo "# default kernel"; timeout -k 10 300 python tools/random_views.py 30; } >> $out/r06_random_views.txt 2>&1
for k in wavefront refill block ldstop; do { echo "# CRT_KERNEL=$k"; CRT_KERNEL=$k timeout -k 10 300 python tools/random_views.py 12; } >> $out/r06_random_views.txt 2>&1; done
grep -c Traceback $out/r06_fuzz.txt $out/r06_random_views.txt || true
tail -2 $out/r06_fuzz.txt; tail -2 $out/r06_random_views.txt
