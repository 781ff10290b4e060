#!/bin/bash
# GPU-box side: the kernel FORMS on more pathological scenes and views than tools/run_fuzz_r06.sh (2,000 scenes x 6 frames and 30 views x 6 scenes per form), final round-6 library.
set -e -o pipefail
out=gpurun_out
: > $out/r06_fuzz_forms.txt; : > $out/r06_random_views_forms.txt
for k in wavefront refill block ldstop; do { echo "# CRT_KERNEL=$k"; CRT_KERNEL=$k timeout -k 10 400 python tools/fuzz_many.py 130000 2000; } >> $out/r06_fuzz_forms.txt 2>&1; tail -1 $out/r06_fuzz_forms.txt | cut -c1-220; done
for k in wavefront refill block ldstop; do { echo "# CRT_KERNEL=$k"; CRT_KERNEL=$k timeout -k 10 300 python tools/random_views.py 30; } >> $out/r06_random_views_forms.txt 2>&1; tail -1 $out/r06_random_views_forms.txt | cut -c1-220; done
