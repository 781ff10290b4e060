#!/bin/bash
run() { CRT_SYNC_SPLIT="$1" timeout -k 10 120 python bench.py --frames-in-flight 1 --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split', '$1', '$2', d['value'], d['ms_per_step'])"; }
run "0,0"
for r in 1 2 4; do for h in 8 16 32 64 128; do run "$r,$h"; done; done
run "0,0"
