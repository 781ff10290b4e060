#!/bin/bash
# GPU-box side: the bench lines of tools/run_round_evidence.sh without the profiling passes.
tag=${1:-r03}
out=gpurun_out
: > $out/${tag}_bench_lines.jsonl
python bench.py >> $out/${tag}_bench_lines.jsonl 2>$out/${tag}_bench.err
for sc in multi-1M-dense sponza-sibenik nanosuit-demo sponza-class-250k cornell-1k; do
  python bench.py --scene $sc --no-cpu-baseline >> $out/${tag}_bench_lines.jsonl 2>>$out/${tag}_bench.err
done
python bench.py --shadows --no-cpu-baseline >> $out/${tag}_bench_lines.jsonl 2>>$out/${tag}_bench.err
python bench.py --scene sponza-class-250k --shadows --no-cpu-baseline >> $out/${tag}_bench_lines.jsonl 2>>$out/${tag}_bench.err   # BASELINE config 3 as written
python bench.py --width 3840 --height 2160 --no-cpu-baseline >> $out/${tag}_bench_lines.jsonl 2>>$out/${tag}_bench.err
python bench.py --frames-in-flight 1 --no-cpu-baseline --no-extras >> $out/${tag}_bench_lines.jsonl 2>>$out/${tag}_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras >> $out/${tag}_bench_lines.jsonl 2>>$out/${tag}_bench.err   # the driver's step counts
echo bench lines done
