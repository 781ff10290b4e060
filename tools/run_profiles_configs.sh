#!/bin/bash
# Runs on the GPU box: tools/run_profiles.sh (kernel-trace stats + separate PMC passes, `lite`) once per BASELINE configuration, so that every
# line of bench.py has a rocprof HBM figure of exactly its workload.  Usage: tools/run_profiles_configs.sh <round-tag> <group>
#   group a: cfg2 (cornell-1k), cfg3 (sponza-class-250k), cfg3s (+ shadow rays), shadow (multi-1M + shadow rays)
#   group b: cfg5 (multi-1M 3840x2160), dense (multi-1M-dense), sponza (sponza-sibenik)
#   group d: nano (nanosuit-demo: upstream's Engine_Start scene) and the bench frame itself in `full` mode (TA / TD passes)
#   group c: the opt-in in-wave compaction kernels on the bench frame (refill, block) and the wavefront form
# Summaries: python tools/profile_summary.py <tag> [kernel] afterwards (tools/summarise_profiles_configs.sh does all of them).
r=${1:-r05}; group=${2:-a}
run() { # tag, PROF_ARGS, [env assignment]
  echo "== $1: $2 $3"
  env $3 PROF_ARGS="$2" tools/run_profiles.sh $1 lite || { echo "profile $1 failed"; exit 1; }
}
case $group in
  a) run ${r}cfg2 "--scene cornell-1k"; run ${r}cfg3 "--scene sponza-class-250k"; run ${r}cfg3s "--scene sponza-class-250k --shadows"; run ${r}shadow "--shadows" ;;
  b) run ${r}cfg5 "--width 3840 --height 2160"; run ${r}dense "--scene multi-1M-dense"; run ${r}sponza "--scene sponza-sibenik" ;;
  d) run ${r}nano "--scene nanosuit-demo"; PROF_ARGS="" tools/run_profiles.sh ${r} full || exit 1 ;;
  c) run ${r}refill "" CRT_KERNEL=refill; run ${r}block "" CRT_KERNEL=block; run ${r}wavefront "" CRT_KERNEL=wavefront ;;
esac
