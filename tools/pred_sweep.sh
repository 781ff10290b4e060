#!/bin/bash
# GPU-box side: tools/predict_scaling.py --driver for N = 8 (and 4) under the knobs that could shorten a small share's burst
out=gpurun_out/r06_pred_sweep.txt
: > $out
run() { echo "## $*" >> $out; env "$@" CRT_PRED_REPS=3 timeout -k 10 120 python tools/predict_scaling.py --driver 20 5 2>&1 | grep "^# [0-9]" >> $out || return 1; }
run CRT_PRED_N=4,8 || exit 1
run CRT_PRED_N=4,8 CRT_FEEDBACK_ASYNC=1 || exit 1
run CRT_PRED_N=8 CRT_FEEDBACK_ASYNC=1 CRT_SPLIT=0 || exit 1
run CRT_PRED_N=8 CRT_FEEDBACK_ASYNC=1 CRT_SPLIT=16 || exit 1
run CRT_PRED_N=8 CRT_FEEDBACK_ASYNC=1 CRT_SPLIT=48 || exit 1
run CRT_PRED_N=8 CRT_BAND=8 || exit 1
run CRT_PRED_N=8 CRT_BAND=8 CRT_FEEDBACK_ASYNC=1 || exit 1
run CRT_PRED_N=8 CRT_BAND=32 || exit 1
run CRT_PRED_N=8 CRT_FLIGHT=4 || exit 1
run CRT_PRED_N=8 CRT_FLIGHT=6 || exit 1
run CRT_PRED_N=4,8 CRT_FLIGHT=5 || exit 1
run CRT_PRED_N=4 CRT_FLIGHT=4 || exit 1
cat $out
