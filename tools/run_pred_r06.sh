#!/bin/bash
# GPU-box side: round 6's single-GPU PREDICTION of the N = 1, 2, 4, 8 bench lines at the driver's command (--steps 20 --warmup 5): one process
# per frames-in-flight setting, as bench.py's ranks are (GPU_MAX_HW_QUEUES is read once per process), then the two barriers' cost on this host.
set -e -o pipefail
out=gpurun_out/r06_pred_driver.txt
: > $out
CRT_PRED_N=1,2 timeout -k 10 200 python tools/predict_scaling.py --driver 20 5 >> $out 2>&1
CRT_PRED_N=4,8 timeout -k 10 300 python tools/predict_scaling.py --driver 20 5 >> $out 2>&1
echo "## 200 steps (steady state dominates)" >> $out
CRT_PRED_N=8 CRT_PRED_REPS=3 timeout -k 10 200 python tools/predict_scaling.py --driver 200 5 >> $out 2>&1
grep "^#" $out
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29557 tools/barrier_cost.py 2.5 200 2>&1 | grep "ranks on" > gpurun_out/r06_barrier_cost.txt
cat gpurun_out/r06_barrier_cost.txt
