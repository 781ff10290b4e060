tools/run_profiles_configs.sh r05 d > gpurun_out/r05_profiles_d.log 2>&1; tail -2 gpurun_out/r05_profiles_d.log
python tools/wave_timeline.py multi-1M 1920 1080 > gpurun_out/r05_wave_timeline.txt 2>&1
CRT_TL_RANKS=8 python tools/wave_timeline.py multi-1M 3840 2160 >> gpurun_out/r05_wave_timeline.txt 2>&1
python tools/cpu_baseline.py > gpurun_out/r05_cpu_baseline.md 2>&1
timeout -k 10 120 tools/ubench/gather > gpurun_out/r05_ubench_gather.txt 2>&1
CHAIN_MIX=0.868,0.722 timeout -k 10 300 tools/ubench/chain gpurun_out/r05_ubench_chain.json > gpurun_out/r05_ubench_chain.txt 2>&1
echo evidence2 done
