#!/bin/bash
# final-build pass B: rocprofv3 rounds of every configuration (+ 65,536 rows), the row
# group's exit timeline, kernarg placement A/B
set -o pipefail
mkdir -p gpurun_out
TAG=r03e bash tools/r03_profiles.sh C1 C3 C5
timeout -k 10 500 bash tools/profile_round.sh r03e_N64K --config N64K --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/prof_r03e_N64K.log 2>&1; echo "profile N64K rc=$?"; tail -12 gpurun_out/prof_r03e_N64K.log
python tools/exit_timeline.py C1 > gpurun_out/exit_timeline_C1.txt 2>&1; echo "exit timeline rc=$?"; cat gpurun_out/exit_timeline_C1.txt
ROUNDS=2 STEPS=2000 bash tools/ab_env.sh HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 > gpurun_out/ab_kernarg.txt 2>&1; cat gpurun_out/ab_kernarg.txt
