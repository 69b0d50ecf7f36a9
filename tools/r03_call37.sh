#!/bin/bash
# final build of the round: suite, bench line (default and the driver's invocation), one-rank
# rehearsal of the N-rank step, rocprofv3 + PMC rounds of every configuration
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03p.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03p.log
[ $rc -eq 0 ] || exit $rc
TAG=r03i bash tools/r03_profiles.sh C1 C3 C5
timeout -k 10 500 bash tools/profile_round.sh r03i_N64K --config N64K --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/prof_r03i_N64K.log 2>&1; echo "profile N64K rc=$?"
