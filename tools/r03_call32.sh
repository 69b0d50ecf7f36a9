#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03n.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03n.log
[ $rc -eq 0 ] || exit $rc
python tools/topology_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/topology_bench_r03i.txt
python tools/daa_bench.py 2>&1 | grep -v amdgpu.ids | tail -6
