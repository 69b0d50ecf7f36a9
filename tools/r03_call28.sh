#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_hip_topology.py tests/test_hip_parity.py tests/test_hip_stats.py -x -q > gpurun_out/t_r03m.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03m.log
[ $rc -eq 0 ] || exit $rc
python tools/topology_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/topology_bench_r03h.txt
