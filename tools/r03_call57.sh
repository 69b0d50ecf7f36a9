#!/bin/bash
set -o pipefail
python tools/quad_range_ab.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03s.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_r03s.log
