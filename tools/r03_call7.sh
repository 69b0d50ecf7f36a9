set -o pipefail
python -m pytest tests/test_hip_fused.py tests/test_hip_parity.py tests/test_hip_trajectory.py tests/test_hip_stats.py tests/test_hip_topology.py tests/test_hip_surface.py -q -x > gpurun_out/t_r03h.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_r03h.log
bash tools/r03_variants.sh 2>&1 | grep -v amdgpu > gpurun_out/variants_r03b.txt; cat gpurun_out/variants_r03b.txt
