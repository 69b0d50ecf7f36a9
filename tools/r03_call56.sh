#!/bin/bash
for lib in libmopoe_hip.so libmopoe_hip_vLW2.so; do
echo "== $lib"
MOPOE_LIB=$lib python tools/quad_range_ab.py 2>&1 | grep -v amdgpu.ids | head -7
MOPOE_LIB=$lib TOPOLOGY_CONFIGS=C1 python tools/topology_bench.py 2>&1 | grep -v amdgpu.ids
done
