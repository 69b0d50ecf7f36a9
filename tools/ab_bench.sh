#!/bin/bash
# A/B timing of library variants on ONE box (boxes differ by a few percent): every variant
# named on the command line (a file name inside the package directory, e.g.
# libmopoe_hip_A.so) runs bench.py in turn, ROUNDS times over.
#   bash tools/ab_bench.sh libmopoe_hip.so libmopoe_hip_A.so ...
rounds=${ROUNDS:-3}
steps=${STEPS:-3000}
for r in $(seq $rounds); do
  for lib in "$@"; do
    out=$(MOPOE_LIB=$lib python3 bench.py --steps $steps --warmup 300 --no-cpu-baseline --quick --settle 300 2>/dev/null)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    ks=$(echo "$out" | grep -o '"kernels_avg_us": {[^}]*}')
    echo "round $r $lib ms_per_step $ms $ks"
  done
done
