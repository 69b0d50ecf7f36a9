#!/bin/bash
# A/B of the four-row form (MOPOE_QUAD=1) against the default forms on ONE box.
rounds=${ROUNDS:-3}
steps=${STEPS:-3000}
for r in $(seq $rounds); do
  for q in 0 1; do
    out=$(MOPOE_QUAD=$q python3 bench.py --steps $steps --warmup 300 --no-cpu-baseline --quick 2>/dev/null)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    ks=$(echo "$out" | grep -o '"kernels_avg_us": {[^}]*}')
    echo "round $r quad=$q ms_per_step $ms $ks"
  done
done
