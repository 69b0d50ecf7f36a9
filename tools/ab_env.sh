#!/bin/bash
# A/B of environment settings on ONE box: every argument is an "NAME=VALUE[,NAME=VALUE...]" set
#   bash tools/ab_env.sh HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1
rounds=${ROUNDS:-2}
steps=${STEPS:-3000}
for r in $(seq $rounds); do
  for set in "$@"; do
    out=$(env $(echo "$set" | tr ',' ' ') python3 bench.py --steps $steps --warmup 300 --no-cpu-baseline --quick 2>/dev/null)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    ks=$(echo "$out" | grep -o '"kernels_avg_us": {[^}]*}')
    echo "round $r [$set] ms_per_step $ms $ks"
  done
done
