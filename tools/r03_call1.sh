#!/bin/bash
# round 3, first GPU call: GPU tests, the one-rank rehearsal of the N-rank step (one-call
# RCCL vs the spelled-out c10d form), rocprofv3 rounds of configs[2] / configs[4]
set -o pipefail
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/t_r03a.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/t_r03a.log
timeout -k 10 200 python bench.py --force-dist --quick --no-cpu-baseline --steps 2000 --warmup 200 > gpurun_out/b_r03a_dp1_rccl.json 2> gpurun_out/b_r03a_dp1_rccl.err; echo "rccl rc=$?"
MOPOE_EXCHANGE=c10d timeout -k 10 200 python bench.py --force-dist --quick --no-cpu-baseline --steps 2000 --warmup 200 > gpurun_out/b_r03a_dp1_c10d.json 2> gpurun_out/b_r03a_dp1_c10d.err; echo "c10d rc=$?"
timeout -k 10 200 python bench.py --quick --no-cpu-baseline --steps 2000 --warmup 200 > gpurun_out/b_r03a_single.json 2> gpurun_out/b_r03a_single.err; echo "single rc=$?"
for c in C3 C5; do
  timeout -k 10 300 bash tools/profile_round.sh r03a_$c --config $c --steps 300 --warmup 50 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/prof_r03a_$c.log 2>&1; echo "profile $c rc=$?"
done
