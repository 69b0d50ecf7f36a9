#!/bin/bash
# The round's rocprofv3 record of one build (run through gpurun):
#   kernel trace + FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh) and the SQ
#   instruction-issue passes (tools/pmc_insts.py) of the configurations named on the command line.
#   TAG=r04e bash tools/r04_profiles.sh C1 C3 C5 N64K
set -o pipefail
tag=${TAG:-r04}
for c in ${@:-C1 C3 C5 N64K}; do
  steps=300; [ "$c" = N64K ] && steps=30
  timeout -k 10 500 bash tools/profile_round.sh ${tag}_$c --config $c --steps $steps --warmup 20 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/prof_${tag}_$c.log 2>&1
  echo "traffic $c rc=$?"
  timeout -k 10 500 python3 tools/pmc_insts.py $tag $c --steps $((steps / 3)) --warmup 10 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/pmc_${tag}_$c.log 2>&1
  echo "insts $c rc=$?"
done
