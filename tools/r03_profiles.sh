#!/bin/bash
# rocprofv3 rounds (kernel trace + FETCH_SIZE / WRITE_SIZE passes) of the configurations named
# on the command line (default C3 C5), each through tools/profile_round.sh
set -o pipefail
tag=${TAG:-r03a}
for c in ${@:-C3 C5}; do
  timeout -k 10 400 bash tools/profile_round.sh ${tag}_$c --config $c --steps 300 --warmup 50 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/prof_${tag}_$c.log 2>&1; echo "profile $c rc=$?"
  tail -3 gpurun_out/prof_${tag}_$c.log
done
