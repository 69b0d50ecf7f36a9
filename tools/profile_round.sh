#!/bin/bash
# Collects what profiles/ holds for one build, on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (kernel durations),
#   2. + 3. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of the same command, each
#      in a pass of its own (MI355X_MICROARCH.md, HBM section),
# then tools/pmc_summary.py turns the CSVs into gpurun_out/prof_<tag>/{kernel_stats.csv,
# pmc_traffic.json}.  Usage: bash tools/profile_round.sh <tag> [bench args]
set -e -o pipefail
tag=${1:-r00}; shift || true
args=${@:---steps 300 --warmup 50 --no-cpu-baseline --no-roofline --quick}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 bench.py $args > "$out/trace.log" 2>&1
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o run -- python3 bench.py $args > "$out/pmc_fetch.log" 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o run -- python3 bench.py $args > "$out/pmc_write.log" 2>&1
echo "WRITE_SIZE pass done"
python3 tools/pmc_summary.py "$out" "$tag"
