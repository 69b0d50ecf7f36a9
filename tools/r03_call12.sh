#!/bin/bash
# final-build pass A: the whole GPU suite, the bench line (default and the driver's
# invocation), the topology figures
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03e.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03e.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/bench_r03e_full.json 2> gpurun_out/bench_r03e_full.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --quick --no-cpu-baseline > gpurun_out/bench_r03e_driver.json 2>/dev/null; echo "driver-style rc=$?"
python tools/topology_bench.py > gpurun_out/topology_bench_r03e.txt 2>&1; echo "topology rc=$?"; cat gpurun_out/topology_bench_r03e.txt
python - <<'PY'
import json
b = json.load(open("gpurun_out/bench_r03e_full.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["kernels_avg_us"])
print({k: v["ms_per_step"] for k, v in b["other_configs"].items()})
print(b["regime_n65536"]["ms_per_step"], {k: v["avg_us"] for k, v in b["regime_n65536"]["kernels"].items()})
print(b["loop"]["cohorts"])
d = json.load(open("gpurun_out/bench_r03e_driver.json")); print("driver", d["ms_per_step"], d["value"])
PY
