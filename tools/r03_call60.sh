#!/bin/bash
# the round's last build: suite, smoke, bench (driver's invocation, full), profile rounds C1 / C3 / 65,536 rows
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03t.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03t.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03n_driver_full.json 2> gpurun_out/bench_r03n.err; echo "bench (driver invocation) rc=$?"
python bench.py --quick --no-cpu-baseline > gpurun_out/bench_r03n_default_quick.json 2>/dev/null; echo "bench default rc=$?"
TAG=r03n bash tools/r03_profiles.sh C1 C3 C5
timeout -k 10 500 bash tools/profile_round.sh r03n_N64K --config N64K --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --no-roofline --quick > gpurun_out/prof_r03n_N64K.log 2>&1; echo "profile N64K rc=$?"
python - <<'PY'
import json
b = json.load(open("gpurun_out/bench_r03n_driver_full.json"))
print("driver invocation:", b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["kernels_avg_us"], "long_run", b["long_run"]["ms_per_step"])
print({k: (v["ms_per_step"], v["roofline"]["kernels_avg_us"]) for k, v in b["other_configs"].items()})
print(b["regime_n65536"]["ms_per_step"], b["regime_n65536_bf16_operands"]["ms_per_step"], {k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in b["regime_n65536"]["kernels"].items()})
print(b["loop"]["cohorts"])
d = json.load(open("gpurun_out/bench_r03n_default_quick.json")); print("default:", d["ms_per_step"], d["value"], d["roofline"]["frac"])
PY
