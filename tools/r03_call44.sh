#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03q.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03q.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r03j_full.json 2> gpurun_out/bench_r03j_full.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --quick --no-cpu-baseline > gpurun_out/bench_r03j_driver.json 2>/dev/null; echo "driver-style rc=$?"
python - <<'PY'
import json
b = json.load(open("gpurun_out/bench_r03j_full.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["kernels_avg_us"], b["roofline"]["traffic"])
print({k: (v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"]) for k, v in b["other_configs"].items()})
for key in ("regime_n65536", "regime_n65536_bf16_operands"):
    print(key, b[key]["ms_per_step"], {k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in b[key]["kernels"].items()})
print(b["loop"]["cohorts"])
print(b["cpu_baseline"]["value"], b["eager_rocm_baseline"])
d = json.load(open("gpurun_out/bench_r03j_driver.json")); print("driver", d["ms_per_step"], d["value"])
PY
