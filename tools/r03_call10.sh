set -o pipefail
python bench.py 2> gpurun_out/bench_full_r03_d.err | grep '^{' > gpurun_out/bench_full_r03_d.json; echo "full rc=$?"
python bench.py --steps 20 --warmup 5 --quick 2> gpurun_out/bench_driver_r03_d.err | grep '^{' > gpurun_out/bench_driver_r03_d.json; echo "driver rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full_r03_d.json"))
print("value", d["value"], d["ms_per_step"], "cold", d["cold_start"]["ms_per_step"], "long", d["long_run"]["ms_per_step"])
print("roofline", d["roofline"]["frac"], d["roofline"]["kernels_avg_us"], d["roofline"]["traffic"])
for k, v in d["other_configs"].items():
    print(k, v["ms_per_step"], v["roofline"]["frac"], v["roofline"]["traffic"], v["roofline"]["kernels_avg_us"])
print("loop", d["loop"]["cohorts"])
print("regime", {k: (v["avg_us"], v["frac_f32_mfma_peak"]) for k, v in d["regime_n65536"]["kernels"].items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["by_threads"])
print("eager", d["eager_rocm_baseline"]["ms_per_step"])
d2 = json.load(open("gpurun_out/bench_driver_r03_d.json"))
print("driver invocation: value", d2["value"], d2["ms_per_step"], "cold", d2["cold_start"]["ms_per_step"], "long", d2["long_run"]["ms_per_step"])
PY
