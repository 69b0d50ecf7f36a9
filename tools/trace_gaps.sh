set -e -o pipefail
out=$PWD/gpurun_out/trace_tmp; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-roofline > "$out/log" 2>&1
head -5 "$out/run_kernel_stats.csv" | cut -c1-160
python3 - "$out/run_kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "k_l" in r["Kernel_Name"] or "k_w" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import statistics
gaps = {}
for a, b in zip(rows[600:-1], rows[601:]):
    ka = a["Kernel_Name"].split("::")[-1].split("(")[0]; kb = b["Kernel_Name"].split("::")[-1].split("(")[0]
    gaps.setdefault((ka, kb), []).append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for k, v in gaps.items():
    print(k, "gap median %.2f us" % (statistics.median(v) / 1e3), len(v))
PY
