#!/bin/bash
# Instruction-cache behaviour of the step's kernels (rocprofv3 PMC, own pass, kernel-trace only).
set -e -o pipefail
out=$PWD/gpurun_out/pmc_icache
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$out/a" -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > "$out/a.log" 2>&1 || { tail -5 "$out/a.log"; exit 1; }
python3 - "$out/a" <<'PY'
import csv, glob, os, sys
tot = {}
cnt = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"]
        short = next((k for k in ("k_linear", "k_latent", "k_wgrad", "k_fused") if k + "(" in name or k + "<" in name), None)
        if not short:
            continue
        key = (short, row["Counter_Name"])
        tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"])
        cnt[key] = cnt.get(key, 0) + 1
for (k, c) in sorted(tot):
    print("%-10s %-20s %14.0f per launch" % (k, c, tot[(k, c)] / cnt[(k, c)]))
PY
