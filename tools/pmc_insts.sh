#!/bin/bash
# Instruction mix of the three kernels (rocprofv3 PMC, own pass, kernel-trace only).
set -e -o pipefail
out=$PWD/gpurun_out/pmc_insts
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d "$out/a" -o run -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline > "$out/a.log" 2>&1
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
    print("%-10s %-18s %12.0f per launch" % (k, c, tot[(k, c)] / cnt[(k, c)]))
PY
