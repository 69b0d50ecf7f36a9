#!/bin/bash
# Instruction-cache behaviour of a general topology's chain of launches (rocprofv3 PMC, own pass):
#   TOPOLOGY_ONLY="enc 2, dec 1, dropout 0.2" TOPOLOGY_CONFIGS=C1 bash tools/pmc_icache_chain.sh
set -e -o pipefail
out=$PWD/gpurun_out/pmc_icache_chain
rm -rf "$out"; mkdir -p "$out"
export TOPOLOGY_ONLY="${TOPOLOGY_ONLY:-enc 2, dec 1, dropout 0.2}" TOPOLOGY_CONFIGS="${TOPOLOGY_CONFIGS:-C1}"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$out/a" -o run -- python3 tools/topology_bench.py > "$out/a.log" 2>&1 || { tail -5 "$out/a.log"; exit 1; }
python3 - "$out/a" <<'PY'
import csv, glob, os, sys
tot, cnt = {}, {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not name.startswith(("k_", "g_")):
            continue
        key = (name, row["Counter_Name"])
        tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"])
        cnt[key] = cnt.get(key, 0) + 1
names = sorted({k for k, _ in tot})
print("%-28s %10s %10s %8s %10s" % ("kernel", "icache req", "misses", "miss %", "waves"))
for n in names:
    g = lambda c: tot.get((n, c), 0.0) / max(cnt.get((n, c), 1), 1)
    print("%-28s %10.0f %10.0f %8.1f %10.0f" % (n, g("SQC_ICACHE_REQ"), g("SQC_ICACHE_MISSES"),
                                               100.0 * g("SQC_ICACHE_MISSES") / max(g("SQC_ICACHE_REQ"), 1), g("SQ_WAVES")))
PY
