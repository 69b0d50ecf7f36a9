#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TOPOLOGY_CONFIGS=C1 TOPOLOGY_ONLY="enc 2, dec 1, dropout 0.2" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_topo -o run -- python3 tools/topology_bench.py > gpurun_out/topo_prof.log 2>&1; echo "rc=$?"; grep "us/step" gpurun_out/topo_prof.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_topo/**/*kernel_stats.csv", recursive=True)
tot = 0
for row in list(csv.DictReader(open(f[0])))[:14]:
    print("%-100s %6s %9.1f us  total %8.1f ms" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
t = glob.glob("gpurun_out/prof_topo/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step in the middle: the kernels between two consecutive g_fuse_fwd launches
idx = [i for i, r in enumerate(rows) if "g_fuse_fwd" in r["Kernel_Name"]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%8.1f us  +%6.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:80]))
PY

