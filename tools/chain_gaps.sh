# kernel durations and the gaps between them for one chain topology (rocprofv3 --kernel-trace):
#   TOPOLOGY_ONLY="enc 2, dec 1, dropout 0.2" TOPOLOGY_CONFIGS=C1 bash tools/chain_gaps.sh
set -e -o pipefail
out=$PWD/gpurun_out/chain_trace; rm -rf $out; mkdir -p $out
export TOPOLOGY_ONLY="${TOPOLOGY_ONLY:-enc 2, dec 1, dropout 0.2}" TOPOLOGY_CONFIGS="${TOPOLOGY_CONFIGS:-C1}"
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 tools/topology_bench.py > "$out/log" 2>&1
cat "$out/log" | tail -3
python3 - "$out/run_kernel_trace.csv" <<'PY'
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
def short(r):
    n = r["Kernel_Name"]
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]
rows = [r for r in rows if short(r).startswith(("k_", "g_"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = from a kernel after k_wgrad to the next k_wgrad; take the step pattern from the tail
names = [short(r) for r in rows]
# (a step ends with its LAST weight-gradient launch: more than fifteen jobs take two)
ends = [i for i, n in enumerate(names) if n.startswith("k_wgrad") and not (i + 1 < len(names) and names[i + 1].startswith("k_wgrad"))]
per = ends[-1] - ends[-2]
print("launches per step:", per)
first = ends[-200] + 1
dur = [[] for _ in range(per)]; gap = [[] for _ in range(per)]
for s in range(first, ends[-2] + 1, per):
    for k in range(per):
        r = rows[s + k]
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        gap[k].append(int(r["Start_Timestamp"]) - int(rows[s + k - 1]["End_Timestamp"]))
td = tg = 0
for k in range(per):
    d, g = statistics.median(dur[k]) / 1e3, statistics.median(gap[k]) / 1e3
    td += d; tg += g
    print("%2d %-28s grid %-8s dur %6.2f us  gap before %6.2f us" % (k, names[first + k], rows[first + k].get("Grid_Size_X", "?"), d, g))
print("sum of durations %.1f us, of gaps %.1f us" % (td, tg))
PY
