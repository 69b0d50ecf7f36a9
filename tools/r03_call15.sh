#!/bin/bash
# the four-row decoder stage from the LDS copy of Wd: suite, same-box A/B, general chain at 65,536 rows
set -o pipefail
mkdir -p gpurun_out
ls -la 2022_cambroise_interpret_multivae_amd/*.so || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/t_r03g.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/t_r03g.log
[ $rc -eq 0 ] || exit $rc
bash tools/r03_variants.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/variants_q3.txt
bash tools/r03_variants.sh 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/variants_q3.txt
python tools/general_big.py 65536 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MOPOE_FORCE_GENERAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_general_big -o run -- python3 tools/general_big.py 65536 > gpurun_out/general_big.log 2>&1; echo "general rc=$?"; grep -v amdgpu.ids gpurun_out/general_big.log | grep -v "^W2026\|^I2026\|^E2026" | tail -5
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_general_big/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:16]:
    print("%-90s %5s %10.1f us" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
