#!/usr/bin/env python3
"""Instruction-issue counters of the step's kernels (rocprofv3 PMC; run on the GPU box through
gpurun):  python3 tools/pmc_insts.py <tag> <config> [bench args]

The SQ block has 8 counter slots per pass (MI355X_MICROARCH.md, rocprofv3 PMC slots), so the
counters are collected in passes of <= 8 over the SAME command `python3 bench.py --config <config>
...` (the program directly after `--`: no shell or env hop between the profiler and the program).
Counter names are checked against `rocprofv3 -L` first: a name this ROCm does not know is left
out (and listed in the summary) instead of failing the pass.

Output: gpurun_out/pmc_insts_<tag>_<config>/{pass*/, summary.json}; summary.json is what
profiles/<round>_pmc_insts_<config>.json keeps.  Per kernel and launch: the raw sums, and

  mfma_busy_frac_of_busy_cu_simd_cycles   SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)
  mfma_busy_frac_of_chip                  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
  wait_any / wait_inst_any / active_inst_* as fractions of SQ_WAVE_CYCLES, instructions per wave

SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles (same guide, cycle-constants table)."""
import csv
import glob
import json
import os
import subprocess
import sys

WANT = [
    # pass 1: what is issued
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_SALU",
     "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"],
    # pass 2: where the wave cycles go
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"],
    # pass 3: the other instruction classes' active cycles
    ["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC",
     "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT"],
]
KERNELS = ("k_linear_big", "k_linear", "k_latent", "k_wgrad_big_reduce", "k_wgrad_big", "k_wgrad",
           "k_adam", "k_finalize", "k_fused", "k_wfrag", "k_partials_fold")


def known_counters():
    p = subprocess.run(["rocprofv3", "-L"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return p.stdout


def short_name(name):
    for k in KERNELS:
        if k + "(" in name or k + "<" in name:
            if k == "k_fused" or k == "k_latent" or k == "k_wgrad":
                i = name.find(k + "<")
                if i >= 0:      # keep the instantiation: k_fused<4>
                    j = name.find(">", i)
                    return name[i:j + 1].replace(" ", "")
            return k
    return None


def derive(kernels):
    """Fractions from the raw per-launch sums.  Units (checked on k_wfrag, whose four waves per
    workgroup live exactly as long as their CU is busy: SQ_WAVE_CYCLES == SQ_BUSY_CU_CYCLES there):
    SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves,
    SQ_BUSY_CU_CYCLES is cycles summed over CUs, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs,
    GRBM_GUI_ACTIVE cycles summed over the 8 XCDs."""
    for k, rec in kernels.items():
        p = rec["per_launch"]
        wc = p.get("SQ_WAVE_CYCLES")
        d = {}
        if wc:
            for c, name in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"),
                            ("SQ_ACTIVE_INST_ANY", "active_inst_any"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu"),
                            ("SQ_ACTIVE_INST_LDS", "active_inst_lds"), ("SQ_ACTIVE_INST_SCA", "active_inst_scalar"),
                            ("SQ_ACTIVE_INST_VMEM", "active_inst_vmem"), ("SQ_ACTIVE_INST_MISC", "active_inst_misc")):
                if c in p:
                    d[name + "_frac_of_wave_cycles"] = round(p[c] / wc, 4)
            if p.get("SQ_WAVES"):
                d["cycles_per_wave"] = round(4.0 * wc / p["SQ_WAVES"], 1)
        busy_cu = p.get("SQ_BUSY_CU_CYCLES")
        mfma = p.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if mfma is not None and busy_cu:
            # of the SIMD-cycles of the CUs that held a workgroup, while they held one
            d["mfma_busy_frac_of_busy_cu_simd_cycles"] = round(mfma / (4.0 * busy_cu), 4)
        gui = p.get("GRBM_GUI_ACTIVE")
        if mfma is not None and gui:
            # of ALL 256 CUs x 4 SIMDs over the kernel's duration: the chip-level MFMA utilisation
            d["kernel_cycles"] = round(gui / 8.0, 1)
            d["mfma_busy_frac_of_chip"] = round(mfma / (gui / 8.0 * 256 * 4), 5)
        if p.get("SQ_WAVES"):
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS",
                      "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if c in p:
                    d[c[3:].lower() + "_per_wave"] = round(p[c] / p["SQ_WAVES"], 1)
        rec["derived"] = d


def main():
    if sys.argv[1] == "--rederive":      # an existing summary.json, in place
        with open(sys.argv[2]) as f:
            doc = json.load(f)
        derive(doc["kernels"])
        with open(sys.argv[2], "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
        return
    tag, cfg = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:] or ["--steps", "100", "--warmup", "20", "--settle", "0", "--no-cpu-baseline",
                             "--no-roofline", "--quick"]
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = os.path.join(root, "gpurun_out", "pmc_insts_%s_%s" % (tag, cfg))
    os.makedirs(out, exist_ok=True)
    listing = known_counters()
    with open(os.path.join(out, "counters_known.txt"), "w") as f:
        f.write(listing)
    missing, tot, cnt = [], {}, {}
    env = dict(os.environ, TMPDIR="/tmp")
    for i, names in enumerate(WANT):
        have = [n for n in names if n in listing]
        missing += [n for n in names if n not in listing]
        if not have:
            continue
        d = os.path.join(out, "pass%d" % i)
        cmd = ["rocprofv3", "--pmc"] + have + ["--output-format", "csv", "-d", d, "-o", "run", "--",
                                               "python3", os.path.join(root, "bench.py"), "--config", cfg] + extra
        with open(os.path.join(out, "pass%d.log" % i), "w") as log:
            rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, cwd="/tmp", env=env).returncode
        print("pass %d (%s): rc %d" % (i, " ".join(have), rc), flush=True)
        if rc:
            sys.exit(rc)
        for fcsv in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(fcsv, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short_name(row["Kernel_Name"])
                    if k is None:
                        continue
                    key = (k, row["Counter_Name"])
                    tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"])
                    cnt[key] = cnt.get(key, 0) + 1
    kernels = {}
    for (k, c), v in tot.items():
        kernels.setdefault(k, {"per_launch": {}, "launches": 0})
        kernels[k]["per_launch"][c] = round(v / cnt[(k, c)], 1)
        kernels[k]["launches"] = max(kernels[k]["launches"], cnt[(k, c)])
    derive(kernels)
    doc = {"note": "rocprofv3 --pmc over `python3 bench.py --config %s %s` (tools/pmc_insts.py, build %s); "
                   "averages per launch, summed over all shader engines / XCDs as rocprofv3 reports them"
                   % (cfg, " ".join(extra), tag),
           "config": cfg, "counters_unknown_to_this_rocm": missing, "kernels": kernels}
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
