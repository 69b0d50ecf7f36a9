"""Reduce the rocprofv3 CSVs of tools/profile_round.sh to the two files profiles/ keeps:
<dir>/kernel_stats.csv (the --stats table, our kernels + everything else on the GPU) and
<dir>/pmc_traffic.json (per-kernel HBM bytes per launch).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KB, collected in separate passes; on gfx950 FETCH_SIZE tallies a 128-byte
request as 64 bytes for wide coalesced reads, so reads are doubled ("corrected", an upper
estimate for our 8-16 B/lane reads); WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys


KERNELS = ("k_linear_big", "k_linear", "k_latent", "k_wgrad_big_reduce", "k_wgrad_big", "k_wgrad",
           "k_adam", "k_finalize", "k_fused", "k_wfrag", "k_xgmi")


def per_kernel(path, counter):
    tot, cnt = {}, {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                short = next((k for k in KERNELS if k + "(" in name or k + "<" in name), None)
                if short is None:
                    continue
                tot[short] = tot.get(short, 0.0) + float(row["Counter_Value"])
                cnt[short] = cnt.get(short, 0) + 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main(out, tag):
    fetch, n_f = per_kernel(os.path.join(out, "pmc_fetch"), "FETCH_SIZE")
    write, n_w = per_kernel(os.path.join(out, "pmc_write"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        kernels[k] = {
            "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
            "launches": n_f.get(k, n_w.get(k, 0)),
            "hbm_bytes_raw": int((f + w) * 1024),
            "hbm_bytes_corrected": int((2 * f + w) * 1024),
        }
    doc = {
        "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over "
                "`python3 bench.py` (tools/profile_round.sh, build %s); averages per "
                "launch; counter unit KB; hbm_bytes_corrected = (2*FETCH_SIZE + "
                "WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 tallies 128-B read "
                "requests as 64 B for wide coalesced reads; upper estimate here)." % tag,
        "kernels": kernels,
    }
    with open(os.path.join(out, "pmc_traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        with open(stats[0]) as src, open(os.path.join(out, "kernel_stats.csv"), "w") as dst:
            dst.write(src.read())
    print(json.dumps(kernels, indent=1))
    if stats:
        with open(stats[0]) as fh:
            for i, line in enumerate(fh):
                if i < 6:
                    print(line.rstrip())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
