"""Copies the rocprofv3 summaries of a gpurun profiling call into profiles/ (tracked) and derives the
per-launch HBM traffic of k_tiles from the PMC passes.

usage: python tools/summarize_profile.py <tag> <kernel_stats.csv> <fetch counter csv> <write counter csv>

Counter handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate --pmc
passes, both are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes fetched, so it is doubled
(calibrated there for wide streaming reads; our reads are narrower, so the figure is an upper-side estimate).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0]
            if name.startswith("void "):           # template instances are reported with their return type
                name = name[5:]
            agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copyfile(stats, os.path.join(out, "%s_kernel_stats.csv" % tag))
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    summary = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("swfr::"):
            continue
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        summary[k] = {"FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB": wk, "fetch_bytes_corrected_x2": int(fk * 2 * 1024),
                      "write_bytes": int(wk * 1024), "hbm_bytes_per_launch": int(fk * 2 * 1024 + wk * 1024)}
    json.dump(summary, open(os.path.join(out, "%s_pmc_summary.json" % tag), "w"), indent=1)
    kt = [v for k, v in summary.items() if k.startswith("swfr::k_tiles")]
    kt = max(kt, key=lambda v: v["hbm_bytes_per_launch"]) if kt else None
    if kt:
        json.dump(kt, open(os.path.join(out, "%s_pmc_k_tiles.json" % tag), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
