"""Summarises a tools/profile_r02.sh run (gpurun_out/<tag>_*) into profiles/<tag>_*: the kernel stats tables as they are, HBM bytes
per launch per kernel from the two PMC passes (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are KiB, separate passes, and on
gfx950 FETCH_SIZE counts half of the fetched bytes -> doubled), SQ counters per wavefront, and the bench line.
usage: python tools/profile_r02.py <tag>"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
OUT = os.path.join(ROOT, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)      # (gpurun_out keeps earlier runs of a tag: the newest)
    return f[-1] if f else None


def kname(n):
    n = n.split("(")[0]
    return n[5:] if n.startswith("void ") else n


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if k.startswith("swfr::")}


for sub, name in (("trace", "kernel_stats"), ("trace1", "kernel_stats_one_frame_in_flight")):
    f = one("%s_%s/*/*kernel_stats.csv" % (tag, sub))
    if f:
        shutil.copyfile(f, os.path.join(OUT, "%s_%s.csv" % (tag, name)))
fe, wr, sq = one("%s_fetch/*/*counter_collection.csv" % tag), one("%s_write/*/*counter_collection.csv" % tag), one("%s_sq/*/*counter_collection.csv" % tag)
summary = {}
if fe and wr:
    F, Wc = counters(fe), counters(wr)
    for k in sorted(set(F) | set(Wc)):
        fk, wk = F.get(k, {}).get("FETCH_SIZE", 0.0), Wc.get(k, {}).get("WRITE_SIZE", 0.0)
        summary[k] = {"FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB": wk, "fetch_bytes_corrected_x2": int(fk * 2 * 1024), "write_bytes": int(wk * 1024),
                      "hbm_bytes_per_launch": int(fk * 2 * 1024 + wk * 1024)}
    json.dump(summary, open(os.path.join(OUT, "%s_pmc_summary.json" % tag), "w"), indent=1)
    kt = [v for k, v in summary.items() if k.startswith("swfr::k2_tiles") or k.startswith("swfr::k_tiles")]
    if kt:
        json.dump(max(kt, key=lambda v: v["hbm_bytes_per_launch"]), open(os.path.join(OUT, "%s_pmc_k_tiles.json" % tag), "w"), indent=1)
if sq:
    S = counters(sq)
    out = {}
    for k, m in S.items():
        w = max(m.get("SQ_WAVES", 1), 1)
        out[k] = {"waves": w, "valu_per_wave": m.get("SQ_INSTS_VALU", 0) / w, "salu_per_wave": m.get("SQ_INSTS_SALU", 0) / w, "lds_per_wave": m.get("SQ_INSTS_LDS", 0) / w,
                  "wave_cycles_per_wave_x4": 4 * m.get("SQ_WAVE_CYCLES", 0) / w, "issue_active_frac": m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
                  "waiting_frac": m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)}
    json.dump(out, open(os.path.join(OUT, "%s_sq_summary.json" % tag), "w"), indent=1)
b = os.path.join(G, "%s_bench.json" % tag)
if os.path.exists(b):
    shutil.copyfile(b, os.path.join(OUT, "%s_bench.json" % tag))
print(json.dumps(summary, indent=1)[:3000])
for sub in ("kernel_stats", "kernel_stats_one_frame_in_flight"):
    f = os.path.join(OUT, "%s_%s.csv" % (tag, sub))
    if os.path.exists(f):
        print(sub)
        for r in csv.DictReader(open(f)):
            if "swfr" in r["Name"]:
                print("  %-34s calls %5s avg %8.2f us" % (kname(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1000))
