"""Turns one tools/final_r04.sh run (gpurun_out/<tag>_*) into the committed evidence: profiles/<tag>_* and the GENERATED tables of
DESIGN.md (between the `<!-- BEGIN GENERATED ... -->` / `<!-- END GENERATED -->` markers), so that no number in those tables is typed.
usage (container): python tools/summarize_r04.py <tag>"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, OUT = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
ALGO_S1 = 4 * 3840 * 2160 + 16 * 9988 + 16 * 1000


def newest(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)
    return f[-1] if f else None


def kname(n):
    n = n.split("(")[0]
    return n[5:] if n.startswith("void ") else n


def counters(path, full_launches_only=False):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, d in agg.items():
        if not k.startswith("swfr::"):
            continue
        out[k] = {}
        for c, v in d.items():
            if full_launches_only:                      # (the run also holds one-frame warm-up launches: keep the 8-frame ones, the large values)
                v = [x for x in v if x >= 0.5 * max(v)]
            out[k][c] = (sum(v) / len(v), len(v))
    return out


def kernel_us(path):
    out = {}
    if path and os.path.exists(path):
        for r in csv.DictReader(open(path)):
            if "swfr" in r["Name"]:
                out[kname(r["Name"])] = (float(r["AverageNs"]) / 1000.0, int(r["Calls"]), float(r["MaxNs"]) / 1000.0)
    return out


# ---- the S1 profile, the two shaded-kernel profiles: tools/profile_r02.py knows their layout
for t in (tag, tag + "_large", tag + "_magnified"):
    subprocess.call([sys.executable, os.path.join(ROOT, "tools", "profile_r02.py"), t], stdout=subprocess.DEVNULL)
# ---- plain copies
for name in ("bench_driver_command.json", "bench_s2.json", "bench_gloo_2ranks_one_gpu_rotate.json", "bench_gloo_2ranks_one_gpu_root.json", "config_bench.txt",
             "wg_timeline_s1.txt", "wg_timeline_s2.txt", "wg_timeline_config3.txt", "wg_timeline_config2h.txt", "short_run.txt", "blocks_timing.json", "tile_stats_s1.txt", "store_shape.txt", "sq_detail.txt", "batched_s1.json", "batched_s0.json",
             "lib_sha16.txt"):
    src = os.path.join(G, "%s_%s" % (tag, name))
    if os.path.exists(src) and os.path.getsize(src):
        if name.endswith(".json") and name.startswith("bench_gloo"):
            lines = [l for l in open(src).read().splitlines() if l.startswith('{"metric"')]       # (gloo prints its banner on stdout)
            open(os.path.join(OUT, "%s_%s" % (tag, name)), "w").write("\n".join(lines) + "\n")
        else:
            shutil.copyfile(src, os.path.join(OUT, "%s_%s" % (tag, name)))
# ---- the saturated runs (8 frames per launch): kernel stats as they are, PMC per launch and per frame
batched = {}
for w in ("s1", "s0"):
    ks = newest("%s_batched_%s_trace/*/*kernel_stats.csv" % (tag, w))
    if ks:
        shutil.copyfile(ks, os.path.join(OUT, "%s_batched_%s_kernel_stats.csv" % (tag, w)))
    fe, wr = newest("%s_batched_%s_fetch/*/*counter_collection.csv" % (tag, w)), newest("%s_batched_%s_write/*/*counter_collection.csv" % (tag, w))
    summ = {}
    if fe and wr:
        F, W = counters(fe, True), counters(wr, True)
        for k in sorted(set(F) | set(W)):
            fk, n = F.get(k, {}).get("FETCH_SIZE", (0.0, 0))
            wk, _ = W.get(k, {}).get("WRITE_SIZE", (0.0, 0))
            summ[k] = {"launches_averaged": n, "frames_per_launch": 8, "FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB": wk,
                       "hbm_bytes_per_launch": int(fk * 2 * 1024 + wk * 1024), "hbm_bytes_per_frame": int((fk * 2 * 1024 + wk * 1024) / 8)}
        json.dump(summ, open(os.path.join(OUT, "%s_batched_%s_pmc_summary.json" % (tag, w)), "w"), indent=1)
    batched[w] = (kernel_us(ks), summ)

# ---- the generated tables
L = []
# the commit of the kernels' sources in THIS tree, and the proof that the measured library was built from it: the GPU box has no .git,
# so the run leaves the first 16 hex digits of its libswfr.so's SHA-256, which must be the in-tree library's
commit = subprocess.check_output(["git", "log", "-1", "--format=%h", "--", "swf_renderer_amd/csrc"], cwd=ROOT).decode().strip()
dirty = subprocess.check_output(["git", "status", "--porcelain", "--", "swf_renderer_amd/csrc"], cwd=ROOT).decode().strip()
import hashlib
here = hashlib.sha256(open(os.path.join(ROOT, "swf_renderer_amd", "libswfr.so"), "rb").read()).hexdigest()[:16]
there = open(os.path.join(G, tag + "_lib_sha16.txt")).read().strip() if os.path.exists(os.path.join(G, tag + "_lib_sha16.txt")) else "?"
if here != there or dirty:
    print("WARNING: the measured library (%s) is not the in-tree build (%s)%s" % (there, here, " / csrc has uncommitted changes" if dirty else ""), file=sys.stderr)
commit = commit + (" (library sha256 %s...)" % there)
open(os.path.join(OUT, tag + "_commit.txt"), "w").write(commit + "\n")
L.append("Evidence tag `%s`, taken on the GPU box from the tree at commit `%s` (`profiles/%s_*`; generated by `tools/summarize_r04.py`, not typed)." % (tag, commit, tag))
L.append("")
iso, flt = kernel_us(os.path.join(OUT, tag + "_kernel_stats_one_frame_in_flight.csv")), kernel_us(os.path.join(OUT, tag + "_kernel_stats.csv"))
pmc = json.load(open(os.path.join(OUT, tag + "_pmc_summary.json"))) if os.path.exists(os.path.join(OUT, tag + "_pmc_summary.json")) else {}
sq = json.load(open(os.path.join(OUT, tag + "_sq_summary.json"))) if os.path.exists(os.path.join(OUT, tag + "_sq_summary.json")) else {}
bench = {}
if os.path.exists(os.path.join(OUT, tag + "_bench.json")):
    bench = json.loads([l for l in open(os.path.join(OUT, tag + "_bench.json")).read().splitlines() if l.startswith("{")][-1])
L.append("**S1, per kernel** (rocprofv3 `--kernel-trace --stats`; PMC `FETCH_SIZE` / `WRITE_SIZE` in their own passes, FETCH doubled for gfx950; SQ counters per wavefront):")
L.append("")
L.append("| kernel | one frame in flight, us (calls) | the bench's mode: two frames per launch, two launches in flight, us per launch | of the 8 TB/s roof (33 353 408 B / duration) | HBM bytes per launch (fetch + write) | wavefronts | VALU / SALU / LDS per wavefront |")
L.append("|---|---|---|---|---|---|---|")
tot_traffic, tot_us = 0, 0.0
for k in ("swfr::k2_bin_b", "swfr::k2_rows_b", "swfr::k2_tiles_solid_b"):
    us, calls, _ = iso.get(k, (0.0, 0, 0.0))
    us4 = flt.get(k, (0.0, 0, 0.0))[0]
    p = pmc.get(k, {})
    s = sq.get(k, {})
    tot_traffic += p.get("hbm_bytes_per_launch", 0)
    tot_us += us
    L.append("| `%s` | %.2f (%d) | %.2f | %.3f | %.2f MB (%.2f + %.2f) | %d | %.0f / %.0f / %.0f |" % (
        k[6:], us, calls, us4, (ALGO_S1 / (us * 1e-6) / 1e9 / 8000.0) if us else 0.0, p.get("hbm_bytes_per_launch", 0) / 1e6,
        p.get("fetch_bytes_corrected_x2", 0) / 1e6, p.get("write_bytes", 0) / 1e6, s.get("waves", 0), s.get("valu_per_wave", 0), s.get("salu_per_wave", 0), s.get("lds_per_wave", 0)))
L.append("| **frame** (sum) | %.2f | | %.3f | **%.2f MB = %.2f x algorithmic** | | |" % (tot_us, (ALGO_S1 / (tot_us * 1e-6) / 1e9 / 8000.0) if tot_us else 0.0, tot_traffic / 1e6, tot_traffic / ALGO_S1))
L.append("")
if bench:
    r = bench["roofline"]
    L.append("**S1 bench lines of the same build**: 300 steps: %.0f Mpx/s, `ms_per_step` %.4f, `roofline.kernel` %s `frac` %.4f, `frame_frac` %.4f, `step_frac` %.4f, batched %.4f ms per frame (`frac` %.4f), "
             "S0 `k2_tiles` %.1f us (`frac` %.3f; batched %.3f)." % (bench["value"], bench["ms_per_step"], r["kernel"], r["frac"], r.get("frame_frac", 0), r["step_frac"],
                                                                  r.get("batched", {}).get("ms_per_frame", 0), r.get("batched", {}).get("frac", 0), r.get("s0", {}).get("k2_tiles_ms", 0) * 1e3,
                                                                  r.get("s0", {}).get("frac", 0), r.get("s0", {}).get("batched", {}).get("frac", 0)))
drv = os.path.join(OUT, tag + "_bench_driver_command.json")
if os.path.exists(drv):
    d = json.loads([l for l in open(drv).read().splitlines() if l.startswith("{")][-1])
    L.append("The driver's command (`--steps 20 --warmup 5`): %.0f Mpx/s, `ms_per_step` %.4f, `step_frac` %.4f." % (d["value"], d["ms_per_step"], d["roofline"]["step_frac"]))
s2 = os.path.join(OUT, tag + "_bench_s2.json")
if os.path.exists(s2):
    d = json.loads([l for l in open(s2).read().splitlines() if l.startswith("{")][-1])
    L.append("S2 (8K, 99 909 edges, verified against the CPU oracle): %.0f Mpx/s, `ms_per_step` %.4f; one frame in flight k2_bin %.1f, k2_rows %.1f, k2_tiles %.1f us." % (
        d["value"], d["ms_per_step"], d["kernel_ms_per_frame"]["k2_bin"] * 1e3, d["kernel_ms_per_frame"]["k2_rows"] * 1e3, d["kernel_ms_per_frame"]["k2_tiles"] * 1e3))
L.append("")
L.append("**Saturated GPU** (8 frames per kernel launch, rocprofv3; per frame = the launch's duration / 8):")
L.append("")
L.append("| scene | ms per frame (HIP events) | k2_bin us / frame | k2_rows us / frame | k2_tiles us / frame | HBM bytes per frame (PMC) |")
L.append("|---|---|---|---|---|---|")
for w in ("s1", "s0"):
    ks, summ = batched[w]
    js = os.path.join(OUT, "%s_batched_%s.json" % (tag, w))
    ms = json.load(open(js))["ms_per_frame"] if os.path.exists(js) else 0.0
    per = lambda k: (ks.get(k, (0.0, 0, 0.0))[2] / 8.0)          # the full 8-frame launches are the longest ones of the run
    tb = sum(v["hbm_bytes_per_frame"] for k, v in summ.items() if any(n in k for n in ("k2_bin_b", "k2_rows_b", "k2_tiles_solid_b")))
    L.append("| %s | %.4f | %.1f | %.1f | %.1f | %.2f MB |" % (w.upper(), ms, per("swfr::k2_bin_b"), per("swfr::k2_rows_b"), per("swfr::k2_tiles_solid_b"), tb / 1e6))
L.append("")
for t, label in ((tag + "_large", "4096x4096 texture (about 1:1)"), (tag + "_magnified", "139x208 fixture texture, magnified")):
    ks = kernel_us(os.path.join(OUT, t + "_kernel_stats_one_frame_in_flight.csv"))
    p = json.load(open(os.path.join(OUT, t + "_pmc_summary.json"))) if os.path.exists(os.path.join(OUT, t + "_pmc_summary.json")) else {}
    s = json.load(open(os.path.join(OUT, t + "_sq_summary.json"))) if os.path.exists(os.path.join(OUT, t + "_sq_summary.json")) else {}
    k = "swfr::k2_tiles_bitmap_b"
    if k in ks:
        L.append("Config 4, %s: `k2_tiles_bitmap_b` %.2f us, HBM traffic %.1f MB per launch, %.0f VALU / %.0f SALU per wavefront = per strip (one frame in flight: one wavefront per strip)." % (
            label, ks[k][0], p.get(k, {}).get("hbm_bytes_per_launch", 0) / 1e6, s.get(k, {}).get("valu_per_wave", 0), s.get(k, {}).get("salu_per_wave", 0)))
cb = os.path.join(OUT, tag + "_config_bench.txt")
if os.path.exists(cb):
    L.append("")
    L.append("**Configs 2-4** (`tools/config_bench.py`; kernel columns: HIP-event intervals with four frames in flight, us):")
    L.append("")
    L.append("| configuration | edges / paths | resident frames/s | render frames/s | batch frames/s (device) | setup / rows / tiles us |")
    L.append("|---|---|---|---|---|---|")
    for l in open(cb):
        if l.startswith("{") and "kernel_us" in l:
            d = json.loads(l)
            L.append("| %s | %d / %d | %.0f | %.0f | %.0f | %.1f / %.1f / %.1f |" % (d["config"], d["edges"], d["paths"], d["device_frames_per_s"], d["full_path_frames_per_s"], d["batch_device_frames_per_s"],
                                                                               d["kernel_us"]["setup"], d["kernel_us"]["rows"], d["kernel_us"]["tiles"]))
bt = os.path.join(OUT, tag + "_blocks_timing.json")
if os.path.exists(bt):
    d = json.load(open(bt))
    L.append("")
    L.append("**One rank's block on an otherwise empty GPU** (`tools/pipeline_timing.py`: k2_bin + k2_rows + k2_tiles of the slowest sampled block, us):")
    L.append("")
    L.append("| ranks | S1 | S2 |")
    L.append("|---|---|---|")
    for n in (1, 2, 4, 8):
        cell = []
        for w in ("s1", "s2"):
            bl = d.get(w, {}).get("blocks_%d" % n, [])
            cell.append("%.1f" % max((b["k2_bin"] + b["k2_rows"] + b["k2_tiles"]) * 1e3 for b in bl) if bl else "-")
        L.append("| %d | %s | %s |" % (n, cell[0], cell[1]))
text = "\n".join(L) + "\n"
open(os.path.join(OUT, tag + "_tables.md"), "w").write(text)
dp = os.path.join(ROOT, "DESIGN.md")
ds = open(dp).read()
m = re.search(r"<!-- BEGIN GENERATED[^>]*-->\n.*?<!-- END GENERATED -->", ds, re.S)
if m:
    ds = ds[:m.start()] + "<!-- BEGIN GENERATED by tools/summarize_r04.py %s -->\n" % tag + text + "<!-- END GENERATED -->" + ds[m.end():]
    open(dp, "w").write(ds)
    print("DESIGN.md: generated block replaced")
print(text)
