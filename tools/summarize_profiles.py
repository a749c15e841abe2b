#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = "gpurun_out/prof_%s/" % tag
os.makedirs("profiles", exist_ok=True)


def first(pattern):
    """Newest match (gpurun merges new runs into the same scratch directory next to older ones)."""
    m = sorted(glob.glob(pattern), key=os.path.getmtime)
    return m[-1] if m else None


ks = first(G + "k1_trace/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, "profiles/%s_k1k2_kernel_stats.csv" % tag)


def pmc(d):
    f = first(G + d + "/*/*_counter_collection.csv")
    out, meta = {}, {}
    if not f:
        return out, meta
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[r["Kernel_Name"]] = (r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in acc.items()}, meta


tabs = {}
for d in ["k1_pmc_a", "k1_pmc_b", "k1_fetch", "k1_write", "k1_tcc"]:
    t, meta = pmc(d)
    for k, v in t.items():
        if "anonymous" in k and "at::native" not in k:
            tabs.setdefault(k, {}).update(v)
            tabs[k]["_meta"] = meta[k]
cols = sorted({c for v in tabs.values() for c in v if c != "_meta"})
if tabs:    # the shape tools/prof_k1.py runs: bench.py uses the traffic figures only for this shape
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench          # the source hash that ties these counters to the kernel they were taken on (bench.measured_traffic)
    json.dump({"B": 32, "H": 320, "W": 1024, "scales": 4, "tool": "tools/prof_k1.py", "k1_source_sha256": bench.k1_source_hash(),
               "k1_sources": list(bench.K1_SOURCES)}, open("profiles/%s_k1k2_shape.json" % tag, "w"))
with open("profiles/%s_k1k2_pmc.csv" % tag, "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "VGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"] + cols)
    for k, v in tabs.items():
        w.writerow([k] + list(v["_meta"]) + [round(v.get(c, 0)) for c in cols])
# FETCH_SIZE calibration for THIS access pattern (one dword per lane, coalesced rows): smooth_fwd_kernel reads the
# disparity and colour pyramids exactly once = 16 B per low-resolution pixel (MI355X_MICROARCH.md: "other access widths
# are uncalibrated: calibrate on a known byte count in your own access pattern"); the x2 of the guide applies to
# 16-byte-per-lane streaming reads only.
for k, v in tabs.items():
    if "photo" in k or "smooth" in k:
        rd, wr = v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024
        print("%-50s FETCH_SIZE %.1f MB + WRITE_SIZE %.1f MB = %.1f MB per launch (dword loads: FETCH_SIZE taken 1:1)" % (
            k[:50], rd / 1e6, wr / 1e6, (rd + wr) / 1e6))

ks3 = first(G + "k3_trace/*/*_kernel_stats.csv")
if ks3:
    shutil.copy(ks3, "profiles/%s_k3_kernel_stats.csv" % tag)
tabs3 = {}
for d in ["k3_pmc_a", "k3_fetch", "k3_write"]:
    t, meta = pmc(d)
    for k, v in t.items():
        if "paste_" in k:
            tabs3.setdefault(k, {}).update(v)
            tabs3[k]["_meta"] = meta[k]
if tabs3:
    cols = sorted({c for v in tabs3.values() for c in v if c != "_meta"})
    with open("profiles/%s_k3_pmc.csv" % tag, "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "VGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"] + cols)
        for k, v in tabs3.items():
            w.writerow([k] + list(v["_meta"]) + [round(v.get(c, 0)) for c in cols])
            print("%-50s FETCH_SIZE %.1f MB WRITE_SIZE %.1f MB" % (k[:50], v.get("FETCH_SIZE", 0) * 1024 / 1e6,
                                                                  v.get("WRITE_SIZE", 0) * 1024 / 1e6))

for kk in ("k14", "k15"):
    f = first(G + kk + "_trace/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, "profiles/%s_%s_kernel_stats.csv" % (tag, kk))
        for r in csv.DictReader(open(f)):
            if "anonymous" in r["Name"]:
                print("%s: %-70s %4s calls avg %.1f us" % (kk, r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))

# K10 at three layer shapes: one row per shape in profiles/<tag>_k10_pmc.csv, kernel-trace averages alongside
SHAPES10 = {"l1": (64, 64, 80, 256, 1, 12), "l3": (256, 256, 20, 64, 1, 12), "up21": (128, 64, 80, 256, 0, 12)}
rows10 = []
for tk, (C_, K_, Ho_, Wo_, pad_, B_) in SHAPES10.items():
    tr = first(G + "k10_%s_trace/*/*_kernel_stats.csv" % tk)
    avg_us = None
    if tr:
        for r in csv.DictReader(open(tr)):
            if "wino_conv_kernel" in r["Name"]:
                avg_us = float(r["AverageNs"]) / 1e3
    vals = {}
    for d in ["k10_%s_pmc_a" % tk, "k10_%s_fetch" % tk, "k10_%s_write" % tk]:
        t, meta = pmc(d)
        for kname, v in t.items():
            if "wino_conv_kernel" in kname:
                vals.update(v)
    if avg_us is None and not vals:
        continue
    direct = 2.0 * 9 * C_ * K_ * Ho_ * Wo_ * B_
    algo = 4.0 * (B_ * C_ * (Ho_ + 2 - 2 * pad_) * (Wo_ + 2 - 2 * pad_) + B_ * K_ * Ho_ * Wo_ + 16 * C_ * K_)
    row = {"shape": "%s C=%d K=%d %dx%d pad=%d B=%d" % (tk, C_, K_, Ho_, Wo_, pad_, B_), "avg_us": avg_us,
           "direct_GFLOP": direct / 1e9, "TFLOPs_direct_equivalent": direct / (avg_us * 1e-6) / 1e12 if avg_us else None,
           "TFLOPs_mfma_issued": direct / 2.25 / (avg_us * 1e-6) / 1e12 if avg_us else None,
           "algorithmic_MB": algo / 1e6,
           # 16-byte-per-lane streaming reads (LDS-DMA, float4 rows): FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)
           "fetch_MB_x2": 2 * vals.get("FETCH_SIZE", 0) * 1024 / 1e6, "write_MB": vals.get("WRITE_SIZE", 0) * 1024 / 1e6}
    for c in ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES",
              "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        row[c] = vals.get(c)
    if vals.get("SQ_BUSY_CU_CYCLES"):
        row["mfma_busy_frac"] = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * vals["SQ_BUSY_CU_CYCLES"])
    rows10.append(row)
    print("k10 %s: %s" % (tk, {k_: (round(v_, 3) if isinstance(v_, float) else v_) for k_, v_ in row.items()}))
if rows10:
    cols = list(rows10[0].keys())
    with open("profiles/%s_k10_pmc.csv" % tag, "w") as f:
        w = csv.writer(f)
        w.writerow(cols)
        for r in rows10:
            w.writerow([r.get(c) for c in cols])

bs = first(G + "bench_trace/*/*_kernel_stats.csv")
bt = first(G + "bench_trace/*/*_kernel_trace.csv")
if bs and bt:
    shutil.copy(bs, "profiles/%s_bench_kernel_stats_full_run.csv" % tag)
    line = [l for l in open(G + "bench.json") if l.startswith("{")][-1]
    open("profiles/%s_bench.json" % tag, "w").write(line)
    j = json.loads(line)
    window = j["ms_per_step"] * j["steps"] * 1e6 * 1.005
    rows = list(csv.DictReader(open(bt)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    ev = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
                for r in rows)
    tend = ev[-1][0]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, d, n in ev:
        if s > tend - window:
            agg[n][0] += d
            agg[n][1] += 1
    tot = sum(v[0] for v in agg.values())
    with open("profiles/%s_bench_timed_region.csv" % tag, "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0]):
            w.writerow([n, c, d, round(d / c, 1), round(100 * d / tot, 3)])
    # the same per kernel family (the tables of DESIGN.md section 6 / profiles/README.md), ms per step
    def family(n):
        for key, fam in (("roi_", "K19 window glue / cost / crop / paste"), ("stem_bwd_win", "K19 window glue / cost / crop / paste"),
                         ("zero_fill", "K10 wino_conv (+ filter transforms)"), ("ssim_", "layers surface"), ("edge_smooth", "layers surface"),
                         ("wino32", "K17 wino32_conv"), ("wino_wrw", "K18 wino_wrw"), ("down_wrw", "K20 / K21 strided + stem weight gradients"),
                         ("stem_wrw", "K20 / K21 strided + stem weight gradients"), ("wino_", "K10 wino_conv (+ filter transforms)"),
                         ("small_conv", "K11 small_conv"), ("small_wrw", "K16 small_wrw"), ("head_wrw", "K13 head weight gradient"),
                         ("down_conv", "K15 down_conv"), ("down_weight_image", "K15 down_conv"), ("elu_pad", "K7 decoder glue"), ("up_cat_pad", "K7 decoder glue"),
                         ("bn_", "K9 encoder glue / BatchNorm"), ("channel_sum", "K9 encoder glue / BatchNorm"),
                         ("stem", "stem (K14 / K12 / K9 stem glue)"), ("head_", "K13 disparity heads"),
                         ("photo_", "K1-K6 loss + attack"), ("smooth_", "K1-K6 loss + attack"), ("finalize", "K1-K6 loss + attack"),
                         ("paste_", "K1-K6 loss + attack"), ("sq_mean", "K1-K6 loss + attack"), ("l0_", "K1-K6 loss + attack"),
                         ("pgd_", "K1-K6 loss + attack"), ("avg_pyramid", "K1-K6 loss + attack"), ("gt_depth", "K1-K6 loss + attack"), ("unpack_sel", "K1-K6 loss + attack"), ("depth_err", "K1-K6 loss + attack"),
                         ("igemm", "MIOpen"), ("miopen", "MIOpen"), ("MIOpen", "MIOpen"), ("batched_transpose", "MIOpen"),
                         ("Sp3Asm", "MIOpen"), ("ck::", "MIOpen"), ("gridwise", "MIOpen"),
                         ("at::native", "ATen / runtime"), ("rocclr", "ATen / runtime")):
            if key in n:
                return fam
        return "other"
    fam = collections.defaultdict(lambda: [0, 0])
    for n, (d, c) in agg.items():
        fam[family(n)][0] += d
        fam[family(n)][1] += c
    with open("profiles/%s_bench_families.csv" % tag, "w") as f:
        w = csv.writer(f)
        w.writerow(["Family", "ms_per_step", "launches_per_step", "Percentage"])
        for n, (d, c) in sorted(fam.items(), key=lambda x: -x[1][0]):
            w.writerow([n, round(d / 1e6 / j["steps"], 3), round(c / j["steps"], 1), round(100 * d / tot, 2)])
    print("bench timed region: %d steps, %.1f ms kernel time of %.1f ms wall" % (j["steps"], tot / 1e6, window / 1e6))
    for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:12]:
        print("  %-80s %8.2f ms %5d calls" % (n[:80], d / 1e6, c))
