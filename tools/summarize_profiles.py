#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
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

ks10 = first(G + "k10_trace/*/*_kernel_stats.csv")
if ks10:
    shutil.copy(ks10, "profiles/%s_k10_kernel_stats.csv" % tag)
tabs10 = {}
for d in ["k10_pmc_a", "k10_fetch", "k10_write"]:
    t, meta = pmc(d)
    for k, v in t.items():
        if "wino_" in k:
            tabs10.setdefault(k, {}).update(v)
            tabs10[k]["_meta"] = meta[k]
if tabs10:
    cols = sorted({c for v in tabs10.values() for c in v if c != "_meta"})
    with open("profiles/%s_k10_pmc.csv" % tag, "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "VGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"] + cols)
        for k, v in tabs10.items():
            w.writerow([k] + list(v["_meta"]) + [round(v.get(c, 0)) for c in cols])
            if "conv_kernel" in k:
                rd, wr = 2 * v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024
                print("%-60s HBM traffic %.1f MB read + %.1f MB written; MFMA busy %.1f %% of CU-cycles" % (
                    k[:60], rd / 1e6, wr / 1e6,
                    100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 4.0 * v.get("SQ_BUSY_CU_CYCLES", 1))))

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
    print("bench timed region: %d steps, %.1f ms kernel time of %.1f ms wall" % (j["steps"], tot / 1e6, window / 1e6))
    for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:12]:
        print("  %-80s %8.2f ms %5d calls" % (n[:80], d / 1e6, c))
