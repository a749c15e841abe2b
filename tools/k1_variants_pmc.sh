#!/bin/bash
# PMC comparison of the K1 forward variants (DMH_K1_FWD_VARIANT): wave-instruction counts and the wave-cycle split.
set -u
OUT=gpurun_out/k1var
mkdir -p $OUT
export TMPDIR=/tmp
for v in ${1:-0 1 2}; do
  export DMH_K1_FWD_VARIANT=$v
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/v${v}_a -- python3 tools/prof_k1.py 2 > $OUT/v${v}_a.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/v${v}_b -- python3 tools/prof_k1.py 2 > $OUT/v${v}_b.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for v in (0, 1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("a", "b"):
        for f in glob.glob("gpurun_out/k1var/v%d_%s/*/*_counter_collection.csv" % (v, d)):
            for r in csv.DictReader(open(f)):
                if "photo_" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print("variant", v, k, {n: round(sum(x) / len(x) / 1e6, 2) for n, x in sorted(c.items())})
PY
