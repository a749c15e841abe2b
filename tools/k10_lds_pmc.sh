#!/bin/bash
# LDS counters of K10 at layer3's attack shape (round 6): separate --pmc passes, no trace domains.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/k10_lds
mkdir -p $OUT
K10="python3 tools/wino_prof.py 256 256 20 64 1 12 2"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- $K10 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -- $K10 > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d $OUT/c -- $K10 > $OUT/c.log 2>&1
find $OUT -name "*counter_collection.csv" | while read f; do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "wino_conv_kernel" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s %16.0f per launch" % (c, v / max(1, cnt[(k, c)])))
PY
done
