#!/usr/bin/env python3
"""Average kernel durations out of a rocprofv3 result database (rocpd SQLite, the default output format):
    python tools/trace_avg.py gpurun_out/atk/atk_results.db roi_glue stem_conv_bwd"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
for pat in sys.argv[2:]:
    rows = list(db.execute("select name, count(*), avg(end - start), sum(end - start) from kernels where name like ? group by name",
                           ("%" + pat + "%",)))
    for n, c, a, s in rows:
        print("%-70s %6d calls  avg %8.2f us  total %9.3f ms" % (n.replace("(anonymous namespace)::", "").split("(")[0][:70], c, a / 1e3, s / 1e6))
