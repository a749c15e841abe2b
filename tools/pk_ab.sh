#!/bin/bash
# Same-box A/B of the packed-add transforms (round 6): var/libdmh_pk0.so = the library with DMH_WINO_PK=0 / DMH_W32_PK=0 objects
# (built by hand, see profiles/README.md), the tree's library = the shipped form.  K17 checksums + times, then the whole step twice each.
set -e
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== K17, packed"; python3 tools/wino32_bench.py 12 20 nomiopen
echo "== K17, scalar"; DMH_HIP_LIB=var/libdmh_pk0.so python3 tools/wino32_bench.py 12 20 nomiopen
for r in 1 2; do
  echo "== step, scalar";  DMH_HIP_LIB=var/libdmh_pk0.so python3 bench.py --steps 10 --warmup 3 --no_cpu_baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"
  echo "== step, packed"; python3 bench.py --steps 10 --warmup 3 --no_cpu_baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"
done
