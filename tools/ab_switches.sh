#!/bin/bash
# A/B of environment switches on ONE box (development tool): bench.py at the headline configuration once per setting,
# alternating, so that box-to-box spread (~1 %) does not hide a 0.5 % effect.
#   bash tools/ab_switches.sh "DMH_WINO_MIN_FILL=0.6 DMH_ZEROS_ONCE=0" "DMH_WINO_MIN_FILL=0.5 DMH_ZEROS_ONCE=1" [rounds=2]
set -e
A="$1"; B="$2"; R="${3:-2}"
for r in $(seq 1 "$R"); do
  for cfg in "$A" "$B"; do
    line=$(env $cfg python bench.py --steps 10 --warmup 3 --no_cpu_baseline 2>/dev/null | tail -1)
    echo "$cfg  $(python -c "import sys,json; d=json.loads(sys.argv[1]); print(d['value'], 'images/s', d['ms_per_step'], 'ms')" "$line")"
  done
done
