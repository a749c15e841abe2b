#!/bin/bash
# development tool: one environment switch, two values, over the BASELINE configs on ONE box (alternating)
#   bash tools/ab_configs.sh DMH_WINO_MIN_ITEMS 200 64 "3 4 5"
VAR=$1; A=$2; B=$3; CFGS=${4:-"2 3 4 5"}
for c in $CFGS; do
  for v in $A $B $A $B; do
    line=$(env $VAR=$v python bench.py --config $c --no_cpu_baseline 2>/dev/null | tail -1)
    echo "config $c $VAR=$v  $(python -c "import sys,json; d=json.loads(sys.argv[1]); print(d['value'], 'images/s', d['ms_per_step'], 'ms')" "$line")"
  done
done
