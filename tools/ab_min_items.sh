#!/bin/bash
# development tool: K10's fill threshold (DMH_WINO_MIN_ITEMS) at the strong-scaling shares -- bench.py per setting, one box
#   bash tools/ab_min_items.sh "4 2" "200 75 50 35" 2
CFG=${1:-"4 2"}; VALUES=${2:-"200 100 50 25"}; ROUNDS=${3:-1}
set -- $CFG
for r in $(seq 1 $ROUNDS); do
  for mi in $VALUES; do
    line=$(DMH_WINO_MIN_ITEMS=$mi python bench.py --batch_size $1 --atk_scenes $2 --steps 10 --warmup 3 --no_cpu_baseline 2>/dev/null | tail -1)
    echo "batch $1 scenes $2 MIN_ITEMS=$mi  $(python -c "import sys,json; d=json.loads(sys.argv[1]); print(d['ms_per_step'], 'ms')" "$line")"
  done
done
