#!/bin/bash
# same-box sweep of the stream-K cost model of K10 (DMH_SK_MODEL = us per chunk, per item epilogue, for the second launch, margin)
for m in "3.05,4,6,0.92" "2.7,2.2,6,0.92" "2.7,2.2,8,0.92" "2.7,2.2,6,0.97" "2.7,2.2,6,0.85" "2.7,2.2,4,0.92" "3.05,4,6,0.92"; do
  echo -n "$m  "; DMH_SK_MODEL=$m python3 bench.py --no_cpu_baseline 2>/dev/null | python3 tools/parse_bench.py /dev/stdin | head -1
done
