#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (one gpurun call):
#   bash tools/collect_profiles.sh r01        -> gpurun_out/prof_r01/*  (then: python tools/summarize_profiles.py r01)
# PMC passes are separate runs (gpurun refuses --pmc combined with trace domains); FETCH_SIZE and WRITE_SIZE do not
# fit one pass (TCC has 4 slots: 3 + 2).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
K1="python3 tools/prof_k1.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1_trace -- $K1 5 > $OUT/k1_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/k1_pmc_a -- $K1 2 > $OUT/k1_pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/k1_pmc_b -- $K1 2 > $OUT/k1_pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/k1_fetch -- $K1 2 > $OUT/k1_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/k1_write -- $K1 2 > $OUT/k1_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/k1_tcc -- $K1 2 > $OUT/k1_tcc.log 2>&1
echo "k1 passes done"
# K10 (Winograd-MFMA convolution) at the encoder layer3 shape of the attack (B=12, 256 -> 256 channels, 20x64)
K10="python3 tools/wino_prof.py 256 256 20 64 1 12"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k10_trace -- $K10 5 > $OUT/k10_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/k10_pmc_a -- $K10 2 > $OUT/k10_pmc_a.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/k10_fetch -- $K10 2 > $OUT/k10_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/k10_write -- $K10 2 > $OUT/k10_write.log 2>&1
echo "k10 passes done"
if [ "${2:-bench}" = "bench" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err
  tail -1 $OUT/bench.json
fi
