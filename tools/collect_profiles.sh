#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (one gpurun call):
#   bash tools/collect_profiles.sh r02        -> gpurun_out/prof_r02/*  (then: python tools/summarize_profiles.py r02)
# PMC passes are separate runs (gpurun refuses --pmc combined with trace domains); FETCH_SIZE and WRITE_SIZE do not
# fit one pass (TCC has 4 slots: 3 + 2).
set -u
TAG=${1:-r03}
MODE=${2:-bench}   # the K10 loop below re-sets the positional parameters
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$MODE" = "benchonly" ]; then     # only the whole-step trace (after a change that leaves the per-kernel passes as they are)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err
  tail -1 $OUT/bench.json | cut -c1-200
  exit 0
fi
K1="python3 tools/prof_k1.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k1_trace -- $K1 5 > $OUT/k1_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/k1_pmc_a -- $K1 2 > $OUT/k1_pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/k1_pmc_b -- $K1 2 > $OUT/k1_pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/k1_fetch -- $K1 2 > $OUT/k1_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/k1_write -- $K1 2 > $OUT/k1_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/k1_tcc -- $K1 2 > $OUT/k1_tcc.log 2>&1
echo "k1 passes done"
# K3 (EOT paste) at the attack shape: 12 scenes 375x1242 -> 320x1024
K3="python3 tools/prof_k3.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k3_trace -- $K3 5 > $OUT/k3_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/k3_fetch -- $K3 2 > $OUT/k3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/k3_write -- $K3 2 > $OUT/k3_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/k3_pmc_a -- $K3 2 > $OUT/k3_pmc_a.log 2>&1
echo "k3 passes done"
# K14 (stem) and K15 (down-sampling block convolutions) against MIOpen at the attack shapes
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k15_trace -- python3 tools/down_bench.py > $OUT/k15_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k14_trace -- python3 tools/stem_bench.py > $OUT/k14_trace.log 2>&1
echo "k14/k15 passes done"
# K10 (Winograd-MFMA convolution) at three layer shapes of the attack pass (12 scenes): layer1 64->64 @80x256, layer3 256->256
# @20x64, upconv(2,1) 128->64 @80x256 (pad 0 on the pre-padded tensor).  Kernel trace + MFMA / traffic counters, one shape per
# run so that per-kernel averages are per shape.
for shp in "l1 64 64 80 256 1 12" "l3 256 256 20 64 1 12" "up21 128 64 80 256 0 12"; do
  set -- $shp; tagk=$1; shift
  K10="python3 tools/wino_prof.py $*"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k10_${tagk}_trace -- $K10 5 > $OUT/k10_${tagk}_trace.log 2>&1
  rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/k10_${tagk}_pmc_a -- $K10 2 > $OUT/k10_${tagk}_pmc_a.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/k10_${tagk}_fetch -- $K10 2 > $OUT/k10_${tagk}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/k10_${tagk}_write -- $K10 2 > $OUT/k10_${tagk}_write.log 2>&1
done
echo "k10 passes done"
if [ "$MODE" = "bench" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 bench.py --steps 3 --warmup 1 --no_cpu_baseline > $OUT/bench.json 2> $OUT/bench.err
  tail -1 $OUT/bench.json
fi
