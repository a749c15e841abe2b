set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_k1b
mkdir -p $OUT
K1="python3 tools/prof_k1.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $K1 5 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_a -- $K1 2 > $OUT/pmc_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_b -- $K1 2 > $OUT/pmc_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $K1 2 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $K1 2 > $OUT/write.log 2>&1
find $OUT -name "*.csv" | head -30
