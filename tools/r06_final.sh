#!/bin/bash
# Round-6 closing measurements on the GPU box (one gpurun call): the four BASELINE workloads, the strong-scaling rank workloads,
# the 2-rank bench line on one GPU (plumbing evidence) and the trainer's own step log.  Outputs under gpurun_out/r06_final/.
set -u
OUT=gpurun_out/r06_final
mkdir -p $OUT
python bench.py > $OUT/cfg2.json 2> $OUT/cfg2.err
echo "cfg2 done"; tail -c 300 $OUT/cfg2.err
for c in 3 4 5; do python bench.py --config $c --no_cpu_baseline > $OUT/cfg$c.json 2> $OUT/cfg$c.err; echo "cfg$c done"; done
python bench.py --steps 20 --warmup 5 --no_cpu_baseline > $OUT/cfg2_20steps.json 2> $OUT/cfg2_20steps.err
echo "20-step run done"
DMH_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --batch_size 4 --atk_scenes 2 --no_cpu_baseline > $OUT/bench_2rank_one_gpu.json 2> $OUT/bench_2rank_one_gpu.err
echo "2-rank line done"
python -m depthmodelhardening_amd.train --dataset synthetic --frame_ids 0 --use_stereo --width 1024 --height 320 --batch_size 32 \
    --weights_init scratch --adv_train --norm_type l_inf --max_steps 12 --num_epochs 1 --synthetic_len 512 --log_dir /tmp/dmh_steplog \
    --model_name steplog --step_log $OUT/steps.jsonl > $OUT/train.log 2>&1
echo "step log done"
python3 tools/strong_scaling_point.py $OUT/strong_scaling_point.json > $OUT/strong.log 2>&1
tail -3 $OUT/strong.log
for f in cfg2 cfg3 cfg4 cfg5 cfg2_20steps; do python tools/parse_bench.py $OUT/$f.json | head -1; done
