#!/bin/bash
# One gpurun call of the round-4 precision-22 ensemble: lead arm = Trainer(precision=22) (the bench's headline mode), reference arm
# = Trainer(precision=32), null arm = precision 22 with another dW summation order; 100 x 100, N_rand 1024, 2500 iterations
# (the configuration of profiles/r03_psnr_ensemble_170seeds_*), seeds = consecutive alive-at-init seeds from --seed-start.
#   tools/r4_p22_ensemble.sh A 0 16
set -e
TAG=$1; START=$2; COUNT=$3
mkdir -p gpurun_out/r04_p22
timeout -k 10 1160 python tools/psnr_ensemble.py --seeds $COUNT --seed-start $START --hw 100 --n-rand 1024 --iters 2500 --every 250 \
  --null-arm --lead-precision 22 --out gpurun_out/r04_p22/batch_$TAG.jsonl > gpurun_out/r04_p22/batch_$TAG.log 2>&1
tail -2 gpurun_out/r04_p22/batch_$TAG.log | cut -c1-300
