#!/bin/bash
# One gpurun call of the round-4 converged-regime PSNR ensemble: 800 x 800, N_rand 1024, 20 000 iterations,
# held-out PSNR every 1000 on 80 000 fixed pixels of 2 test views; arms bf16 / fp32 (reference arithmetic) / bf16b (null arm).
# Seeds = the first 24 alive-at-init seeds from 0, fixed BEFORE any run: 4 10 18 21 | 28 33 47 58 | 64 69 71 81 | 88 101 103 123 |
# 127 138 143 169 | 178 185 196 200
#   tools/r4_converged.sh A 4,10,18,21
set -e
TAG=$1; SEEDS=$2
mkdir -p gpurun_out/r04_conv
timeout -k 10 1160 python tools/psnr_ensemble.py --seed-list $SEEDS --hw 800 --n-rand 1024 --iters 20000 --every 1000 \
  --eval-pixels 80000 --null-arm --dead-every 20 --out gpurun_out/r04_conv/batch_$TAG.jsonl > gpurun_out/r04_conv/batch_$TAG.log 2>&1
tail -2 gpurun_out/r04_conv/batch_$TAG.log | cut -c1-300
