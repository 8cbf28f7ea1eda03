#!/bin/bash
# Precision-22 arm of the converged-regime ensemble (same configuration and seeds as tools/r4_converged.sh; lead arm only: its
# batches, uniforms, initial weights and evaluation pixels are functions of (seed, iteration), so the rows pair with the stored
# bf16 / fp32 / null arms of profiles/r04_psnr_converged_24seeds_*.jsonl).
#   tools/r4_converged_p22.sh A 4,10,18,21,28,33,47,58
set -e
TAG=$1; SEEDS=$2
mkdir -p gpurun_out/r04_conv
timeout -k 10 1160 python tools/psnr_ensemble.py --seed-list $SEEDS --hw 800 --n-rand 1024 --iters 20000 --every 1000 \
  --eval-pixels 80000 --dead-every 20 --lead-precision 22 --lead-only --out gpurun_out/r04_conv/p22_$TAG.jsonl > gpurun_out/r04_conv/p22_$TAG.log 2>&1
tail -2 gpurun_out/r04_conv/p22_$TAG.log | cut -c1-300
