#!/bin/bash
# A/B of split-fp16 kernel builds (tools/diag/libnerf_f22_*.so): accuracy + speed lines of tools/check_f22.py per build
for n in "$@"; do
  echo "=== $n"
  NERF_HIP_LIB=$PWD/tools/diag/libnerf_f22_$n.so python tools/check_f22.py 2>&1 | grep -E "B=1024|split_fp16|Error|error"
done
