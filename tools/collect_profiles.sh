#!/bin/bash
# Runs on the GPU box (gpurun): bench line, rocprofv3 kernel stats, the two PMC passes.  Raw output under gpurun_out/.
set -e
export TMPDIR=/tmp
R=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_n1.json.log 2> $O/bench_n1.err      # the default line: burst + sustained + fp32 + ngp legs + cpu_baseline
python bench.py --steps 10 --warmup 3 --hw 400 --n-importance 0 --precision 16 --no-cpu-baseline --no-extra-legs > $O/bench_cfg1_400_coarse_only.json.log 2>> $O/bench_n1.err
python bench.py --steps 10 --warmup 3 --n-rand 1024 --no-cpu-baseline --no-extra-legs > $O/bench_nrand1024.json.log 2>> $O/bench_n1.err
python bench.py --steps 10 --warmup 3 --config ngp > $O/bench_configs4_ngp.json.log 2>> $O/bench_n1.err
python tools/bench_kernels.py > $O/hbm_kernels.csv 2>> $O/bench_n1.err
python tools/probe_fp32.py > $O/fp32_mode.csv 2>> $O/bench_n1.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_prof.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/pmc_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/pmc_write.log 2>&1
echo write done
mkdir -p $O/summary
python tools/summarize_rocprof.py --round $R --stats $O/prof_bench --fetch $O/pmc_fetch --write $O/pmc_write --out $O/summary
# the declared reduced-precision bf16 mode of the same step: kernel stats only (continuity with rounds 1-3)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench_bf16 -- python3 bench.py --precision 16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_prof_bf16.log 2>&1
mkdir -p $O/summary_bf16
python tools/summarize_rocprof.py --round ${R}_bf16 --stats $O/prof_bench_bf16 --out $O/summary_bf16
find $O/prof_bench_bf16 -type f ! -name "*_kernel_stats.csv" -size +4M -delete
echo bf16 stats done
du -sh $O/prof_bench $O/pmc_fetch $O/pmc_write
# raw traces stay on the box unless small: keep the per-kernel stats file, drop the rest
find $O/prof_bench $O/pmc_fetch $O/pmc_write -type f ! -name "*_kernel_stats.csv" -size +4M -delete
echo summarised
