#!/bin/bash
# Round-3 probe collection on the GPU box (gpurun): NGP scatter by level / mode, the hand-off microbenchmark (+ PMC),
# the fp32 mode's kernel split.  Raw output under gpurun_out/r03p/.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03p
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $O/t.log 2>&1; echo "pytest rc=$?"; tail -6 $O/t.log
python tools/probe_ngp_scatter.py > $O/ngp_scatter.csv 2> $O/ngp_scatter.err; echo "ngp rc=$?"; grep -E "0-15|train_step|adam" $O/ngp_scatter.csv
H=tools/diag/handoff_probe
: > $O/handoff.csv
for a in "128 16 2 256 0 16" "128 16 1 256 0 16" "128 16 3 256 0 16" "128 32 2 256 0 16" "64 8 2 256 0 16" "128 16 2 256 0 8" "1024 128 2 256 0 16" "8192 128 2 256 0 16" "128 16 2 256 800 16"; do
  timeout -k 5 120 $H $a >> $O/handoff.csv 2>&1 || echo "handoff $a rc=$?"
done
cat $O/handoff.csv | grep -v ring_KiB | awk -F, '{print $1,$2,$3,$5,$6,$7,"ms",$10,"chipGB/s",$12,"xcdGB/s",$13,"cuGB/s",$14,"bad",$15,"err",$16}' | awk 'NR%3==0'
for cfg in "128 16 2 64 0 16" "8192 128 2 64 0 16"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$tag -- $H $cfg > $O/pmc_fetch_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$tag -- $H $cfg > $O/pmc_write_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r03p"
for d in sorted(glob.glob(O + "/pmc_*_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        tot = {}
        for r in rows:
            if "handoff" in r.get("Kernel_Name", ""):
                tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in tot.items():
            print(os.path.basename(d), k, "per launch:", [round(x / 1e6, 1) for x in v], "(counter units x 1e6)")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -- python3 tools/probe_fp32.py > $O/fp32_prof.log 2>&1
f=$(find $O/prof_fp32 -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
find $O -type f -size +4M -delete
echo done
