#!/bin/bash
# PMC passes over tools/probe_split22.py (counters only, one set per run): matrix-pipe busy + clock, instruction mix, waits.
#   tools/pmc_split22.sh r04 [tag]      -> gpurun_out/<r>/split22_pmc<tag>.csv
set -e
export TMPDIR=/tmp
R=${1:-r04}; TAG=${2:-}
O=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 tools/probe_split22.py > $O/split22_timing$TAG.log 2>&1
cat $O/split22_timing$TAG.log
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  REPS=2 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/split22_pmc$TAG/set$i -- python3 tools/probe_split22.py > $O/split22_pmc$TAG.set$i.log 2>&1 || echo "set $i failed: $SET"
done
python3 - <<PY
import csv, glob, statistics, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/split22_pmc$TAG/set*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "nerf::" not in n or not any(s in n for s in ("mlp22", "s16_", "mlp32", "mlp_fwd", "mlp_bwd", "mlp_dw")):
            continue
        k = (n.replace("void ", "").split("(")[0], int(r["Grid_Size"]))
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[k]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for v in per.values() for c in v if c != "us"})
with open("$O/split22_pmc$TAG.csv", "w") as fp:
    fp.write("kernel,grid,avg_us_of_largest_class," + ",".join(names) + ",clock_GHz,mfma_busy_frac\n")
    for k in sorted(per):
        us = per[k]["us"]; top = max(us)
        row = []
        for c in names:
            v = per[k].get(c, [])
            # counters of the largest launch class only (values scale with the launch)
            big = [x for x in v if x >= 0.6 * max(v)] if v else []
            row.append(f"{statistics.mean(big):.0f}" if big else "")
        t = statistics.mean([u for u in us if u >= 0.6 * top])
        g = per[k].get("GRBM_GUI_ACTIVE"); mf = per[k].get("SQ_VALU_MFMA_BUSY_CYCLES")
        clk = busy = ""
        if g and mf:
            gb = statistics.mean([x for x in g if x >= 0.6 * max(g)]); mb = statistics.mean([x for x in mf if x >= 0.6 * max(mf)])
            clk = f"{gb / 8 / t / 1e3:.3f}"; busy = f"{mb / (gb / 8 * 1024):.3f}"
        fp.write(f"\"{k[0]}\",{k[1]},{t:.1f}," + ",".join(row) + f",{clk},{busy}\n")
print(open("$O/split22_pmc$TAG.csv").read())
PY
find $O/split22_pmc$TAG -type f -size +2M -delete
