#!/bin/bash
# PMC passes over tools/probe_split22.py (counters only, one set per run): matrix-pipe busy + clock, instruction mix, waits.
#   tools/pmc_split22.sh r04 [tag] [program args...]     -> gpurun_out/<r>/split22_pmc<tag>.csv
# default program: tools/probe_split22.py (isolated back-to-back launches); e.g. `bench.py --precision 16 --steps 6 --warmup 2
# --no-cpu-baseline --no-extra-legs` profiles the kernels in the order and thermal context of the bench step.  Launches of one
# kernel are split into duration classes (large: fine pass / training size, mid: the 64-sample coarse pass).
set -e
export TMPDIR=/tmp
R=${1:-r04}; TAG=${2:-}
shift; shift || true
PROG="${@:-tools/probe_split22.py}"
O=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $O
rm -rf $O/split22_pmc$TAG
cd $GRAFT_REPO_ROOT
python3 $PROG > $O/split22_timing$TAG.log 2>&1
cat $O/split22_timing$TAG.log
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  if [ $i -gt ${NSETS:-5} ]; then break; fi
  REPS=2 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/split22_pmc$TAG/set$i -- python3 $PROG > $O/split22_pmc$TAG.set$i.log 2>&1 || echo "set $i failed: $SET"
done
python3 - <<PY
import csv, glob, statistics, collections
# one record per dispatch: counters of the same dispatch id are merged; launches of a kernel are then grouped into duration
# classes (>= 0.6 of the longest: "large" = the fine pass / training size; 0.2-0.6: "mid" = the 64-sample coarse pass)
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/split22_pmc$TAG/set*/**/*_counter_collection.csv", recursive=True)):
    runs = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "nerf::" not in n or not any(s in n for s in ("mlp22", "s16_", "mlp32", "mlp_fwd", "mlp_bwd", "mlp_dw")):
            continue
        d = runs[r["Dispatch_Id"]]
        d["k"] = (n.replace("void ", "").split("(")[0], int(r["Grid_Size"]))
        d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    tops = collections.defaultdict(float)
    for d in runs.values():
        tops[d["k"]] = max(tops[d["k"]], d["us"])
    for d in runs.values():
        frac = d["us"] / tops[d["k"]]
        cls = "large" if frac >= 0.6 else "mid" if frac >= 0.2 else None
        if cls is None:
            continue
        for c, v in d.items():
            if c != "k":
                per[d["k"] + (cls,)][c].append(v)
names = sorted({c for v in per.values() for c in v if c != "us"})
with open("$O/split22_pmc$TAG.csv", "w") as fp:
    fp.write("kernel,grid,class,launches_seen,avg_us," + ",".join(names) + ",clock_GHz,mfma_busy_frac\n")
    for k in sorted(per):
        row = [f"{statistics.mean(per[k][c]):.0f}" if per[k].get(c) else "" for c in names]
        g = per[k].get("GRBM_GUI_ACTIVE"); mf = per[k].get("SQ_VALU_MFMA_BUSY_CYCLES")
        clk = busy = ""
        t = statistics.mean(per[k]["us"])
        if g and mf:
            # clock and busy fraction from the pass that collected both (its own durations)
            gb, mb = statistics.mean(g), statistics.mean(mf)
            clk = f"{gb / 8 / t / 1e3:.3f}"; busy = f"{mb / (gb / 8 * 1024):.3f}"
        fp.write(f"\"{k[0]}\",{k[1]},{k[2]},{len(per[k]['us'])},{t:.1f}," + ",".join(row) + f",{clk},{busy}\n")
print(open("$O/split22_pmc$TAG.csv").read())
PY
find $O/split22_pmc$TAG -type f -size +2M -delete
