#!/bin/bash
# Extra PMC passes (own runs, counters only): matrix-pipe busy cycles + GPU active cycles, LDS conflicts.
set -e
export TMPDIR=/tmp
R=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/pmc_mfma.log 2>&1
echo mfma done
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lds -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $O/pmc_lds.log 2>&1
echo lds done
python - <<PY
import csv, glob, statistics, collections
def load(d):
    f = sorted(glob.glob("$O/" + d + "/**/*_counter_collection.csv", recursive=True))[-1]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "nerf::mlp" in r["Kernel_Name"] or "nerf::s16" in r["Kernel_Name"] or "nerf::f22" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].replace("void ", "").split("(")[0], int(r["Grid_Size"]))
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[k]["us:" + r["Counter_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per
a, b = load("pmc_mfma"), load("pmc_lds")
with open("$O/pmc_mfma_lds.csv", "w") as fp:
    fp.write("kernel,grid,launches,avg_us,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,clock_GHz,mfma_busy_frac_of_cycles,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,lds_conflict_frac\n")
    for k in sorted(a):
        # per launch class: the largest launches of each (kernel, grid) (render fine pass / fine training pass)
        us = a[k]["us:GRBM_GUI_ACTIVE"]; top = max(us)
        sel = [i for i, u in enumerate(us) if u > 0.6 * top]
        m = statistics.mean([a[k]["SQ_VALU_MFMA_BUSY_CYCLES"][i] for i in sel]); g = statistics.mean([a[k]["GRBM_GUI_ACTIVE"][i] for i in sel])
        t = statistics.mean([us[i] for i in sel])
        usb = b[k]["us:SQ_LDS_IDX_ACTIVE"]; selb = [i for i, u in enumerate(usb) if u > 0.6 * max(usb)]
        c = statistics.mean([b[k]["SQ_LDS_BANK_CONFLICT"][i] for i in selb]); ia = statistics.mean([b[k]["SQ_LDS_IDX_ACTIVE"][i] for i in selb])
        cyc = g / 8                      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        fp.write(f"\"{k[0]}\",{k[1]},{len(sel)},{t:.1f},{m:.0f},{g:.0f},{cyc / t / 1e3:.3f},{m / (cyc * 1024):.3f},{c:.0f},{ia:.0f},{c / max(ia, 1):.4f}\n")
print(open("$O/pmc_mfma_lds.csv").read())
PY
find $O/pmc_mfma $O/pmc_lds -type f -size +4M -delete
