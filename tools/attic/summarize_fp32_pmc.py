import csv, glob, statistics, collections
f = sorted(glob.glob("gpurun_out/f32/pmc2/**/*_counter_collection.csv", recursive=True))[-1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "mlp32" in r["Kernel_Name"]:
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[k]["us:" + r["Counter_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,launches,avg_us,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,clock_GHz,mfma_busy_frac_of_cycles")
for k in sorted(per):
    a = per[k]; us = a["us:GRBM_GUI_ACTIVE"]; top = max(us)
    sel = [i for i, u in enumerate(us) if u > 0.6 * top]
    m = statistics.mean([a["SQ_VALU_MFMA_BUSY_CYCLES"][i] for i in sel]); g = statistics.mean([a["GRBM_GUI_ACTIVE"][i] for i in sel])
    t = statistics.mean([us[i] for i in sel]); cyc = g / 8
    print(f"\"{k}\",{len(sel)},{t:.1f},{m:.0f},{g:.0f},{cyc / t / 1e3:.3f},{m / (cyc * 1024):.3f}")
