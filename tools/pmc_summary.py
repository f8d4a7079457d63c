"""Join the PMC passes of tools/pmc_layer.sh with their kernel traces: per kernel, per-dispatch averages."""
import collections, csv, glob, sys
root = sys.argv[1]
CLK = 2.4e9
out = collections.defaultdict(dict)
for d in sorted(glob.glob(f"{root}/p*")):
    cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not cc:
        continue
    rows = list(csv.DictReader(open(cc[0])))
    per = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in ids[k]:
            ids[k].add(r["Dispatch_Id"])
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if kt:
        dur = collections.defaultdict(float)
        for r in csv.DictReader(open(kt[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, v in per.items():
        n = len(ids[k])
        out[k]["n"] = n
        out[k]["dur_us"] = dur[k] / n / 1e3 if n else 0
        for c, val in v.items():
            out[k][c] = val / n
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("dur_us", 0)):
    if not any(t in k for t in ("k_pw", "k_dw", "k_stem")):
        continue
    print(f"== {k}  n={v['n']}  dur={v['dur_us']:.1f} us (under PMC)")
    iv = v.get("SQ_INSTS_VALU", 0); w = max(v.get("SQ_WAVES", 1), 1)
    print("   " + "  ".join(f"{c[3:]}={val:.4g}" for c, val in v.items() if c.startswith("SQ_")))
    oth = {c: val for c, val in v.items() if c.startswith(("TA_", "TCP_", "TCC_"))}
    if oth:
        print("   " + "  ".join(f"{c}={val:.4g}" for c, val in oth.items()))
    if iv:
        print(f"   per wave: valu={iv/w:.0f} salu={v.get('SQ_INSTS_SALU',0)/w:.0f} lds={v.get('SQ_INSTS_LDS',0)/w:.0f} "
              f"vmem={(v.get('SQ_INSTS_VMEM_RD',0)+v.get('SQ_INSTS_VMEM_WR',0))/w:.0f}   "
              f"valu_per_simd_cycle={iv/(v['dur_us']*1e-6*CLK*1024):.3f}  all_per_simd_cycle={(iv+v.get('SQ_INSTS_SALU',0)+v.get('SQ_INSTS_LDS',0))/(v['dur_us']*1e-6*CLK*1024):.3f}")
