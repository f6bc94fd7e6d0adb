#!/usr/bin/env python3
"""Per-op PMC summary from rocprofv3 --pmc counter_collection.csv (long format), joined with the plan."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.prof_ops import plan_names
d = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # batch of the traced run (tools/prof_ops.py --run --batch B)
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
disp = {}
for r in csv.DictReader(open(f)):
    if not r["Kernel_Name"].startswith(("void rib::", "rib::")): continue
    k = int(r["Dispatch_Id"])
    e = disp.setdefault(k, {"name": r["Kernel_Name"], "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"]),
                            "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "lds": r["LDS_Block_Size"]})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
ops = plan_names(B, 512, 512)
n = len(ops)
ids = ids[-n:]
ctrs = [c for c in disp[ids[0]] if c not in ("name", "t0", "t1", "vgpr", "agpr", "lds")]
print("%-46s %8s %5s %6s " % ("op", "us", "vgpr", "lds") + " ".join("%14s" % c[-14:] for c in ctrs))
for o, i in zip(ops, ids):
    e = disp[i]
    print("%-46s %8.1f %5s %6s " % (o["name"][-46:], (e["t1"] - e["t0"]) / 1e3, e["vgpr"] + "+" + e["agpr"], e["lds"]) + " ".join("%14.0f" % e.get(c, 0) for c in ctrs))
