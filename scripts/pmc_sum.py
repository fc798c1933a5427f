"""Diagnostic: sum rocprofv3 --pmc counters per dispatch.  argv: <dir with */x_counter_collection.csv> <kernel substring>"""
import csv, collections, glob, sys
root, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(f"{root}/*/x_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    tm = {}
    for r in rows:
        if pat not in r["Kernel_Name"]:
            continue
        agg[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        tm[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for did in sorted(agg)[-2:]:
        print(f"dispatch {did} {tm[did]:.3f} ms", {k: f"{v:.4g}" for k, v in agg[did].items()})
