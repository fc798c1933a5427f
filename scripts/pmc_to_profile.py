"""Condense the rocprofv3 --pmc passes over bench.py (one directory per counter group, as written by
scripts/final_prof.sh) into profiles/<name>.json: per scan launch of the last timed step, HBM traffic
(FETCH_SIZE x 2 on gfx950), L2 hit rate, clock, MFMA busy fraction; the sparse select launch: TCC misses, wait split.
    python scripts/pmc_to_profile.py gpurun_out/pmc_r02 profiles/r02_pmc_scan.json"""
import csv, collections, json, sys
root, out_path = sys.argv[1], sys.argv[2]
SP = "k_sparse_select"

def load(name):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f"{root}/{name}/x_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "k_scan" not in k and SP not in k:
            continue
        d = int(r["Dispatch_Id"])
        e = agg.setdefault(d, {"kernel": k.split("(")[0], "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return agg

def last(agg, pat, n):
    ks = [d for d in agg if pat in agg[d]["kernel"]]
    return [agg[d] for d in ks[-n:]]

f, t, s = load("FETCH_SIZE"), load("TCC_HIT_sum"), load("SQ_WAVE_CYCLES")
n_scan = 4
res = {"command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-secondary  (one pass per group: FETCH_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_* GRBM_GUI_ACTIVE)",
       "note": "FETCH_SIZE is in KiB and, on gfx950, reports half the bytes of 16 B/lane streaming reads "
               "(MI355X_MICROARCH.md, HBM): traffic_gb = FETCH_SIZE*1024*2/1e9; TCC_MISS*128 B agrees.",
       "config": {"rows": 10000000, "dim": 768, "batch": 1024, "n_gpus": 1}, "scan_launches_of_one_step": []}
for i, e in enumerate(last(f, "k_scan", n_scan)):
    tt, ss = last(t, "k_scan", n_scan)[i], last(s, "k_scan", n_scan)[i]
    res["scan_launches_of_one_step"].append({
        "kernel": e["kernel"], "ms": round(e["ms"], 3), "FETCH_SIZE_KiB": e["FETCH_SIZE"],
        "traffic_gb": round(e["FETCH_SIZE"] * 1024 * 2 / 1e9, 3), "TCC_HIT": tt["TCC_HIT_sum"], "TCC_MISS": tt["TCC_MISS_sum"],
        "l2_hit_rate": round(tt["TCC_HIT_sum"] / (tt["TCC_HIT_sum"] + tt["TCC_MISS_sum"]), 3),
        "clock_ghz": round(ss["GRBM_GUI_ACTIVE"] / 8 / (ss["ms"] * 1e6), 3),
        "mfma_busy_frac_at_clock": round(ss["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (ss["GRBM_GUI_ACTIVE"] / 8), 3),
        **{k: ss[k] for k in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}})
sp, spt, sps = last(f, SP, 1)[0], last(t, SP, 1)[0], last(s, SP, 1)[0]
res["sparse_select_launch"] = {
    "kernel": sp["kernel"], "ms": round(sp["ms"], 3), "FETCH_SIZE_KiB": sp["FETCH_SIZE"],
    "tcc_miss_x128B_gb": round(spt["TCC_MISS_sum"] * 128 / 1e9, 3), "TCC_HIT": spt["TCC_HIT_sum"], "TCC_MISS": spt["TCC_MISS_sum"],
    "clock_ghz": round(sps["GRBM_GUI_ACTIVE"] / 8 / (sps["ms"] * 1e6), 3),
    "wait_any_share_of_wave_cycles": round(sps["SQ_WAIT_ANY"] / sps["SQ_WAVE_CYCLES"], 3),
    **{k: sps[k] for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}}
tot = sum(x["traffic_gb"] for x in res["scan_launches_of_one_step"])
res["scan_traffic_gb_per_step"] = round(tot, 3)
res["scan_traffic_gb_per_launch"] = round(tot / n_scan, 3)
json.dump(res, open(out_path, "w"), indent=1)
for x in res["scan_launches_of_one_step"]:
    print(x["kernel"][-30:], x["ms"], "ms traffic", x["traffic_gb"], "GB l2hit", x["l2_hit_rate"], "clock", x["clock_ghz"], "mfma", x["mfma_busy_frac_at_clock"])
print("sparse select", res["sparse_select_launch"]["ms"], "ms, TCC_MISS*128 =", res["sparse_select_launch"]["tcc_miss_x128B_gb"], "GB, wait share",
      res["sparse_select_launch"]["wait_any_share_of_wave_cycles"])
