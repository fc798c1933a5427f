# LDS bank-conflict counters of the sparse select pass on the bench workload
# usage (GPU box): bash scripts/sp_pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-r04}
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sp_$V -o x -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-alone > $R/gpurun_out/pmc_sp_$V.log 2>&1
python - <<PY
import csv, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open("$R/gpurun_out/pmc_sp_$V/x_counter_collection.csv")):
    if "k_sparse_select" not in r["Kernel_Name"]: continue
    e = agg.setdefault(int(r["Dispatch_Id"]), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for d, e in agg.items():
    print(d, {k: (round(v, 3) if k == "ms" else v) for k, v in e.items()}, "conflict/active", round(e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"], 3))
PY
