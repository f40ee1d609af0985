"""Condense the rocprofv3 outputs of tools/profile_round.sh into the two files kept under profiles/:
<tag>_kernel_stats.csv (copy of the --stats table) and <tag>_gram_pmc.json (per-launch counter means of the hot kernels).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 32 B per 64-B request for 16-B/lane streams, so it is
doubled (MI355X_MICROARCH.md, HBM section; calibrated on row_moments_kernel whose algorithmic read volume is known)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
res = {"source": "tools/profile_round.sh: rocprofv3 --kernel-trace --stats (bench.py --steps 8 --warmup 3) and one --pmc pass per counter "
                 "group (bench.py --steps 3 --warmup 3); means over all launches of the timed+warm-up steps"}
stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
HOT = ("gram_dma_kernel", "gram_f32_kernel", "gram_kernel", "row_moments_kernel", "flow_eval_kernel", "chol_diag_kernel", "chol_syrk_diag_kernel", "chol_update_diag16_kernel", "chol_strip_kernel",
       "reduce_splits_kernel", "solve_bwd_persistent", "finalize_S_kernel")
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        key = next((h for h in HOT if h in name), None)
        if key is None:
            continue
        if key == "gram_kernel":
            key = "gram_kernel" + name[name.find("<"):name.find(">") + 1] if "<" in name else key
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur[key][row["Counter_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
for k in acc:
    res[k] = {c: {"mean": sum(v) / len(v), "launches": len(v), "mean_ms": sum(dur[k][c]) / len(v)} for c, v in acc[k].items()}
    g = res[k]
    if "FETCH_SIZE" in g and "WRITE_SIZE" in g:
        g["hbm_bytes_per_launch"] = 2.0 * 1024.0 * g["FETCH_SIZE"]["mean"] + 1024.0 * g["WRITE_SIZE"]["mean"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "SQ_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 4 SIMDs of all 256 CUs
        cycles = g["GRBM_GUI_ACTIVE"]["mean"] / 8.0
        g["mfma_busy_frac_of_cycles"] = g["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (cycles * 256 * 4)
        g["clock_ghz"] = cycles / (g["GRBM_GUI_ACTIVE"]["mean_ms"] * 1e6)
import datetime
res["date"] = datetime.date.today().isoformat()
json.dump(res, open(os.path.join(out, f"{tag}_gram_pmc.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)} for k, v in res.items() if isinstance(v, dict)}, indent=1))
