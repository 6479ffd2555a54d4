"""MFMA utilisation of the local-BA kernels from rocprofv3 --pmc passes (counter_collection CSV, one row per dispatch and counter) + a kernel-trace CSV of the
same program.  Usage: pmc_mfma_summary.py out.json kernel_trace.csv pmc1_counter_collection.csv [pmc2 ...]
Per kernel: launches, mean duration, MFMA instructions (SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512-flop units... reported raw), SQ_VALU_MFMA_BUSY_CYCLES, and
mfma_util = busy cycles / (duration x 2.4 GHz x 4 SIMDs x workgroups' CUs) — the share of the matrix pipes of the CUs the kernel occupies."""
import csv, json, sys
from collections import defaultdict
out, trace, files = sys.argv[1], sys.argv[2], sys.argv[3:]
dur = defaultdict(list); grid = {}
for r in csv.DictReader(open(trace)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oslam::", "")
    dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    grid[k] = max(grid.get(k, 0), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])))
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oslam::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    if not k.startswith("k_"):
        continue
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e["launches"] = len(dur.get(k, []))
    e["mean_us"] = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
    e["workgroups_max"] = grid.get(k)
    busy = e.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if busy is not None and e["mean_us"]:
        cus = min(256, max(1, grid.get(k, 1)))
        e["mfma_util_of_occupied_cus"] = busy / (e["mean_us"] * 1e-6 * 2.4e9 * 4 * cus)
        e["mfma_util_chip"] = busy / (e["mean_us"] * 1e-6 * 2.4e9 * 4 * 256)
    res[k] = e
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items()):
    print(k, {c: (round(v, 4) if isinstance(v, float) else v) for c, v in e.items()})
