"""Front-end PMC summary in the form bench.py reads (profiles/rNN_pmc_traffic.json): per kernel, per launch of the bench batch (the launches with the largest
grid), memory-side bytes (FETCH_SIZE + WRITE_SIZE, KB -> bytes, with MI355X_MICROARCH.md's correction for 16-byte-per-lane loads) and the VALU issue share
(SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / (average duration from the kernel trace x 2.4 GHz)).
usage: pmc_frontend_traffic.py out.json kernel_trace.csv fetch_counter_collection.csv write_counter_collection.csv sq_counter_collection.csv"""
import csv, json, sys
from collections import defaultdict

out, kt, files = sys.argv[1], sys.argv[2], sys.argv[3:]
short = lambda k: k.split("(")[0].replace("void ", "").replace("oslam::", "")
FETCH_X2 = ("k_resize_lds",)   # kernels whose loads are 16 B per lane: FETCH_SIZE counts their requests at half size (guide, HBM section)

acc = defaultdict(lambda: defaultdict(list))
for f in files:
    rows = list(csv.DictReader(open(f)))
    gmax = defaultdict(int)
    for r in rows:
        gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) == gmax[r["Kernel_Name"]]:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
rows = list(csv.DictReader(open(kt)))
gmax = defaultdict(int)
for r in rows:
    gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))
for r in rows:
    if int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) == gmax[r["Kernel_Name"]]:
        dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {"_note": "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES) and a --kernel-trace pass of tools/frontend_pmc.py = the S2 stage "
                "of bench.py at 512 frames per launch; per-launch averages over the launches with the largest grid of each kernel; FETCH_SIZE / WRITE_SIZE in KB; "
                "k_resize_lds loads 16 B per lane, so its FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM); the other kernels load 4 B per lane. "
                "valu_issue_frac = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / (trace duration x 2.4 GHz)."}
for k, cs in sorted(acc.items()):
    if not k.startswith("k_"):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    fx = 2.0 if k.startswith(FETCH_X2) else 1.0
    t = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
    row = {"fetch_kb": round(m.get("FETCH_SIZE", 0.0), 1), "write_kb": round(m.get("WRITE_SIZE", 0.0), 1), "fetch_correction": fx,
           "bytes_per_launch": int((m.get("FETCH_SIZE", 0.0) * fx + m.get("WRITE_SIZE", 0.0)) * 1024), "valu_insts": m.get("SQ_INSTS_VALU"), "salu_insts": m.get("SQ_INSTS_SALU"),
           "lds_insts": m.get("SQ_INSTS_LDS"), "waves": m.get("SQ_WAVES"), "trace_avg_ns": round(t, 1) if t else None}
    if t and m.get("SQ_INSTS_VALU") is not None:
        row["valu_issue_frac_at_4_cycles"] = round(m["SQ_INSTS_VALU"] * 4 / 1024 / (t * 2.4), 4)
    res[k] = row
    print(k, row)
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
