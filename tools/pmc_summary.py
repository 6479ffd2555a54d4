"""Per-kernel averages of rocprofv3 --pmc counter_collection CSV files (one pass per file), restricted to the launches with the largest grid of each
kernel (the bench batch).  Usage: pmc_summary.py out.json pass1_counter_collection.csv [pass2 ...]
Writes {kernel: {counter: average per launch}} and prints a table."""
import csv, json, sys
from collections import defaultdict
out, files = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    rows = list(csv.DictReader(open(f)))
    gmax = defaultdict(int)
    for r in rows:
        gmax[r["Kernel_Name"]] = max(gmax[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) == gmax[r["Kernel_Name"]]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k.split("(")[0].replace("void ", "").replace("oslam::", ""): {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if "oslam" in k}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, cs in sorted(res.items()):
    print(k, {c: round(v, 1) for c, v in cs.items()})
