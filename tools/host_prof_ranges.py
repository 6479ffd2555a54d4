"""Aggregates the innermost-line table of a tools/host_prof.py report by the functions of slam_driver.hip / slam_map.h (ranges from the `static ... {` lines of the
source AS IT IS NOW: run it on the tree the profile was taken from).  usage: host_prof_ranges.py report.txt [top=40]"""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rep = open(sys.argv[1]).read()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sec = rep.split("source lines (innermost inlined frame):")[1].split("source lines (outermost")[0]
rows = []
for l in sec.strip().split("\n"):
    m = re.match(r"\s*([\d.]+) %\s+(\S+):(\d+)", l)
    if m: rows.append((float(m.group(1)), m.group(2), int(m.group(3))))
def ranges(path):
    out = []
    for i, l in enumerate(open(path), 1):
        m = re.match(r"^(?:static |inline |template.*|int |void |struct |    (?:int|void|bool|IntSpan) )\s*.*?([A-Za-z_0-9:]+)\s*\(.*\)\s*(?:const)?\s*\{", l)
        if m and not l.startswith("        "): out.append((i, m.group(1)))
    return out
files = {"slam_driver.hip": ranges(os.path.join(ROOT, "object_slam_amd/csrc/slam_driver.hip")), "slam_map.h": ranges(os.path.join(ROOT, "object_slam_amd/csrc/slam_map.h"))}
tot = {}
for pct, f, ln in rows:
    key = f
    if f in files:
        name = "?"
        for a, n in files[f]:
            if a <= ln: name = n
            else: break
        key = "%s:%s" % (f, name)
    tot[key] = tot.get(key, 0) + pct
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:top]: print("%6.2f %%  %s" % (v, k))
