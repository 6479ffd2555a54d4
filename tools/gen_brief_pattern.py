#!/usr/bin/env python3
"""Extract the 256x4 learned rBRIEF sampling table (data, not code) from the
reference (src/ORBextractor.cc:150-408, `bit_pattern_31_`) and write it as a
bare comma-separated int8 list usable from C/C++/HIP via #include.

Run in the build container only (needs /root/reference); the generated file is
committed so nothing reads /root/reference at run time.
"""
import hashlib
import re
import sys

SRC = "/root/reference/src/ORBextractor.cc"
OUT = "object_slam_amd/csrc/brief_pattern.inc"

text = open(SRC).read()
m = re.search(r"static int bit_pattern_31_\[256\*4\]\s*=\s*\{(.*?)\};", text, re.S)
body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
vals = [int(v) for v in re.findall(r"-?\d+", body)]
assert len(vals) == 1024, len(vals)
assert min(vals) >= -13 and max(vals) <= 12
digest = hashlib.sha256(",".join(map(str, vals)).encode()).hexdigest()
with open(OUT, "w") as f:
    f.write("// rBRIEF 31x31 learned test pattern: 256 tests x (x0,y0,x1,y1), int8 range [-13,12].\n")
    f.write("// DATA table; source: reference src/ORBextractor.cc:150-408. sha256(comma-joined ints)=\n")
    f.write("// %s\n" % digest)
    for i in range(0, 1024, 16):
        f.write(",".join("%d" % v for v in vals[i:i + 16]) + ",\n")
print(digest)
