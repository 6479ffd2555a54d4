"""Generates tests/golden/io_golden.json by importing the reference's associate.py (importable under python3;
only read_file_list runs there — associate() uses python2 dict semantics).  Run in the build container:
    python tests/golden/gen_io_golden.py
The two *_head.txt fixtures are the first rows of data files the reference ships
(Examples/RGB-D/associations/fr2_desk.txt, ExpResults/KITTI/groundtruth/00.txt)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/ExpResults/TUM/Localization")
import associate  # noqa: E402

d = associate.read_file_list(os.path.join(HERE, "fr2_desk_head.txt"))
json.dump({"read_file_list": sorted([[k, v] for k, v in d.items()])}, open(os.path.join(HERE, "io_golden.json"), "w"), indent=1)
print(len(d), "rows")
