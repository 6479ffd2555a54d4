"""One-off fuzz: random extractor geometries, HIP vs oracle, bit-exact keypoints + descriptors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from object_slam_amd import ORBextractor, synth
from oracle import oracle_py as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for t in range(n_cfg):
    w = int(rng.integers(96, 900)); h = int(rng.integers(96, 700))
    nf = int(rng.integers(100, 3000)); nl = int(rng.integers(1, 9)); sf = float(np.float32(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0])))
    ini, mn = int(rng.integers(10, 40)), int(rng.integers(3, 10))
    # top level must keep a FAST region: skip geometries the reference itself cannot run
    tw, th = w / sf ** (nl - 1), h / sf ** (nl - 1)
    if min(tw, th) < 50:
        continue
    # DistributeOctTree starts from round(width / height) root nodes (reference src/ORBextractor.cc:543): a region more than twice as tall as wide has none
    # and the reference divides by zero; the HIP extractor refuses such a geometry (1..64 roots), so the fuzz skips it
    if any(round(((w / sf ** l) - 32 + 6) / max((h / sf ** l) - 32 + 6, 1)) < 1 for l in range(nl)):
        continue
    kind = rng.integers(0, 3)
    if kind == 0:
        img = synth.make_stream(1, w, h, seed=int(rng.integers(1, 1 << 30)))[0][0]
    elif kind == 1:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        img[h // 3:h // 2, w // 4:w // 2] = 255 - img[0, 0]
    ex = None
    try:
        ex = ORBextractor(nf, sf, nl, ini, mn, w, h)
        k, d = ex(img)
        ox = O.OrbExtractor(nf, sf, nl, ini, mn)
        ok_, od = ox.extract(img)
        same = len(k) == len(ok_) and np.array_equal(np.asarray(k).view(np.uint8), np.asarray(ok_).view(np.uint8)) and np.array_equal(d, od)
    except Exception as e:
        same = False
        print("EXC", repr(e)[:200])
    print("%s %dx%d nf=%d nl=%d sf=%.2f th=%d/%d kind=%d -> %d kps" % ("ok " if same else "BAD", w, h, nf, nl, sf, ini, mn, kind, len(k) if 'k' in dir() else -1), flush=True)
    bad += not same
    if ex is not None and hasattr(ex, "close"): ex.close()
print("bad", bad)
