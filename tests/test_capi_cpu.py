"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/oslam_hip.h
declares, and fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "oslam_hip.h")).read() + open(os.path.join(ROOT, "include", "oslam_slam.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(oslam_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_library_builds_and_exports_all_symbols():
    from object_slam_amd import build
    so = build.build_hip()
    L = C.CDLL(so)
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing export %s" % n


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from object_slam_amd import ORBextractor, ORBmatcher, OslamError
    with pytest.raises(OslamError) as ei:
        ORBextractor(1000, 1.2, 8, 20, 7, 640, 480)
    assert "no CPU fallback" in str(ei.value)
    with pytest.raises(OslamError):
        ORBmatcher(0.9, True)


def test_product_does_not_use_oracle():
    """The oracle is test infrastructure: nothing under object_slam_amd/ or include/ may import,
    include, link or execute it."""
    pat = re.compile(r"(^\s*(from|import)\s+oracle)|(#include\s+\"[^\"]*oracle)|(liboslam_oracle)|(oracle_py)", re.M)
    for top in ("object_slam_amd", "include"):
        for root, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cc", ".inc")):
                    src = open(os.path.join(root, f)).read()
                    assert not pat.search(src), os.path.join(root, f)


def test_ctypes_mirrors_have_the_sizes_of_the_headers(tmp_path):
    """The Python views mirror the C structs of include/*.h by hand.  A mirror that is SHORTER than its struct is a buffer overflow on both sides of the C ABI (round 5:
    SlamOps missed the ten operators added that round — 248 instead of 328 bytes — and every oslam_slam_create_with_ops of the CPU suite wrote 80 bytes past a ctypes
    buffer: the suite crashed or not depending on the order of its tests).  gcc prints the sizes of the headers; the library reports its own (oslam_slam_struct_sizes)."""
    import ctypes as C
    import subprocess
    from object_slam_amd import mappoint, matcher, optimizer, slam
    from object_slam_amd._lib import lib
    pairs = [("oslam_slam_config_t", slam.SlamConfig), ("oslam_slam_ops_t", slam.SlamOps), ("oslam_slam_objects_t", slam.SlamObjects),
             ("oslam_tri_kf_t", mappoint.TriKF), ("oslam_camera_t", matcher.Camera), ("oslam_match_frames_t", matcher.MatchFrames),
             ("oslam_match_last_t", matcher.MatchLast), ("oslam_bow_side1_t", matcher.BowSide1), ("oslam_bow_side2_t", matcher.BowSide2),
             ("oslam_semantic_t", optimizer.Semantic), ("oslam_lba_problem_t", optimizer.LbaProblem)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "oslam_hip.h"\n#include "oslam_slam.h"\nint main(void) {\n'
                   + "".join('    printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n, _ in pairs) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    c_sizes = dict((l.split()[0], int(l.split()[1])) for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, cls in pairs:
        assert C.sizeof(cls) == c_sizes[name], (name, C.sizeof(cls), c_sizes[name])
    out = (C.c_int32 * 4)()
    assert lib().oslam_slam_struct_sizes(out) == 0
    assert (out[0], out[1], out[2]) == (c_sizes["oslam_slam_config_t"], c_sizes["oslam_slam_ops_t"], c_sizes["oslam_slam_objects_t"])
