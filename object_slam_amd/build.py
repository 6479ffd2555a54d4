"""Builds the in-tree gfx950 shared library (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liboslam_hip.so")
SOURCES = ["orb_extractor.hip", "matcher.hip", "pose_opt.hip", "lba.hip", "stereo.hip", "bow_matcher.hip", "mappoint.hip", "frame.hip", "slam_driver.hip", "slam_ops_hip.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (os.environ["OSLAM_EXTRA_FLAGS"].split() if os.environ.get("OSLAM_EXTRA_FLAGS") else []) + (["-DOSLAM_LBA_PROFILE"] if os.environ.get("OSLAM_LBA_PROFILE") else []) + (["-DOSLAM_FAST_PROFILE"] if os.environ.get("OSLAM_FAST_PROFILE") else []) + (["-DOSLAM_MATCH_PROFILE"] if os.environ.get("OSLAM_MATCH_PROFILE") else []) + (["-DOSLAM_MATCH_ABLATE=" + os.environ["OSLAM_MATCH_ABLATE"]] if os.environ.get("OSLAM_MATCH_ABLATE") else []) + ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
         "-ffp-contract=off",  # host AND device: reference float expressions round once per operator
         "-Wall", "-Wno-unused-function"]


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files]
    out.append(os.path.join(HERE, "..", "include", "oslam_hip.h"))
    out.append(os.path.join(HERE, "..", "include", "oslam_slam.h"))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_hip(force="-f" in sys.argv, verbose=True)
