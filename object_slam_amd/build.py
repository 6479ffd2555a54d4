"""Builds the in-tree gfx950 shared library (hipcc cross-compiles without a GPU).

Every source is compiled to its own object under object_slam_amd/_obj/ (only when it or a header changed, up to OSLAM_BUILD_JOBS at a time) and the
objects are linked into liboslam_hip.so: an edit of one kernel file costs one compile, not ten."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "liboslam_hip.so")
SOURCES = ["orb_extractor.hip", "matcher.hip", "pose_opt.hip", "lba.hip", "stereo.hip", "bow_matcher.hip", "mappoint.hip", "frame.hip", "slam_driver.hip", "slam_ops_hip.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (os.environ["OSLAM_EXTRA_FLAGS"].split() if os.environ.get("OSLAM_EXTRA_FLAGS") else []) + (["-DOSLAM_LBA_PROFILE"] if os.environ.get("OSLAM_LBA_PROFILE") else []) + (["-DOSLAM_FAST_PROFILE"] if os.environ.get("OSLAM_FAST_PROFILE") else []) + (["-DOSLAM_MATCH_PROFILE"] if os.environ.get("OSLAM_MATCH_PROFILE") else []) + (["-DOSLAM_MATCH_ABLATE=" + os.environ["OSLAM_MATCH_ABLATE"]] if os.environ.get("OSLAM_MATCH_ABLATE") else []) + ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17",
         "-ffp-contract=off",  # host AND device: reference float expressions round once per operator
         "-Wall", "-Wno-unused-function"]


def _headers():
    out = []
    for root, _, files in os.walk(CSRC):   # (every file under csrc/ that is not a translation unit: headers, .inc — and .hip files that are #included, e.g. orb_kernels.hip)
        out += [os.path.join(root, f) for f in files if f not in SOURCES]
    out.append(os.path.join(HERE, "..", "include", "oslam_hip.h"))
    out.append(os.path.join(HERE, "..", "include", "oslam_slam.h"))
    return out


def _deps():
    return _headers() + [os.path.join(CSRC, s) for s in SOURCES]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _obj_path(src):
    tag = hashlib.sha1(" ".join(FLAGS).encode()).hexdigest()[:10]   # objects of another flag set are not reused
    return os.path.join(OBJ, "%s.%s.o" % (os.path.splitext(src)[0], tag))


def _stale(src, obj):
    """obj is older than src or than any file the compiler listed as included (-MD); system headers under /opt/rocm are skipped"""
    if not os.path.exists(obj) or not os.path.exists(obj + ".d"):
        return True
    t = os.path.getmtime(obj)
    deps = open(obj + ".d").read().replace("\\\n", " ").split()[1:]
    for d in [src] + [d for d in deps if not d.startswith(("/opt/", "/usr/"))]:
        if not os.path.exists(d) or os.path.getmtime(d) > t:
            return True
    return False


def build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    todo = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), _obj_path(s)
        if force or _stale(src, obj):
            todo.append((src, obj))

    def compile_one(job):
        # objects and dependency files appear under their final names only when complete (os.replace): several ranks / test workers may build the same
        # stale library at once, and none of them may link or date-check another one's half-written file
        tmp = "%s.%d.tmp" % (job[1], os.getpid())
        cmd = [HIPCC] + FLAGS + ["-MD", "-MF", tmp + ".d", "-MT", job[1], "-c", job[0], "-o", tmp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(tmp + ".d", job[1] + ".d")
        os.replace(tmp, job[1])

    jobs = int(os.environ.get("OSLAM_BUILD_JOBS", "0")) or min(6, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(compile_one, todo))
    lib_tmp = "%s.%d.tmp" % (LIB, os.getpid())
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj_path(s) for s in SOURCES] + ["-o", lib_tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(lib_tmp, LIB)
    return LIB


if __name__ == "__main__":
    build_hip(force="-f" in sys.argv, verbose=True)
