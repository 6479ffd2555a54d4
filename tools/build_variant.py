"""Kernel experiments: a copy of the library in which ONE source file is compiled with extra flags (the other objects are the cached ones of the normal build).
    python tools/build_variant.py <name> <source.hip> <flags...>    ->  tools/_build/liboslam_hip_<name>.so   (use with OSLAM_LIB_PATH=...)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_slam_amd import build as B

name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_hip()
out_dir = os.path.join(ROOT, "tools", "_build")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, "%s.%s.o" % (os.path.splitext(src)[0], name))
subprocess.check_call([B.HIPCC] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj])
lib = os.path.join(out_dir, "liboslam_hip_%s.so" % name)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj if s == src else B._obj_path(s) for s in B.SOURCES] + ["-o", lib])
print(lib)
