"""Measured fp64 peak of the card: v_fma_f64 and v_mfma_f64_16x16x4_f64 issue rates with every CU busy (tools/mfma_f64_rate.hip).
Prints one JSON line; bench.py calls measure() and puts the result beside the 78.6 TFLOP/s datasheet figure it prices the optimisers against."""
import ctypes as C
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mfma_f64_rate.hip")
LIB = os.path.join(HERE, "_build", "libf64rate.so")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", SRC, "-o", LIB])
    return LIB


def measure(device=0, iters=20000, reps=3):
    L = C.CDLL(build())
    L.oslam_tool_f64_rate.restype = C.c_double
    L.oslam_tool_f64_rate.argtypes = [C.c_int] * 5
    out = {}
    for kind, name in ((0, "v_fma_f64"), (1, "v_mfma_f64_16x16x4_f64")):
        rates = {w: round(L.oslam_tool_f64_rate(kind, w, iters, reps, device), 2) for w in (1, 2, 4)}
        out[name] = {"TFLOPs_by_waves_per_simd": rates, "best_TFLOPs": max(rates.values())}
    out["datasheet_TFLOPs"] = 78.6
    out["note"] = "8 independent chains per wave, 256-thread blocks, every CU busy; best of %d launches of %d iterations" % (reps, iters)
    return out


if __name__ == "__main__":
    print(json.dumps(measure()))
