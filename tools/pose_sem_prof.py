"""k_pose_optimize<true> (ObjectOptimizer::PoseOptimization2, the driver's TrackLocalMap stage) alone: B frames per launch built from 16 synthetic semantic problems
(N = 1000 keypoints, 3 objects, ~1000 object map points + M_joint candidates per frame like the headline stream), timing per launch; with a profiling build
(OSLAM_EXTRA_FLAGS=-DOSLAM_POSE_PROFILE or tools/build_variant.py) the phase cycles of frame 0.   usage: pose_sem_prof.py [B=1024] [objmp_mult=4]"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_slam_amd import synth
from object_slam_amd._lib import check, lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mult = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = lib()
base = []
for i in range(16):
    p = synth.make_semantic_problem(300 + i, N=1000, n_obj=3)
    rng = np.random.default_rng(i)
    p["objmp_Xw"] = np.concatenate([p["objmp_Xw"] + rng.normal(0, 0.01, p["objmp_Xw"].shape).astype(np.float32) for _ in range(mult)])
    p["objmp_obj"] = np.concatenate([p["objmp_obj"]] * mult)
    base.append(p)
cap = 1000
h = C.c_void_p()
check(L.oslam_poseopt_create(C.byref(h), B, cap, 0))
masks_dev = [[torch.from_numpy(m.copy()).cuda() for m in p["masks"]] for p in base]
Tcw = np.stack([base[b % 16]["Tcw"] for b in range(B)]).astype(np.float32)
n = np.full(B, cap, np.int32)
Xw = np.stack([base[b % 16]["Xw"] for b in range(B)]); obs = np.stack([base[b % 16]["obs"] for b in range(B)])
inv = np.stack([base[b % 16]["invSigma2"] for b in range(B)]); has = np.stack([base[b % 16]["has_mp"] for b in range(B)])
fr = np.zeros((B, 6), np.int32)
ptrs, oXw, oObj, jk, jo = [], [], [], [], []
no = nm = nj = 0
for b in range(B):
    p = base[b % 16]
    fr[b] = (len(p["masks"]), no, len(p["objmp_obj"]), nm, len(p["joint_kp"]), nj)
    ptrs += [m.data_ptr() for m in masks_dev[b % 16]]
    oXw.append(p["objmp_Xw"]); oObj.append(p["objmp_obj"]); jk.append(p["joint_kp"]); jo.append(p["joint_obj"])
    no += len(p["masks"]); nm += len(p["objmp_obj"]); nj += len(p["joint_kp"])
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
d = dict(T=t(Tcw), n=t(n), Xw=t(Xw.astype(np.float32)), obs=t(obs.astype(np.float32)), inv=t(inv.astype(np.float32)), has=t(has.astype(np.uint8)), fr=t(fr), ptr=t(np.array(ptrs, np.int64)),
         oXw=t(np.concatenate(oXw).astype(np.float32)), oObj=t(np.concatenate(oObj).astype(np.int32)), jk=t(np.concatenate(jk).astype(np.int32)), jo=t(np.concatenate(jo).astype(np.int32)))
K5 = np.asarray(base[0]["K"], np.float32)
bounds = np.array([0, 0, 640, 480], np.float32)
vp = lambda x: C.c_void_p(x.data_ptr())
def run(nb):
    check(L.oslam_pose_optimize2_batch_device(h, nb, cap, vp(d["n"]), vp(d["T"]), vp(d["Xw"]), vp(d["obs"]), vp(d["inv"]), vp(d["has"]), C.c_void_p(K5.ctypes.data), vp(d["fr"]), no, vp(d["ptr"]),
                                              480, 640, 640, nm, vp(d["oXw"]), vp(d["oObj"]), nj, vp(d["jk"]), vp(d["jo"]), C.c_void_p(bounds.ctypes.data), C.c_float(1.0), None))
print("per frame: object map points %.0f, M_joint candidates %.0f" % (nm / B, nj / B))
for nb in sorted(set([1, 256, B])):
    for _ in range(2): run(nb)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5): run(nb)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    print("pose-opt2 batch %d: %.3f ms/launch, %.2f us/frame" % (nb, dt * 1e3, dt / nb * 1e6))
pS = C.c_void_p()
check(L.oslam_poseopt_semantic_results_device(h, C.byref(pS)))
ns = np.zeros(B, np.int32)
check(L.oslam_memcpy_from_device(C.c_void_p(ns.ctypes.data), pS, C.c_size_t(4 * B)))
print("semantic constraints per frame (nSemNum): mean %.0f" % ns.mean())
if hasattr(L, "oslam_pose_debug_profile"):
    run(1)
    out = (C.c_ulonglong * 8)()
    L.oslam_pose_debug_profile(out)
    names = ["build pass", "sum of 28", "solve + exp", "eval pass", "sum of 1", "classify+sem", "prologue", "total"]
    for n_, v in zip(names, out): print("%-12s %10d cyc %5.1f%%" % (n_, v, 100.0 * v / max(out[7], 1)))
