"""Shared helpers of the tracking-driver tests: the product driver (object_slam_amd/csrc/slam_driver.hip) is run over an
operator table — the HIP one (product) or the CPU oracle's (oracle/slam_ops_oracle.cc, test infrastructure)."""
import ctypes as C

import numpy as np

from object_slam_amd import slam, synth
from object_slam_amd.io import horn_align_ate

W, H, Z0 = 640, 480, 2.0


def oracle_ops(cfg):
    from oracle import oracle_py as O
    ops = slam.SlamOps()
    assert O.lib().oo_slam_make_ops(C.byref(cfg), C.byref(ops)) == 0
    return ops


def make_streams(S, n, margin=1200, seed0=11):
    return [synth.make_stream(n, W, H, seed=seed0 + s, margin=margin) for s in range(S)]


def run(system, streams, n):
    S = len(streams)
    depth = np.full((H, W), Z0, np.float32)
    poses, states = [], []
    for t in range(n):
        T, st = system.TrackRGBD([streams[s][0][t] for s in range(S)], [depth] * S, [t / 30.0] * S)
        poses.append(T.copy())
        states.append(st.copy())
    return np.array(poses), np.array(states)


def ate(system, cfg, streams, s):
    stamps, Twc = system.trajectory(s)
    off = (streams[s][1] - streams[s][1][0]).astype(np.float64)
    gt = np.stack([off[:, 0] * Z0 / cfg.fx, off[:, 1] * Z0 / cfg.fy, np.zeros(len(off))], 1)
    return horn_align_ate(Twc[:, :, 3], gt[:len(stamps)]), Twc


KW, KH, KDISP = 1241, 376, 32


def make_stereo_streams(S, n, margin=600, seed0=21):
    return [synth.make_stereo_stream(n, KW, KH, seed=seed0 + s, margin=margin, disparity=KDISP) for s in range(S)]


def stereo_config(S, **kw):
    return slam.make_config(KW, KH, S, cam=slam.KITTI00, nFeatures=2000, iniThFAST=20, minThFAST=7, sensor=slam.STEREO, **kw)


def run_stereo(system, streams, n):
    S = len(streams)
    poses, states = [], []
    for t in range(n):
        T, st = system.TrackStereo([streams[s][0][t] for s in range(S)], [streams[s][1][t] for s in range(S)], [t / 10.0] * S)
        poses.append(T.copy())
        states.append(st.copy())
    return np.array(poses), np.array(states)


def ate_stereo(system, cfg, streams, s):
    z0 = cfg.bf / KDISP
    stamps, Twc = system.trajectory(s)
    off = (streams[s][2] - streams[s][2][0]).astype(np.float64)
    gt = np.stack([off[:, 0] * z0 / cfg.fx, off[:, 1] * z0 / cfg.fy, np.zeros(len(off))], 1)
    return horn_align_ate(Twc[:, :, 3], gt[:len(stamps)]), Twc


# ---- 3-D scenes (object_slam_amd/scene.py): SE3 motion with rotation, depth variation, object masks ----
def make_scene_streams(S, n, seed0=0, speed=2.0):
    from object_slam_amd import scene
    return [scene.make_rgbd_sequence(seed0 + s, n, speed=speed) for s in range(S)]


def run_scene(system, seqs, n):
    S = len(seqs)
    poses, states = [], []
    for t in range(n):
        T, st = system.TrackRGBD([seqs[s]["gray"][t] for s in range(S)], [seqs[s]["depth"][t] for s in range(S)], [t / 30.0] * S)
        poses.append(T.copy())
        states.append(st.copy())
    return np.array(poses), np.array(states)


def ate_scene(system, seqs, s):
    stamps, Twc = system.trajectory(s)
    return horn_align_ate(Twc[:, :, 3], seqs[s]["Twc"][:len(stamps), :3, 3]), Twc


def make_scene_stereo(S, n, seed0=0):
    from object_slam_amd import scene
    return [scene.make_stereo_sequence(seed0 + s, n) for s in range(S)]


def run_scene_stereo(system, seqs, n):
    S = len(seqs)
    poses, states = [], []
    for t in range(n):
        T, st = system.TrackStereo([seqs[s]["gray"][t] for s in range(S)], [seqs[s]["right"][t] for s in range(S)], [t / 10.0] * S)
        poses.append(T.copy())
        states.append(st.copy())
    return np.array(poses), np.array(states)
