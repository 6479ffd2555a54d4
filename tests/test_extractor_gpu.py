"""GPU parity: HIP ORB extractor (through the C ABI) vs the CPU oracle, stage by stage and end to
end.  Bit-exact: pyramid bytes, blurred bytes, FAST candidates (order, coords, score), quad-tree
survivors (order), keypoints (all 7 fields) and descriptors."""
import numpy as np
import pytest

from object_slam_amd import ORBextractor, synth

pytestmark = pytest.mark.gpu

TUM = dict(nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)
KITTI = dict(nfeatures=2000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)


def _stages(oracle, cfg, img):
    h, w = img.shape
    ex = ORBextractor(width=w, height=h, **cfg)
    oe = oracle.OrbExtractor(cfg["nfeatures"], cfg["scaleFactor"], cfg["nlevels"], cfg["iniThFAST"], cfg["minThFAST"])
    kps, desc = ex(img)
    okps, odesc = oe.extract(img)
    for l in range(cfg["nlevels"]):
        assert ex.level_size(l) == oe.level_size(l)
        np.testing.assert_array_equal(ex.pyramid_level(l), oe.level(l), err_msg="pyramid level %d" % l)
        cand = ex.debug_candidates(l)
        oc = oe.candidates(l)
        assert len(cand) == len(oc), "level %d: %d vs %d candidates" % (l, len(cand), len(oc))
        np.testing.assert_array_equal(cand[:, 0], oc["x"].astype(np.int32))
        np.testing.assert_array_equal(cand[:, 1], oc["y"].astype(np.int32))
        np.testing.assert_array_equal(cand[:, 2], oc["response"].astype(np.int32))
        keys = ex.debug_level_keys(l)
        ok = oe.level_keys(l)
        assert len(keys) == len(ok), "level %d: %d vs %d survivors" % (l, len(keys), len(ok))
        np.testing.assert_array_equal(keys[:, 0], ok["x"].astype(np.int32))
        np.testing.assert_array_equal(keys[:, 1], ok["y"].astype(np.int32))
        ob = oe.blurred(l)
        if ob is not None:
            np.testing.assert_array_equal(ex.debug_blurred(l), ob, err_msg="blur level %d" % l)
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        np.testing.assert_array_equal(kps[f], okps[f], err_msg=f)
    np.testing.assert_array_equal(desc, odesc)
    ex.close()
    return len(kps)


def test_tum_shape_stage_parity(oracle):
    frames, _ = synth.make_stream(3, 640, 480)
    for i in range(3):
        n = _stages(oracle, TUM, frames[i])
        assert 900 <= n <= 1100


def test_kitti_shape_stage_parity(oracle):
    """Full-size C-KITTI stage parity (pyramid, FAST cells, quad-tree, orientation, blur, descriptors) on 8 frames: 3 crops of the planar canvas and 5
    frames (left and right cameras) of the street scene, whose facades at 5 - 60 m fill the pyramid levels unevenly."""
    from object_slam_amd import scene
    frames, _ = synth.make_stream(3, 1241, 376, seed=7)
    st = scene.make_stereo_sequence(2, 3)
    imgs = [frames[0], frames[1], frames[2], st["gray"][0], st["right"][0], st["gray"][1], st["gray"][2], st["right"][2]]
    for i, im in enumerate(imgs):
        n = _stages(oracle, KITTI, im)
        assert (1800 if i < 3 else 1200) <= n <= 2100, (i, n)


def test_odd_sizes_and_flat_images(oracle):
    rng = np.random.default_rng(5)
    # small odd geometry, low-texture image (forces the minThFAST fallback and empty cells)
    img = (rng.integers(0, 6, size=(211, 317)) + 100).astype(np.uint8)
    img[60:120, 80:200] = 140
    _stages(oracle, dict(nfeatures=300, scaleFactor=1.2, nlevels=4, iniThFAST=20, minThFAST=7), img)
    # constant image: no keypoints at all
    ex = ORBextractor(500, 1.2, 8, 20, 7, 640, 480)
    k, d = ex(np.full((480, 640), 128, np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    # empty image: silent return like the reference (src/ORBextractor.cc:1046)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0
    ex.close()


def test_noise_image_many_candidates(oracle):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(480, 640)).astype(np.uint8)
    _stages(oracle, TUM, img)


def test_batch_equals_single(oracle):
    import torch
    frames, _ = synth.make_stream(6, 640, 480, seed=3)
    ex = ORBextractor(width=640, height=480, max_batch=6, **TUM)
    d = torch.from_numpy(frames).cuda()
    ex.extract_batch_device(d.data_ptr(), 6, 640, 640 * 480, torch.cuda.current_stream().cuda_stream)
    oe = oracle.OrbExtractor()
    for b in range(6):
        k, de = ex.fetch(b)
        ok, od = oe.extract(frames[b])
        assert len(k) == len(ok)
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            np.testing.assert_array_equal(k[f], ok[f])
        np.testing.assert_array_equal(de, od)
    ex.close()


def test_golden_fixture_frontend():
    """HIP path vs the committed golden vectors (tests/golden/gen_golden.py), no oracle involved."""
    import os
    from object_slam_amd import ORBmatcher
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_320x240.npz"))
    ex = ORBextractor(300, 1.2, 5, 20, 7, 320, 240)
    k0, d0 = ex(z["frame0"])
    k1, d1 = ex(z["frame1"])
    assert k0.tobytes() == z["k0"].tobytes() and k1.tobytes() == z["k1"].tobytes()
    np.testing.assert_array_equal(d0, z["d0"])
    np.testing.assert_array_equal(d1, z["d1"])
    m = ORBmatcher(0.9, True, max_keypoints=ex.cap, max_queries=ex.cap)
    nm, qm, qd, km = m.search_last_frame(k1, z["uR"], d1, None, tuple(z["bounds"]), z["Xw"], z["has"], k0, d0, z["Tcw"],
                                         z["Tlw"], tuple(z["cam"]), z["scale"], 15.0, False)
    assert nm == int(z["nm"])
    np.testing.assert_array_equal(qm, z["qm"]); np.testing.assert_array_equal(qd, z["qd"]); np.testing.assert_array_equal(km, z["km"])
    m2 = ORBmatcher(0.8, True, max_keypoints=ex.cap, max_queries=ex.cap)
    nm2, qm2, qd2, km2 = m2.search_window(k1, z["uR"], d1, None, tuple(z["bounds"]), z["queries"], True, False)
    assert nm2 == int(z["nm2"])
    np.testing.assert_array_equal(qm2, z["qm2"]); np.testing.assert_array_equal(km2, z["km2"])


def test_fast_worklist_overflow_cells(oracle):
    """A 3x3-block checkerboard (plus noise so that scores differ) makes almost every pixel pass FAST's quick test: the wavefront kernel's 880-entry
    LDS worklist overflows in nearly every cell and the cells are redone by k_fast_cells_ovf.  Candidates, keypoints and descriptors stay bit-exact."""
    rng = np.random.default_rng(11)
    h, w = 240, 320
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.where(((xx // 3 + yy // 3) & 1) == 0, 60, 190).astype(np.int32) + rng.integers(-25, 26, size=(h, w))
    img = np.clip(img, 0, 255).astype(np.uint8)
    n = _stages(oracle, TUM, img)
    assert n > 500


def test_large_quota_on_few_levels_node_tables_in_hbm(oracle):
    """2600 features on one or two levels: the quad-tree's node tables (88 B per node) no longer fit the 160 KB of LDS, the kernel then
    keeps them (and the candidate list) in HBM; same algorithm, same result."""
    from object_slam_amd import synth
    img = synth.make_stream(1, 640, 480, seed=77)[0][0]
    for nl in (1, 2):
        n = _stages(oracle, dict(nfeatures=2600, scaleFactor=1.2, nlevels=nl, iniThFAST=20, minThFAST=7), img)
        assert n > 1200, n


def test_experiment_knobs_keep_results(oracle, monkeypatch):
    """Off-by-default stream layouts (level-0 FAST beside the pyramid kernels, batches cut in two halves on two stream pairs, resize without the
    LDS-staged kernel) must not change a bit of the output."""
    from object_slam_amd import synth
    import torch
    B = 6
    frames, _ = synth.make_stream(B, 640, 480, seed=31)
    d = torch.from_numpy(frames).cuda()
    st = torch.cuda.current_stream().cuda_stream

    def run():
        ex = ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, max_batch=B)
        ex.extract_batch_device(d.data_ptr(), B, 640, 640 * 480, st)
        torch.cuda.synchronize()
        out = [ex.fetch(b) for b in range(B)]
        ex.close()
        return out

    ref = run()
    for env in ({"OSLAM_ORB_FAST0_STREAM": "1"}, {"OSLAM_ORB_SPLIT_MIN": "2"}, {"OSLAM_ORB_NO_LDS_RESIZE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k)
        for (ka, da), (kb, db) in zip(ref, got):
            assert ka.tobytes() == kb.tobytes() and np.array_equal(da, db), env
