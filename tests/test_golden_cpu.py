"""CPU: the oracle reproduces the committed golden fixtures (tests/golden/gen_golden.py)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_frontend_golden(oracle):
    z = np.load(os.path.join(G, "frontend_320x240.npz"))
    e = oracle.OrbExtractor(300, 1.2, 5, 20, 7)
    for fr, k, d in ((z["frame0"], z["k0"], z["d0"]), (z["frame1"], z["k1"], z["d1"])):
        kk, dd = e.extract(fr)
        assert kk.tobytes() == k.tobytes()
        np.testing.assert_array_equal(dd, d)
    q = oracle.project_last_frame(z["Xw"], z["has"], z["k0"], z["d0"], z["Tcw"], z["Tlw"], z["cam"], z["bounds"],
                                  z["scale"], 15.0, False)
    assert q.tobytes() == z["queries"].tobytes()
    nm, qm, qd, km = oracle.search_by_projection(z["k1"], z["uR"], z["d1"], None, z["bounds"], q, 0.9, False, True)
    assert nm == int(z["nm"]) and nm > 100
    np.testing.assert_array_equal(qm, z["qm"]); np.testing.assert_array_equal(qd, z["qd"]); np.testing.assert_array_equal(km, z["km"])
    nm2, qm2, qd2, km2 = oracle.search_by_projection(z["k1"], z["uR"], z["d1"], None, z["bounds"], q, 0.8, True, False)
    assert nm2 == int(z["nm2"])
    np.testing.assert_array_equal(qm2, z["qm2"]); np.testing.assert_array_equal(km2, z["km2"])
