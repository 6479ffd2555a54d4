"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU-oracle backend for the end-to-end harness
(object_slam_amd/e2e.py): same call surface as e2e.HipBackend, every operator is the oracle's."""
import numpy as np

from . import oracle_py as O


class OracleBackend:
    def __init__(self, width, height, nfeatures=1000, nlevels=8):
        self.ex = O.OrbExtractor(nfeatures, 1.2, nlevels, 20, 7)
        t = self.ex.tables()
        self.scale, self.inv_sigma2 = t["scale"], t["inv_sigma2"]

    def extract(self, img):
        return self.ex.extract(img)

    def search_last(self, kc, uR, dc, bounds, Xw, has, kl, dl, Tcw, Tlw, cam, th):
        q = O.project_last_frame(Xw, has, kl, dl, Tcw, Tlw, cam, bounds, self.scale, th, False)
        nm, qm, qd, km = O.search_by_projection(kc, uR, dc, None, bounds, q, 0.9, False, True)
        return nm, km

    def search_map(self, kc, uR, dc, blocked, bounds, queries):
        nm, qm, qd, km = O.search_by_projection(kc, uR, dc, blocked, bounds, queries, 0.8, True, False)
        return nm, km

    def pose_opt(self, Tcw, Xw, obs, inv, has, K5):
        n, T, outl, _ = O.pose_optimization(Tcw, Xw, obs, inv, has, K5)
        return n, T, outl

    def lba(self, poses, fixed, points, ekf, ept, eobs, einv, K5):
        po, xo, er, _ = O.local_bundle_adjustment(poses, fixed, points, ekf, ept, eobs, einv, K5)
        return po, xo, er
