"""Array-form model of the quad-tree distribution exactly as the HIP kernel k_octree performs it
(object_slam_amd/csrc/orb_kernels.hip).  Used on CPU to prove the list-order reformulation equal
to the oracle's literal std::list restatement of reference src/ORBextractor.cc:539-763."""
import numpy as np


def child_of(kx, ky, x0, x1, y0, y1):
    hx = (x1 - x0 + 1) >> 1
    hy = (y1 - y0 + 1) >> 1
    return (0 if kx < x0 + hx else 1) + 2 * (0 if ky < y0 + hy else 1)


def distribute(xs, ys, resp, region_w, region_h, N):
    """xs, ys: region coords (ints); returns indices of survivors in reference order."""
    n = len(xs)
    nIni = int(np.round(np.float32(region_w) / np.float32(region_h)))
    hX = np.float32(region_w) / np.float32(nIni)
    rootx = [int(np.float32(hX) * np.float32(i)) for i in range(nIni + 1)]
    root_of = [min(int(np.float32(x) / hX), nIni - 1) for x in xs]
    cnt0 = [0] * nIni
    for r in root_of:
        cnt0[r] += 1
    nodes = []  # dict per list position
    remap = {}
    for i in range(nIni):
        if cnt0[i] > 0:
            remap[i] = len(nodes)
            nodes.append(dict(x0=rootx[i], x1=rootx[i + 1], y0=0, y1=region_h, cnt=cnt0[i], seq=i))
    knode = [remap[r] for r in root_of]
    careful = False
    for _ in range(64):
        S = len(nodes)
        childcnt = [[0, 0, 0, 0] for _ in range(S)]
        kch = [0] * n
        for i in range(n):
            p = knode[i]
            nd = nodes[p]
            if nd["cnt"] > 1:
                c = child_of(xs[i], ys[i], nd["x0"], nd["x1"], nd["y0"], nd["y1"])
                kch[i] = c
                childcnt[p][c] += 1
        div = [p for p in range(S) if nodes[p]["cnt"] > 1]
        if not careful:
            byrank = div
        else:
            byrank = sorted(div, key=lambda p: (nodes[p]["cnt"], nodes[p]["seq"]), reverse=True)
            size = S
            M = len(byrank)
            for j, p in enumerate(byrank):
                size += sum(1 for c in childcnt[p] if c > 0) - 1
                if size >= N:
                    M = j + 1
                    break
            byrank = byrank[:M]
        prank = {p: j for j, p in enumerate(byrank)}
        front = []
        childpos = {}
        for j in reversed(range(len(byrank))):
            p = byrank[j]
            nd = nodes[p]
            hx = nd["x0"] + ((nd["x1"] - nd["x0"] + 1) >> 1)
            hy = nd["y0"] + ((nd["y1"] - nd["y0"] + 1) >> 1)
            for ch in (3, 2, 1, 0):
                cc = childcnt[p][ch]
                if cc > 0:
                    childpos[(p, ch)] = len(front)
                    front.append(dict(x0=hx if ch & 1 else nd["x0"], x1=nd["x1"] if ch & 1 else hx,
                                      y0=hy if ch & 2 else nd["y0"], y1=nd["y1"] if ch & 2 else hy,
                                      cnt=cc, seq=j * 4 + ch))
        stay = {}
        rest = []
        for p in range(S):
            if p not in prank:
                stay[p] = len(front) + len(rest)
                rest.append(nodes[p])
        nexp = sum(1 for nd in front if nd["cnt"] > 1)
        for i in range(n):
            p = knode[i]
            knode[i] = childpos[(p, kch[i])] if p in prank else stay[p]
        nodes = front + rest
        newS = len(nodes)
        if newS >= N or newS == S:
            break
        if not careful and newS + 3 * nexp > N:
            careful = True
    best = [-1] * len(nodes)
    for i in range(n):
        key = (int(resp[i]) << 14) | (16383 - i)
        if key > best[knode[i]]:
            best[knode[i]] = key
    return [16383 - (b & 16383) for b in best]
