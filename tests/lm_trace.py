"""Comparison of two Levenberg-Marquardt trial traces (HIP kernel vs CPU oracle), shared by the pose-optimisation and local-BA parity tests."""


def compare_lm_traces(th, to, rho_tol=1e-6):
    """The Levenberg-Marquardt schedule of the HIP path against the oracle's, trial by trial: the same accept / reject decision, the same damping
    (lambda follows from the decisions and rho, so it checks both) and the same cost, for every trial whose decision is decidable — i.e. until the
    first trial where |rho| is below rho_tol in either trace (F0 - F1 is then rounding noise of the fp64 sums, whose order differs between a
    parallel reduction and the oracle's index order, and the reference's `rho > 0` test flips on it).  Returns (trials compared, undecidable
    trials met): after an undecidable trial the two runs may take different but equally valid branches (ten rejected trials vs an early exit), so
    the comparison restarts at the next round (which starts from the shared start pose with lambda re-initialised)."""
    i = j = compared = undecidable = 0
    while i < len(th) and j < len(to):
        h, o = th[i], to[j]
        if abs(h[2]) < rho_tol or abs(o[2]) < rho_tol:
            undecidable += 1
            i = next((k for k in range(i + 1, len(th)) if th[k][5] == 1), len(th))   # both restart at the next round's first trial
            j = next((k for k in range(j + 1, len(to)) if to[k][5] == 1), len(to))
            continue
        assert h[4] == o[4], ("accept/reject differs at a decidable trial", i, j, h, o)
        assert abs(h[3] - o[3]) <= 1e-4 * abs(o[3]) + 1e-12, ("lambda differs", i, j, h, o)
        assert abs(h[0] - o[0]) <= 1e-7 * abs(o[0]) + 1e-9 and abs(h[1] - o[1]) <= 1e-7 * abs(o[1]) + 1e-9, ("cost differs", i, j, h, o)
        assert abs(h[2] - o[2]) <= 1e-4 * max(1.0, abs(o[2])), ("rho differs", i, j, h, o)
        assert h[5] == o[5], ("round boundary differs", i, j, h, o)
        compared += 1
        i += 1; j += 1
    return compared, undecidable
