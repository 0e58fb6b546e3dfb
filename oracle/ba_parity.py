"""One linearisation of a bundle adjustment, device against oracle, at sizes where the oracle's own unblocked dense LDL^T is out of reach
(C4: 1 000 cameras / 200 000 points / 500 cuboids, a 10 494-unknown reduced system).

TEST INFRASTRUCTURE (same rule as oracle_py.py): used by tests/test_ba_gpu.py and by bench.py's cpu_baseline leg, never by the product.

What is compared, all at the problem's CURRENT estimates:
  chi2        robust chi2 of all edges                               sparse_optimizer.cpp:100-114
  b           the gradient, poses then landmarks                     block_solver.hpp:551-557
  H_ll, H_pl  landmark blocks, camera-landmark blocks of every edge
  S, b_schur  the damped reduced system: the oracle forms g2o's (cameras + cuboids, block_solver.hpp:373-439); when the device also
              eliminated the cuboids (cs_ba_reduced_size) the cuboids' block elimination is applied to the oracle's S with numpy (the 9 x 9
              blocks are decoupled from each other), every block then compared in the cameras' own order
  x           one damped solve: the oracle's S factorised by LAPACK (scipy.linalg.cho_factor -- the oracle's textbook LDL^T would stream
              the 881 MB matrix once per column), the oracle's own landmark back-substitution (block_solver.hpp:457-482)
Returns max relative differences (max |device - oracle| over max |oracle|, per quantity; S additionally per 6 x 6 block).
"""
from __future__ import annotations

import numpy as np


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    d = np.abs(b).max() if b.size else 0.0
    return float(np.abs(a - b).max() / d) if d > 0 else float(np.abs(a - b).max() if a.size else 0.0)


def compare_linearisation(G, R, pr, lam, with_solve=True, one_thread=False):
    """G: cube_slam_wu_amd.capi.BaProblem, R: oracle.ba_oracle_py.Problem on the same problem dict pr (both at the same estimates).
    one_thread: factorise the oracle's S on ONE thread and report the seconds (bench.py's cpu_baseline prices the reference's dense
    solve with it: LAPACK's blocked dpotrf on the real 10 494-unknown matrix, measured instead of extrapolated)."""
    import contextlib
    import time
    import scipy.linalg

    out = {}
    chi_r = R.compute_errors()[0]
    out["chi2"] = abs(G.compute_errors() - chi_r) / chi_r
    _, Hll_g, Hpl_g, b_g = G.build_system(dense_hpp=False)
    Hll_r, Hpl_r, b_r = R.build_system_blocks()
    out["b"] = _rel(b_g, b_r)
    out["H_ll"] = _rel(Hll_g, Hll_r)
    out["H_pl"] = _rel(Hpl_g, Hpl_r)
    nc, no = len(pr["cams"]), len(pr["cuboids"])
    cam_fixed, cub_fixed = np.asarray(pr["cam_fixed"]) != 0, np.asarray(pr["cub_fixed"]) != 0
    # g2o's order: free cameras (6 each), then free cuboids (9 each)
    ref_cam = np.full(nc, -1); ref_cub = np.full(no, -1)
    col = 0
    for i in range(nc):
        if not cam_fixed[i]:
            ref_cam[i] = col; col += 6
    for i in range(no):
        if not cub_fixed[i]:
            ref_cub[i] = col; col += 9
    n_pose = col
    S_r, bs_r = R.schur(lam)
    assert S_r.shape == (n_pose, n_pose)
    S_g, rhs_g, cam_col, cub_col = G.reduced_system(lam)
    n_red, elim = G.reduced_size()
    cam_idx = np.concatenate([np.arange(ref_cam[i], ref_cam[i] + 6) for i in range(nc) if ref_cam[i] >= 0]) if (~cam_fixed).any() else np.zeros(0, int)
    if elim:
        # the device eliminated the free cuboids as well: S_cc -= S_co S_oo^-1 S_oc, b_c -= S_co S_oo^-1 b_o, cuboid by cuboid (a cuboid
        # is coupled to its observing cameras only -- the rows of its column block that are non-zero)
        S_cmp = S_r.copy(); b_cmp = bs_r.copy()
        for o in range(no):
            if ref_cub[o] < 0:
                continue
            sl = slice(ref_cub[o], ref_cub[o] + 9)
            rows = cam_idx[np.any(S_cmp[cam_idx, sl] != 0, axis=1)]
            Mo = S_cmp[np.ix_(rows, np.arange(sl.start, sl.stop))]
            Di = np.linalg.inv(S_cmp[sl, sl])
            S_cmp[np.ix_(rows, rows)] -= Mo @ Di @ Mo.T
            b_cmp[rows] -= Mo @ (Di @ b_cmp[sl])
        ref_of, dev_of = [], []
        for i in range(nc):
            if ref_cam[i] >= 0:
                ref_of.append(np.arange(ref_cam[i], ref_cam[i] + 6)); dev_of.append(np.arange(cam_col[i], cam_col[i] + 6))
    else:
        S_cmp, b_cmp = S_r, bs_r
        ref_of, dev_of = [], []
        for i in range(nc):
            if ref_cam[i] >= 0:
                ref_of.append(np.arange(ref_cam[i], ref_cam[i] + 6)); dev_of.append(np.arange(cam_col[i], cam_col[i] + 6))
        for i in range(no):
            if ref_cub[i] >= 0:
                ref_of.append(np.arange(ref_cub[i], ref_cub[i] + 9)); dev_of.append(np.arange(cub_col[i], cub_col[i] + 9))
    ref_of, dev_of = np.concatenate(ref_of), np.concatenate(dev_of)
    assert len(dev_of) == n_red and sorted(dev_of.tolist()) == list(range(n_red))
    S_ref = S_cmp[np.ix_(ref_of, ref_of)]
    S_dev = S_g[np.ix_(dev_of, dev_of)]
    out["S"] = _rel(S_dev, S_ref)
    out["b_schur"] = _rel(rhs_g[dev_of], b_cmp[ref_of])
    # ... and block by block (6 x 6 tiles of the permuted matrices; a block the oracle holds as exact zero must be zero on the device)
    nb = n_red // 6 if elim or no == 0 else 0
    if nb:
        T_ref = np.abs(S_ref[:6 * nb, :6 * nb]).reshape(nb, 6, nb, 6).max(axis=(1, 3))
        T_dif = np.abs(S_dev[:6 * nb, :6 * nb] - S_ref[:6 * nb, :6 * nb]).reshape(nb, 6, nb, 6).max(axis=(1, 3))
        nz = T_ref > 0
        out["S_worst_block"] = float((T_dif[nz] / T_ref[nz]).max())
        out["S_blocks_compared"] = int(nz.sum())
        out["S_nonzero_where_oracle_is_zero"] = float(T_dif[~nz].max()) if (~nz).any() else 0.0
    del S_ref, S_dev, S_g
    if with_solve:
        ok_g, x_g = G.solve(lam)
        limit = contextlib.nullcontext()
        if one_thread:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1)
        with limit:
            t0 = time.perf_counter()
            c, low = scipy.linalg.cho_factor(S_r, lower=True, overwrite_a=True, check_finite=False)
            xp = scipy.linalg.cho_solve((c, low), bs_r, check_finite=False)
            out["dense_solve_seconds"] = time.perf_counter() - t0
            out["dense_solve_threads"] = 1 if one_thread else 0      # 0 = LAPACK's default thread count
            out["dense_solve_unknowns"] = int(n_pose)
        x_r = R.backsub(lam, xp)
        out["solve_ok"] = bool(ok_g)
        out["x_pose"] = _rel(x_g[:n_pose], x_r[:n_pose])
        out["x_landmarks"] = _rel(x_g[n_pose:], x_r[n_pose:])
    return out



def compare_trajectory(G, R, iters=3):
    """iters LM iterations of the device (G.optimize) beside the same of the oracle (R.optimize; for C4 with R.use_lapack_solver(): the
    10 494-unknown dense solve through LAPACK, everything else of the iteration ba_oracle.cpp's -- optimization_algorithm_levenberg.cpp:61-189
    restated at ba_oracle.cpp lm_solve).  north_star's bar: the same accept / reject sequence, states within 1e-5 relative after the same
    iteration count.  Returns the trial sequences, max relative differences of the chi2 / lambda histories and of the final states
    (positions relative to the scene's extent, quaternions and cuboid half sizes absolute)."""
    import time
    t0 = time.perf_counter()
    n_g = G.optimize(iters)
    t1 = time.perf_counter()
    n_r = R.optimize(iters)
    t2 = time.perf_counter()
    chi_g, lam_g, tr_g = G.history()
    chi_r, lam_r, tr_r = R.history()
    cg, og, pg = G.state()
    cr, orr, prr = R.state()
    scale = float(np.abs(prr).max()) if prr.size else 1.0
    out = {"iterations_device": int(n_g), "iterations_oracle": int(n_r), "trials_device": [int(t) for t in tr_g], "trials_oracle": [int(t) for t in tr_r],
           "same_trial_sequence": bool(np.array_equal(tr_g, tr_r)),
           "chi2": float(np.abs(chi_g - chi_r).max() / np.abs(chi_r).max()) if len(chi_g) == len(chi_r) and len(chi_r) else float("nan"),
           "lambda": float(np.abs(lam_g / lam_r - 1).max()) if len(lam_g) == len(lam_r) and len(lam_r) else float("nan"),
           "chi2_first_last": [float(chi_r[0]), float(chi_r[-1])] if len(chi_r) else [],
           "points": float(np.abs(pg - prr).max() / scale) if prr.size else 0.0,
           "camera_positions": float(np.abs(cg[:, :3] - cr[:, :3]).max() / scale), "camera_quaternions": float(np.abs(cg[:, 3:] - cr[:, 3:]).max()),
           "cuboid_positions": float(np.abs(og[:, :3] - orr[:, :3]).max() / scale) if orr.size else 0.0,
           "cuboid_quaternions_and_sizes": float(np.abs(og[:, 3:] - orr[:, 3:]).max()) if orr.size else 0.0,
           "device_seconds": t1 - t0, "oracle_seconds": t2 - t1}
    return out
