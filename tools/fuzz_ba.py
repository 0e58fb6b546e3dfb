"""Differential run of the BA path against its CPU restatement on random problem shapes: same accept / reject sequence, chi2 history within
1e-6 relative, every state within 1e-5 relative (BASELINE.json).  python tools/fuzz_ba.py [problems]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi, synth_ba
from oracle import ba_oracle_py as O

n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(99)
bad = 0
t0 = time.time()
for k in range(n_prob):
    nc = int(rng.integers(6, 140)); npnt = int(rng.integers(60, 3000)); no = int(rng.integers(0, 9)); huber = bool(rng.integers(0, 2))
    pr = synth_ba.make_problem(n_cams=nc, n_points=npnt, n_cuboids=no, seed=int(rng.integers(1, 10**6)), huber=huber)
    cf = bool(rng.integers(0, 2))
    pr["cam_fixed"] = pr["cam_fixed"].copy(); pr["pt_fixed"] = pr["pt_fixed"].copy(); pr["cub_fixed"] = pr["cub_fixed"].copy()
    if rng.uniform() < 0.4:
        pr["cam_fixed"][rng.integers(0, nc, size=max(1, nc // 10))] = 1
    if rng.uniform() < 0.4:
        pr["pt_fixed"][:: int(rng.integers(3, 11))] = 1
    if no and rng.uniform() < 0.3:
        pr["cub_fixed"][int(rng.integers(0, no))] = 1
    # (half of the problems: per-edge information matrices / a second camera's intrinsics -- the handle then reads the per-edge records;
    # the others share them and take the constants, BaView::info_u / intr_u)
    if rng.uniform() < 0.5:
        pr["e_info"] = pr["e_info"] * (1.2 ** -rng.integers(0, 8, len(pr["e_pt"])))[:, None]
    if rng.uniform() < 0.3:
        second = rng.random(len(pr["e_pt"])) < 0.3
        pr["e_intr"] = np.where(second[:, None], pr["e_intr"] * np.array([1.01, 0.99, 1.0, 1.0]), pr["e_intr"])
    iters = int(rng.integers(3, 9))
    G = capi.ba_from_dict(pr, cuboids_first=cf)
    R = O.Problem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"], cuboids_first=cf)
    R.set_edges_proj(pr["e_pt"], pr["e_cam"], pr["e_uv"], pr["e_info"], pr["e_intr"], pr["e_huber"])
    if len(pr["ce_cam"]): R.set_edges_cuboid(pr["ce_cam"], pr["ce_cub"], pr["ce_meas"], pr["ce_info"])
    if len(pr["oe_i"]): R.set_edges_odom(pr["oe_i"], pr["oe_j"], pr["oe_meas"], pr["oe_info"])
    ng, nr = G.optimize(iters), R.optimize(iters)
    cg, lg, tg = G.history(); cr, lr, tr = R.history()
    sg, sr = G.state(), R.state()
    scale = max(1.0, float(np.abs(sr[2]).max()))
    ok = ng == nr and np.array_equal(tg, tr) and np.allclose(cg, cr, rtol=1e-6)
    ok = ok and np.abs(sg[2] - sr[2]).max() < 1e-5 * scale and np.abs(sg[0][:, :3] - sr[0][:, :3]).max() < 1e-5 * scale and np.abs(sg[0][:, 3:] - sr[0][:, 3:]).max() < 1e-5
    if no:
        ok = ok and np.abs(sg[1][:, :3] - sr[1][:, :3]).max() < 1e-5 * scale and np.abs(sg[1][:, 3:] - sr[1][:, 3:]).max() < 1e-5
    if not ok:
        bad += 1
        print("mismatch", k, dict(nc=nc, np=npnt, no=no, huber=huber, cuboids_first=cf, iters=iters), ng, nr, list(tg), list(tr))
    G.close(); R.close()
print("%d problems, %d mismatches, %.0f s" % (n_prob, bad, time.time() - t0))
sys.exit(1 if bad else 0)
