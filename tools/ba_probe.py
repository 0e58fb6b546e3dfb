import sys, time, numpy as np
sys.path.insert(0, '.')
from cube_slam_wu_amd import capi, synth_ba
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
nc, npts, no = (200, 20000, 50) if cfg == "C3" else (1000, 200000, 500)
t = time.time(); pr = synth_ba.make_problem(n_cams=nc, n_points=npts, n_cuboids=no); print("gen %.1fs edges %d cub %d" % (time.time() - t, len(pr['e_pt']), len(pr['ce_cam'])))
t = time.time(); P = capi.ba_from_dict(pr); chi = P.compute_errors(); print("setup+structure %.2fs chi0 %.4g sizes %s" % (time.time() - t, chi, P.sizes()))
t = time.time(); n = P.optimize(10); dt = time.time() - t
print("optimize: %d iters in %.3fs -> %.1f it/s" % (n, dt, n / dt))
print("hist", P.history()[0], P.history()[2])
tm = P.timing(); print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()})
c, o, p = P.state()
print("cam err", np.abs(c[:, :3] - pr['truth']['cams'][:, :3]).max(), "pt err med", np.median(np.abs(p - pr['truth']['points'])))
