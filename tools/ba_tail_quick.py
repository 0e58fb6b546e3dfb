"""A map with a TAIL of long tracks: C4's trajectory (1 000 cameras, 200 000 landmarks of <= 5 views) where 2 % of the landmarks are
seen from up to 25 cameras.  Before round 4's ba_schur_long_kernel one such landmark sent the whole problem to the pair-major Schur build
(and kept the cuboids in the reduced system):   python tools/ba_tail_quick.py      (CS_BA_SCHUR_PAIRS=1: that path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi, synth_ba
a = synth_ba.make_problem(n_cams=1000, n_points=196000, n_cuboids=500, seed=42)
b = synth_ba.make_problem(n_cams=1000, n_points=4000, n_cuboids=0, seed=43, obs_per_point=25)
pr = dict(a)
off = len(a["points"])
pr["points"] = np.concatenate([a["points"], b["points"]]); pr["pt_fixed"] = np.concatenate([a["pt_fixed"], b["pt_fixed"]])
for k in ("e_cam", "e_uv", "e_info", "e_intr", "e_huber"):
    pr[k] = np.concatenate([a[k], b[k]])
pr["e_pt"] = np.concatenate([a["e_pt"], b["e_pt"] + off]).astype(np.int32)
P = capi.ba_from_dict(pr)
P.stage_timing(True)
t0 = time.perf_counter(); P.sizes(); t_struct = (time.perf_counter() - t0) * 1e3
print("edges %d, longest track %d, reduced system %s, Schur layout %s, structure %.0f ms" % (len(pr["e_pt"]), np.bincount(pr["e_pt"]).max(), P.reduced_size(), P.schur_layout(), t_struct))
P.optimize(1)
tb = P.timing(); t0 = time.perf_counter(); n = P.optimize(10); el = time.perf_counter() - t0; ta = P.timing()
d = {k: (ta[k] - tb[k]) / max(1, n) for k in ta if k.endswith("_ms")}
print("%d iterations, %.1f it/s, %.3f ms/it; per iteration: %s; chi2 %.6e" % (n, n / el, el / n * 1e3, " ".join("%s %.3f" % (k[:-3], v) for k, v in d.items() if k != "total_ms"), P.history()[0][-1]))
