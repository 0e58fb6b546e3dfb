"""LM iterations/s of the C4 (or C3) bundle adjustment alone, with the stage split -- a quick A / B harness for the BA path:
   python tools/ba_quick.py [C4|C3] [iterations] [views per point]      (environment switches such as CS_BAND_BCR=0 apply)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi, synth_ba

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nc, npt, no = (1000, 200000, 500) if cfg == "C4" else (200, 20000, 50)
views = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pr = synth_ba.make_problem(n_cams=nc, n_points=npt, n_cuboids=no, seed=42, obs_per_point=views)
t0 = time.perf_counter()
P = capi.ba_from_dict(pr)
P.sizes()
t_struct = (time.perf_counter() - t0) * 1e3
P.optimize(1)
t0 = time.perf_counter()
n0 = P.optimize(iters)
el0 = time.perf_counter() - t0
print("%s, stage split off (the default): %d iterations, %.1f it/s, %.3f ms/it" % (cfg, n0, n0 / el0, el0 / max(1, n0) * 1e3))
if len(sys.argv) > 4 and sys.argv[4] == "nosplit":      # (tools/ba_timeline.sh: the timeline of the default mode)
    P.close()
    sys.exit(0)
# the same once more with the stage split on (cs_ba_set_stage_timing: phase marks on the stream, no speculative linearisation), on a fresh handle
P.close()
P = capi.ba_from_dict(pr)
P.stage_timing(True)
P.optimize(1)
tb = P.timing()
t0 = time.perf_counter()
n = P.optimize(iters)
el = time.perf_counter() - t0
ta = P.timing()
d = {k: (ta[k] - tb[k]) / max(1, n) for k in ta if k.endswith("_ms")}
print("views per point <= %d, edges %d, reduced system %s, path %s, Schur layout %s" % (views, len(pr["e_pt"]), P.reduced_size(), P.solver_path(detail=True), P.schur_layout()))
print("%s: %d iterations, %d trials, %.1f it/s, %.3f ms/it; structure %.1f ms; per iteration: %s; chi2 %.6e" %
      (cfg, n, ta["n_solves"] - tb["n_solves"], n / el, el / n * 1e3, t_struct, " ".join("%s %.3f" % (k[:-3], v) for k, v in d.items() if k != "total_ms"), P.history()[0][-1]))
P.close()
