"""The general sparse reduced solve against the banded / dense paths on a survey-flight (2-D covisibility mesh) graph:
   python tools/ba_mesh_quick.py [nx ny n_points]      (CS_BA_SPARSE=1 / 0 selects the path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cube_slam_wu_amd import capi, synth_ba
nx, ny, npt = (int(a) for a in (sys.argv[1:4] + ["16", "16", "20000"][len(sys.argv) - 1:]))
pr = synth_ba.make_mesh_problem(nx, ny, npt)
G = capi.ba_from_dict(pr)
t0 = time.perf_counter(); n_red, elim = G.reduced_size(); t1 = time.perf_counter()
print("cams %d points %d edges %d  reduced system %d  structure %.1f ms  path %s" % (len(pr["cams"]), len(pr["points"]), len(pr["e_pt"]), n_red, (t1 - t0) * 1e3, G.solver_path(detail=True)))
G.compute_errors(); G.build_system(dense_hpp=False)
ok, x = G.solve(1e-3)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); ok, x = G.solve(1e-3); ts.append(time.perf_counter() - t0)
print("solve ok %s  |x| %.6e  reduce+factor+substitute %.3f ms (min of 5)" % (ok, np.abs(x).max(), min(ts) * 1e3))
n = G.optimize(6)
chi = G.history()[0]
tm = G.timing() if hasattr(G, "timing") else None
print("LM iterations %d  chi2 %.6e -> %.6e" % (n, chi[0], chi[-1]), tm if tm is None else {k: round(v, 3) for k, v in tm.items() if isinstance(v, float)})
np.save(os.environ.get("MESH_OUT", "/tmp/mesh_x.npy"), x)
