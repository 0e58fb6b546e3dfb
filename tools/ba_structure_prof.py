"""Phase clock of the BA structure phase at C4 (CS_BA_PROF=1 prints it to stderr): python tools/ba_structure_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CS_BA_PROF"] = "1"
from cube_slam_wu_amd import capi, synth_ba
pr = synth_ba.make_problem(n_cams=1000, n_points=200000, n_cuboids=500, seed=42)
for rep in range(3):
    t0 = time.perf_counter()
    P = capi.ba_from_dict(pr)
    t1 = time.perf_counter()
    P.sizes()
    t2 = time.perf_counter()
    print("rep %d: set_* (copies of the caller's arrays) %.1f ms, structure phase %.1f ms" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3), file=sys.stderr)
    P.close()
