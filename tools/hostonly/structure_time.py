"""Host-side clock of the BA structure phase WITHOUT a GPU (run under tools/hostonly/nohip_shim.cpp: no kernel runs, nothing is
computed; the index maps / orderings / schedules are host work and are what is timed here):
   LD_PRELOAD=build_tmp/nohip_shim.so CS_BA_PROF=1 python tools/hostonly/structure_time.py [C4|C3] [repeats]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cube_slam_wu_amd import capi, synth_ba

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nc, npt, no = (1000, 200000, 500) if cfg == "C4" else (200, 20000, 50)
pr = synth_ba.make_problem(n_cams=nc, n_points=npt, n_cuboids=no, seed=42)
for r in range(rep):
    P = capi.ba_from_dict(pr)
    t0 = time.perf_counter()
    P.sizes()
    print("%s structure %.1f ms; reduced %s path %s schur %s" % (cfg, (time.perf_counter() - t0) * 1e3, P.reduced_size(), P.solver_path(detail=True), P.schur_layout()), flush=True)
    P.close()
