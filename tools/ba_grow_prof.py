"""Phase clock of the structure phase of a GROWING graph (C3-sized, one frame appended at a time -- the bench's growing_graph leg) with
CS_BA_PROF=1's marks on stderr:   python tools/ba_grow_prof.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["CS_BA_PROF"] = "1"
from cube_slam_wu_amd import capi, synth_ba

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pr3 = synth_ba.make_problem(n_cams=200, n_points=20000, n_cuboids=50, seed=42)
nc3 = len(pr3["cams"])
fp = np.full(len(pr3["points"]), nc3); np.minimum.at(fp, pr3["e_pt"], pr3["e_cam"])
order3 = np.argsort(fp, kind="stable"); rank3 = np.empty_like(order3); rank3[order3] = np.arange(len(order3))
fp = fp[order3]
ep3 = rank3[pr3["e_pt"]]
T0 = nc3 - 10
keep_c = pr3["ce_cam"] < T0; sel = pr3["e_cam"] < T0; selo = np.maximum(pr3["oe_i"], pr3["oe_j"]) < T0
n_p = int((fp < T0).sum())
Pg = capi.BaProblem(pr3["cams"][:T0], pr3["cam_fixed"][:T0], pr3["cuboids"], pr3["cub_fixed"], pr3["points"][order3][:n_p], pr3["pt_fixed"][order3][:n_p])
Pg.set_edges_proj(ep3[sel], pr3["e_cam"][sel], pr3["e_uv"][sel], pr3["e_info"][sel], pr3["e_intr"][sel], pr3["e_huber"][sel])
Pg.set_edges_cuboid(pr3["ce_cam"][keep_c], pr3["ce_cub"][keep_c], pr3["ce_meas"][keep_c], pr3["ce_info"][keep_c])
Pg.set_edges_odom(pr3["oe_i"][selo], pr3["oe_j"][selo], pr3["oe_meas"][selo], pr3["oe_info"][selo])
Pg.optimize(5)
for t in range(T0, T0 + frames):
    sel = pr3["e_cam"] == t; keep_c = pr3["ce_cam"] == t; selo = np.maximum(pr3["oe_i"], pr3["oe_j"]) == t
    n_p2 = int((fp < t + 1).sum())
    a_v = (pr3["cams"][t:t + 1], pr3["cam_fixed"][t:t + 1], None, None, pr3["points"][order3][n_p:n_p2], pr3["pt_fixed"][order3][n_p:n_p2])
    a_p = (ep3[sel], pr3["e_cam"][sel], pr3["e_uv"][sel], pr3["e_info"][sel], pr3["e_intr"][sel], pr3["e_huber"][sel])
    a_c = (pr3["ce_cam"][keep_c], pr3["ce_cub"][keep_c], pr3["ce_meas"][keep_c], pr3["ce_info"][keep_c])
    a_o = (pr3["oe_i"][selo], pr3["oe_j"][selo], pr3["oe_meas"][selo], pr3["oe_info"][selo])
    t1 = time.perf_counter()
    Pg.append_vertices(*a_v); Pg.append_edges_proj(*a_p); Pg.append_edges_cuboid(*a_c); Pg.append_edges_odom(*a_o)
    t2 = time.perf_counter()
    Pg.sizes()
    t3 = time.perf_counter()
    Pg.optimize(5)
    t4 = time.perf_counter()
    print("frame %d: %d edges appended; append calls %.2f ms, structure phase %.2f ms, optimize(5) %.2f ms" % (t, int(sel.sum()), (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), file=sys.stderr)
    n_p = n_p2
Pg.close()
