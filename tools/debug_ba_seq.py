import os, sys, numpy as np
sys.path.insert(0, '.')
from cube_slam_wu_amd import capi
from oracle import ba_oracle_py as O
DATA = 'tests/golden/object_slam_data'
def mk(cams, cam_fixed, cuboid, cub_edges, odom_edges):
    P = capi.BaProblem(cams, cam_fixed, cuboids=cuboid[None, :], cub_fixed=[0], cuboids_first=True)
    if cub_edges:
        P.set_edges_cuboid([e[0] for e in cub_edges], [0] * len(cub_edges), np.array([e[1] for e in cub_edges]), np.array([e[2] for e in cub_edges]))
    if odom_edges:
        P.set_edges_odom([e[0] for e in odom_edges], [e[1] for e in odom_edges], np.array([e[2] for e in odom_edges]), np.tile(np.eye(6).ravel(), (len(odom_edges), 1)))
    return P
cam_g, obj_g, it_g, fin_g = O.run_offline_sequence(DATA, make_problem=mk)
cam_r, obj_r, it_r, fin_r = O.run_offline_sequence(DATA)
print("iters equal frames:", (it_g == it_r).sum(), "of", len(it_r))
print("per-frame obj diff max:", np.abs(obj_g - obj_r).max(axis=1)[18:28])
print("obj diff", np.abs(obj_g - obj_r).max(), "cam diff", np.abs(cam_g - cam_r).max(), "final cams diff", np.abs(fin_g - fin_r).max())
