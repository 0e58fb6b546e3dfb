"""Host logic of the bundle adjustment's structure phase (cube_slam_wu_amd/csrc/ba_host.cpp: BlockSolver::buildStructure,
block_solver.hpp:142-295 -- index maps, camera sets, orderings, edge orders, the Schur schedule) WITHOUT a GPU: the library runs under
tools/hostonly/nohip_shim.cpp (device memory is host memory, copies are memcpy, kernels do not run -- nothing is computed), and
cs_ba_structure_digest fingerprints every index table the phase uploads.  Held to each other:
  * the threaded loops (two-level grouping of the edges by landmark, hashed camera sets, per-range counting sorts; from 20 k edges) and the
    same phase on one thread (CS_BA_STRUCT_THREADS=1);
  * a graph grown frame by frame through cs_ba_append_* and the same graph set up at once.
What the tables MEAN is the GPU suite's business (tests/test_ba_gpu.py holds the linear system they produce to the oracle at full C4 size)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_INC = "/opt/rocm/include"

SCRIPT = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from cube_slam_wu_amd import capi, synth_ba
pr = synth_ba.make_problem(n_cams=200, n_points=20000, n_cuboids=50, seed=42)
P = capi.ba_from_dict(pr)
print("ATONCE", P.reduced_size(), P.schur_layout(), P.structure_digest())
P.close()
# the same graph with its points ordered by the first camera that sees them, set up for the first 196 cameras and grown by four frames
nc = len(pr["cams"])
fp = np.full(len(pr["points"]), nc); np.minimum.at(fp, pr["e_pt"], pr["e_cam"])
order = np.argsort(fp, kind="stable"); rank = np.empty_like(order); rank[order] = np.arange(len(order))
fp = fp[order]; ept = rank[pr["e_pt"]]
pts, ptf = pr["points"][order], pr["pt_fixed"][order]
T0 = nc - 4
def edges(sel_p, sel_c, sel_o):
    return ((ept[sel_p], pr["e_cam"][sel_p], pr["e_uv"][sel_p], pr["e_info"][sel_p], pr["e_intr"][sel_p], pr["e_huber"][sel_p]),
            (pr["ce_cam"][sel_c], pr["ce_cub"][sel_c], pr["ce_meas"][sel_c], pr["ce_info"][sel_c]),
            (pr["oe_i"][sel_o], pr["oe_j"][sel_o], pr["oe_meas"][sel_o], pr["oe_info"][sel_o]))
omax = np.maximum(pr["oe_i"], pr["oe_j"])
n_p = int((fp < T0).sum())
G = capi.BaProblem(pr["cams"][:T0], pr["cam_fixed"][:T0], pr["cuboids"], pr["cub_fixed"], pts[:n_p], ptf[:n_p])
# (2 %% of the early cameras' projection edges are held back and arrive with the last frame: appended edges whose camera sorts BEFORE the
# landmark's existing ones -- the grown graph's camera lists are extended, not rebuilt, and must still come out sorted)
late = (np.random.default_rng(1).random(len(ept)) < 0.02) & (pr["e_cam"] < T0)
ep, ec, eo = edges((pr["e_cam"] < T0) & ~late, pr["ce_cam"] < T0, omax < T0)
G.set_edges_proj(*ep); G.set_edges_cuboid(*ec); G.set_edges_odom(*eo)
G.structure_digest()
parts = [(ep, ec, eo)]
for t in range(T0, nc):
    n_p2 = int((fp < t + 1).sum())
    G.append_vertices(pr["cams"][t:t + 1], pr["cam_fixed"][t:t + 1], None, None, pts[n_p:n_p2], ptf[n_p:n_p2])
    ep, ec, eo = edges((pr["e_cam"] == t) | (late if t == nc - 1 else False), pr["ce_cam"] == t, omax == t)
    G.append_edges_proj(*ep); G.append_edges_cuboid(*ec); G.append_edges_odom(*eo)
    parts.append((ep, ec, eo)); n_p = n_p2
    G.structure_digest()                      # (the structure phase after every frame, as optimize() would run it)
print("GROWN", G.reduced_size(), G.schur_layout(), G.structure_digest())
G.close()
cat = lambda k, q: np.concatenate([p[k][q] for p in parts])
H = capi.BaProblem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pts, ptf)
H.set_edges_proj(*[cat(0, q) for q in range(6)]); H.set_edges_cuboid(*[cat(1, q) for q in range(4)]); H.set_edges_odom(*[cat(2, q) for q in range(4)])
print("SAMEORDER", H.reduced_size(), H.schur_layout(), H.structure_digest())
H.close()
"""


def _run(shim, threads):
    env = dict(os.environ, LD_PRELOAD=str(shim), CS_BA_UPLOAD_KERNEL="0")     # (kernels do not run under the shim: the tables go up by per-table copies)
    env.pop("CS_BA_STRUCT_THREADS", None)
    if threads:
        env["CS_BA_STRUCT_THREADS"] = str(threads)
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = {l.split(" ", 1)[0]: l.split(" ", 1)[1] for l in out.stdout.splitlines() if l.split(" ", 1)[0] in ("ATONCE", "GROWN", "SAMEORDER")}
    assert set(lines) == {"ATONCE", "GROWN", "SAMEORDER"}, out.stdout[-2000:]
    return lines


def test_structure_tables_threaded_equal_sequential_and_grown_equals_at_once(tmp_path):
    if shutil.which("g++") is None or not os.path.exists(os.path.join(HIP_INC, "hip", "hip_runtime_api.h")):
        pytest.skip("needs g++ and the HIP headers")
    shim = tmp_path / "nohip_shim.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INC, os.path.join(ROOT, "tools", "hostonly", "nohip_shim.cpp"), "-o", str(shim)])
    threaded, sequential = _run(shim, 0), _run(shim, 1)
    assert threaded == sequential                                   # every table of all three builds, threads or not
    assert threaded["GROWN"] == threaded["SAMEORDER"]               # appended frame by frame = set up at once (same vertex and edge order)
    assert "None" not in threaded["ATONCE"]
