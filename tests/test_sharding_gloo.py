"""N > 1 paths on CPU: two processes, torch.distributed with the gloo backend (no GPU involved).

Path A shards frames (independent units, no data-path collective): each rank runs its frames, rank 0 gathers.
Path B shards landmarks by camera subsequence and sums the ranks' partial reduced systems [S | b_schur] with one
all-reduce per damped solve.  The GPU library cannot run here, so each rank builds its partial system with the CPU
oracle restricted to the edges the library's own ownership rule (cs_ba_shard_landmark_owners, host-only) gives it;
the summed system must equal the unsharded one.  The same code path on real GPUs is covered by
tests/test_ba_gpu.py::test_sharded_ba_equals_single_rank.
"""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, out):
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from cube_slam_wu_amd import capi, synth, synth_ba
        from oracle import ba_oracle_py as O
        from oracle import oracle_py

        # ---------------- path A: frames round-robin, gather on rank 0
        seeds = [7000 + i for i in range(5)]
        mine = capi.shard_frames(len(seeds), rank, WORLD)
        local = {}
        for i in mine:
            fr = synth.make_frame(seeds[i], n_boxes=2, n_lines=120)
            res, _ = oracle_py.detect_cuboid(fr, oracle_py.default_params(), atan2_mode=1)
            local[i] = [[c["box_corners_2d"].tolist() for c in r] for r in res]
        gathered = [None] * WORLD
        dist.all_gather_object(gathered, local)
        if rank == 0:
            merged = {}
            for g in gathered:
                merged.update(g)
            assert sorted(merged) == list(range(len(seeds)))
            for i, s in enumerate(seeds):
                fr = synth.make_frame(s, n_boxes=2, n_lines=120)
                res, _ = oracle_py.detect_cuboid(fr, oracle_py.default_params(), atan2_mode=1)
                assert merged[i] == [[c["box_corners_2d"].tolist() for c in r] for r in res]

        # ---------------- path B: landmark-sharded reduced system, one all-reduce
        pr = synth_ba.make_problem(n_cams=16, n_points=400, n_cuboids=3, seed=9)
        nc, npnt = len(pr["cams"]), len(pr["points"])
        owners = capi.landmark_owners(WORLD, nc, npnt, pr["e_pt"], pr["e_cam"])
        first_cam = np.full(npnt, 10 ** 9)
        np.minimum.at(first_cam, pr["e_pt"], pr["e_cam"])
        assert np.array_equal(owners, capi.cam_rank(first_cam, nc, WORLD))           # the documented rule
        keep = owners[pr["e_pt"]] == rank
        ce_keep = capi.cam_rank(pr["ce_cam"], nc, WORLD) == rank
        oe_keep = capi.cam_rank(pr["oe_j"], nc, WORLD) == rank

        def build(ek, ck, ok_):
            P = O.Problem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"])
            P.set_edges_proj(pr["e_pt"][ek], pr["e_cam"][ek], pr["e_uv"][ek], pr["e_info"][ek], pr["e_intr"][ek], pr["e_huber"][ek])
            P.set_edges_cuboid(pr["ce_cam"][ck], pr["ce_cub"][ck], pr["ce_meas"][ck], pr["ce_info"][ck])
            P.set_edges_odom(pr["oe_i"][ok_], pr["oe_j"][ok_], pr["oe_meas"][ok_], pr["oe_info"][ok_])
            return P

        lam = 7.5
        Pl = build(keep, ce_keep, oe_keep)
        chi_local = Pl.compute_errors()[0]
        Hpp, Hll, Hpl, b = Pl.build_system()
        n = Hpp.shape[0]
        S = Hpp.copy()
        if rank == 0:
            S += lam * np.eye(n)                      # lambda on the pose diagonal is added by exactly one rank
        rhs = b[:n].copy()
        cam_col = np.full(nc, -1); cam_col[1:] = 6 * np.arange(nc - 1)   # camera 0 is fixed
        e_pt_l, e_cam_l = pr["e_pt"][keep], pr["e_cam"][keep]
        for p in np.unique(e_pt_l):
            D = Hll[p].reshape(3, 3) + lam * np.eye(3)
            Dinv = np.linalg.inv(D)
            ks = np.nonzero(e_pt_l == p)[0]
            for ka in ks:
                if cam_col[e_cam_l[ka]] < 0:
                    continue
                Wa = Hpl[ka].reshape(6, 3)
                ca = cam_col[e_cam_l[ka]]
                rhs[ca:ca + 6] -= Wa @ Dinv @ b[n + 3 * p:n + 3 * p + 3]
                for kb in ks:
                    if cam_col[e_cam_l[kb]] < 0:
                        continue
                    cb = cam_col[e_cam_l[kb]]
                    S[ca:ca + 6, cb:cb + 6] -= Wa @ Dinv @ Hpl[kb].reshape(6, 3).T
        buf = torch.from_numpy(np.concatenate([S.ravel(), rhs, [chi_local]]))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)    # the ONE collective of a damped solve (+ the chi2 scalar)
        S_sum = buf[: n * n].numpy().reshape(n, n); rhs_sum = buf[n * n:n * n + n].numpy(); chi_sum = float(buf[-1])
        x_sharded = np.linalg.solve(S_sum, rhs_sum)

        full = np.ones(len(pr["e_pt"]), bool)
        Pf = build(full, np.ones(len(pr["ce_cam"]), bool), np.ones(len(pr["oe_i"]), bool))
        chi_full = Pf.compute_errors()[0]
        Pf.build_system()
        ok, x_full = Pf.solve(lam)
        assert ok
        assert abs(chi_sum - chi_full) <= 1e-9 * chi_full
        assert np.abs(x_sharded - x_full[:n]).max() <= 1e-8 * np.abs(x_full[:n]).max()
        out.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(WORLD)]
    [p.join(timeout=60) for p in procs]
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
