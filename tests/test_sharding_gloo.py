"""N > 1 paths on CPU: two / three processes, torch.distributed with the gloo backend (no GPU involved).

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


def _sep_worker(rank, world, port, out):
    """Separator mode of the sharded solve (include/cubeslam_hip.h, ba_host.cpp solve_device_sep) restated in numpy, one process
    per rank: ownership by lowest column, the rank's partial system in its own columns + the next separator's diagonal block,
    interior Cholesky, the message [LL | RL | RR | tL | tR], one all-gather, the separator system solved by every rank, interior
    back-substitution, one all-reduce of the solution vector."""
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cube_slam_wu_amd import synth_ba
        from oracle import ba_oracle_py as O

        pr = synth_ba.make_problem(n_cams=90, n_points=2500, n_cuboids=0, seed=4)
        nc = len(pr["cams"])
        cam_col = np.full(nc, -1); cam_col[1:] = 6 * np.arange(nc - 1)     # camera 0 is fixed; trajectory order = a banded system
        n = 6 * (nc - 1)
        lam = 3.0

        def build(ek, ok_):
            P = O.Problem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"])
            P.set_edges_proj(pr["e_pt"][ek], pr["e_cam"][ek], pr["e_uv"][ek], pr["e_info"][ek], pr["e_intr"][ek], pr["e_huber"][ek])
            P.set_edges_odom(pr["oe_i"][ok_], pr["oe_j"][ok_], pr["oe_meas"][ok_], pr["oe_info"][ok_])
            return P

        def reduced(P, e_pt_l, e_cam_l, lam_cols):
            Hpp, Hll, Hpl, b = P.build_system()
            S = Hpp.copy()
            S[lam_cols, lam_cols] += lam
            rhs = b[:n].copy()
            for p in np.unique(e_pt_l):
                Dinv = np.linalg.inv(Hll[p].reshape(3, 3) + lam * np.eye(3))
                ks = np.nonzero(e_pt_l == p)[0]
                for ka in ks:
                    ca = cam_col[e_cam_l[ka]]
                    if ca < 0:
                        continue
                    Wa = Hpl[ka].reshape(6, 3)
                    rhs[ca:ca + 6] -= Wa @ Dinv @ b[n + 3 * p:n + 3 * p + 3]
                    for kb in ks:
                        cb = cam_col[e_cam_l[kb]]
                        if cb >= 0:
                            S[ca:ca + 6, cb:cb + 6] -= Wa @ Dinv @ Hpl[kb].reshape(6, 3).T
            return S, rhs

        full = np.ones(len(pr["e_pt"]), bool); fo = np.ones(len(pr["oe_i"]), bool)
        Sf, rf = reduced(build(full, fo), pr["e_pt"], pr["e_cam"], np.arange(n))
        x_full = np.linalg.solve(Sf, rf)
        nz = np.nonzero(np.abs(Sf) > 0)
        bw = int((nz[0] - nz[1]).max())
        # the cut: rank r owns [cut[r], cut[r + 1]); its first sepw[r] >= bw columns are the separator Z_r (none on rank 0)
        cut = [0] + [6 * int(np.ceil(n * r / world / 6)) for r in range(1, world)] + [n]
        w = 6 * int(np.ceil(bw / 6))
        sepw = [0] + [w] * (world - 1)
        assert all(cut[r + 1] - cut[r] - sepw[r] > bw for r in range(world))
        rank_of_col = lambda c: int(np.searchsorted(cut, c, side="right") - 1)
        # owners: lowest column among the free cameras
        lo = np.full(len(pr["points"]), 10 ** 9)
        cc = cam_col[pr["e_cam"]]
        np.minimum.at(lo, pr["e_pt"][cc >= 0], cc[cc >= 0])
        lm_owner = np.array([rank_of_col(c) if c < 10 ** 9 else 0 for c in lo])
        ek = lm_owner[pr["e_pt"]] == rank
        oc = np.stack([cam_col[pr["oe_i"]], cam_col[pr["oe_j"]]], 1).astype(float); oc[oc < 0] = np.inf
        ok_ = np.array([rank_of_col(int(c)) if np.isfinite(c) else 0 for c in oc.min(1)]) == rank
        S, rhs = reduced(build(ek, ok_), pr["e_pt"][ek], pr["e_cam"][ek], np.arange(cut[rank], cut[rank + 1]))
        # everything this rank built lies in its own columns and in the next separator's diagonal block
        zl, wl = cut[rank], sepw[rank]
        ci, ni = zl + wl, cut[rank + 1] - zl - wl
        zr, wr = cut[rank + 1], (sepw[rank + 1] if rank + 1 < world else 0)
        mask = np.zeros((n, n), bool)
        mask[zl:zr, zl:zr] = True; mask[zr:zr + wr, zl:zr + wr] = True; mask[zl:zr + wr, zr:zr + wr] = True
        assert np.abs(S[~mask]).max(initial=0.0) == 0 and np.abs(S[zr:zr + wr, zl:ci]).max(initial=0.0) == 0
        # interior factorisation, Y = B L^-T, y = L^-1 b_I
        I = slice(ci, ci + ni)
        L = np.linalg.cholesky(S[I, I])
        Z = np.r_[zl:zl + wl, zr:zr + wr]
        Bm = S[Z][:, I]
        Y = np.linalg.solve(L, Bm.T).T
        y = np.linalg.solve(L, rhs[I])
        T = S[np.ix_(Z, Z)] - Y @ Y.T
        t = rhs[Z] - Y @ y
        msg = np.zeros(3 * w * w + 2 * w)
        LL, RL, RR = msg[:w * w].reshape(w, w), msg[w * w:2 * w * w].reshape(w, w), msg[2 * w * w:3 * w * w].reshape(w, w)
        LL[:wl, :wl] = T[:wl, :wl]; RL[:wr, :wl] = T[wl:, :wl]; RR[:wr, :wr] = T[wl:, wl:]
        msg[3 * w * w:3 * w * w + wl] = t[:wl]; msg[3 * w * w + w:3 * w * w + w + wr] = t[wl:]
        assert msg.nbytes < 0.02 * S[np.tril_indices(n)].nbytes + 8 * n or world == 2      # the message against the band
        msgs = [torch.zeros(len(msg), dtype=torch.float64) for _ in range(world)]
        dist.all_gather(msgs, torch.from_numpy(msg))            # collective 1
        msgs = [m.numpy() for m in msgs]
        # the separator system (every rank): separator k = left of rank k = right of rank k - 1
        ns = w * (world - 1)
        Ss, rs = np.zeros((ns, ns)), np.zeros(ns)
        blk = lambda m, i: m[i * w * w:(i + 1) * w * w].reshape(w, w)
        for k in range(1, world):
            o = (k - 1) * w
            Ss[o:o + w, o:o + w] = blk(msgs[k], 0) + blk(msgs[k - 1], 2)
            rs[o:o + w] = msgs[k][3 * w * w:3 * w * w + w] + msgs[k - 1][3 * w * w + w:3 * w * w + 2 * w]
            if k + 1 < world:
                Ss[o + w:o + 2 * w, o:o + w] = blk(msgs[k], 1)
                Ss[o:o + w, o + w:o + 2 * w] = blk(msgs[k], 1).T
        xs = np.linalg.solve(Ss, rs)
        x = np.zeros(n)
        for k in range(1, world):
            x[cut[k]:cut[k] + w] = xs[(k - 1) * w:k * w]
        x[I] = np.linalg.solve(L.T, y - Y.T @ x[Z])
        xi = np.zeros(n); xi[I] = x[I]
        buf = torch.from_numpy(xi)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)             # collective 2
        x_all = buf.numpy()
        for k in range(1, world):
            x_all[cut[k]:cut[k] + w] = xs[(k - 1) * w:k * w]
        assert np.abs(x_all - x_full).max() <= 1e-9 * np.abs(x_full).max(), np.abs(x_all - x_full).max()
        out.put((rank, "ok"))
    except Exception:
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_separator_mode_world_size_3_gloo():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sep_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


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
