"""Differential run of the sharded BA (R handles on one GPU, one thread each, in-process all-reduce callback) against the unsharded
optimisation on random trajectory lengths and shard counts -- both exchange modes occur (all-reduce of the band for short shares, separator mode
for long ones).  python tools/fuzz_sharded_ba.py [problems]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cube_slam_wu_amd import capi, synth_ba
import test_ba_gpu as T

n_prob = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(5)
bad = 0
modes = {0: 0, 1: 0}
t0 = time.time()
for k in range(n_prob):
    R = int(rng.integers(2, 6)); nc = int(rng.integers(30, 420)); no = int(rng.integers(0, max(1, nc // 6)))
    pr = synth_ba.make_problem(n_cams=nc, n_points=int(rng.integers(30, 80)) * nc, n_cuboids=no, seed=int(rng.integers(1, 10**6)), huber=bool(rng.integers(0, 2)))
    G = capi.ba_from_dict(pr)
    n1 = G.optimize(5)
    chi1, lam1, tr1 = G.history(); c1, o1, p1 = G.state(); G.close()
    try:
        done, (chiS, lamS, trS), cS, oS, pS = T._run_sharded_in_threads(pr, R, 5)
    except AssertionError as e:
        bad += 1; print("assertion inside the sharded run", k, dict(R=R, nc=nc, no=no), e); continue
    modes[int(T._run_sharded_in_threads.info[0]["sep_mode"])] += 1
    scale = max(1.0, float(np.abs(p1).max()))
    ok = done == [n1] * R and np.array_equal(tr1, trS) and np.allclose(chi1, chiS, rtol=1e-9) and np.allclose(lam1, lamS, rtol=1e-9)
    ok = ok and np.abs(pS - p1).max() < 1e-7 * scale and np.abs(cS - c1).max() < 1e-7 * scale and (not len(o1) or np.abs(oS - o1).max() < 1e-7 * scale)
    if not ok:
        bad += 1; print("mismatch", k, dict(R=R, nc=nc, no=no), done, n1, list(tr1), list(trS))
print("%d problems (%d through the band all-reduce, %d in separator mode), %d mismatches, %.0f s" % (n_prob, modes[0], modes[1], bad, time.time() - t0))
sys.exit(1 if bad else 0)
