#!/usr/bin/env python3
"""tests/golden/ba_proj_schur_30.npz -- a pin for the projection-edge + Schur half of path B that does not come from oracle/.

A from-scratch numpy restatement of what the reference runs for EdgeSE3ProjectXYZ graphs, written against the reference's sources
only (nothing under oracle/ or cube_slam_wu_amd/csrc is imported or consulted by this script):

  error / Jacobians     object_slam/Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:155-160, types_six_dof_expmap.cpp:148-192
  vertex update         types_six_dof_expmap.h:73-76 (SE3Quat::exp(update) * estimate), se3quat.h:272-324 (exp), :58-60,:346-351
  quadratic form        core/base_binary_edge.hpp:54-120 with the Huber kernel core/robust_kernel_impl.cpp:78-91 (rho' only:
                        core/base_edge.h:96-102)
  Schur complement      core/block_solver.hpp:367-486 (lambda on pose AND landmark diagonals, :573-587)
  Levenberg-Marquardt   core/optimization_algorithm_levenberg.cpp:61-189 (tau = 1e-5, step factors 1/3 .. 2/3, 10 trials, the
                        vendored early-stop rule :153-161)

Rotations are carried as matrices here (the reference carries quaternions; same group elements), everything is dense numpy, the
reduced system is solved with numpy.linalg.solve.  The problem is a 30-camera / 900-point KITTI-shaped chain from
cube_slam_wu_amd.synth_ba (workload generator only), projection edges with Huber kernels, camera 0 fixed.  The script stores the
inputs, the linear system at the initial state (per-vertex diagonal blocks, per-edge H_pl blocks, b), the chi2 / lambda / trial
history of 5 LM iterations and the final states.  tests/test_ba_oracle.py holds oracle/ba_oracle.cpp to this file on the CPU,
tests/test_ba_gpu.py holds the HIP path to it on the GPU.

Run in the build container:  python tools/make_ba_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# ---------------------------------------------------------------- SE(3) as (R, t), quaternions only at the file boundary
def quat_to_R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(R).as_quat()        # x y z w, unit
    return -q if q[3] < 0 else q


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def se3_exp(u):
    """se3quat.h:272-324: u = (omega, upsilon)."""
    omega, ups = u[:3], u[3:]
    th = np.linalg.norm(omega)
    Om = skew(omega)
    if th < 1e-5:
        R = np.eye(3) + Om + Om @ Om
        V = R
    else:
        Om2 = Om @ Om
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / (th * th) * Om2
        V = np.eye(3) + (1 - np.cos(th)) / (th * th) * Om + (th - np.sin(th)) / th ** 3 * Om2
    # SE3Quat(Quaterniond(R), t) normalises the quaternion: the rotation actually stored is the nearest one to R along that path
    return quat_to_R(R_to_quat_raw(R)), V @ ups


def R_to_quat_raw(R):
    """Eigen's Quaterniond(Matrix3d) on a matrix that is only approximately orthonormal (the small-angle branch), then normalised."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(4)
        q[i] = 0.5 * s
        s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


class Graph:
    def __init__(self, cams7, cam_fixed, points, e_pt, e_cam, uv, info4, intr4, huber):
        self.R = np.array([quat_to_R(c[3:7]) for c in cams7])
        self.t = cams7[:, :3].copy()
        self.cam_fixed = cam_fixed.astype(bool)
        self.X = points.copy()
        self.e_pt, self.e_cam, self.uv = e_pt, e_cam, uv
        self.info = info4.reshape(-1, 2, 2)
        self.intr, self.huber = intr4, huber
        self.nc, self.npt, self.ne = len(cams7), len(points), len(e_pt)
        self.cam_col = np.full(self.nc, -1)
        self.cam_col[~self.cam_fixed] = 6 * np.arange((~self.cam_fixed).sum())
        self.n_pose = 6 * int((~self.cam_fixed).sum())

    # ---- errors: obs - project(T * X)
    def errors(self):
        Xc = np.einsum("eij,ej->ei", self.R[self.e_cam], self.X[self.e_pt]) + self.t[self.e_cam]
        fx, fy, cx, cy = self.intr.T
        proj = np.stack([Xc[:, 0] / Xc[:, 2] * fx + cx, Xc[:, 1] / Xc[:, 2] * fy + cy], 1)
        return self.uv - proj, Xc

    def robust(self, chi):
        """Huber: rho(e), rho'(e) of the squared error e (robust_kernel_impl.cpp:78-91); delta = 0 means no kernel."""
        rho0, rho1 = chi.copy(), np.ones_like(chi)
        d = self.huber
        # RobustKernelHuber keeps delta^2 in a `float` member (robust_kernel_impl.h:86: `float dsqr`, assigned delta * delta by setDelta,
        # robust_kernel_impl.cpp:65-69): the inlier test and rho(e) see the single-precision square
        dsqr = (d * d).astype(np.float32).astype(np.float64)
        out = (d > 0) & (chi > dsqr)
        s = np.sqrt(chi[out])
        rho0[out] = 2 * s * d[out] - dsqr[out]
        rho1[out] = d[out] / s
        return rho0, rho1

    def chi2(self):
        e, _ = self.errors()
        chi = np.einsum("ei,eij,ej->e", e, self.info, e)
        return float(self.robust(chi)[0].sum())

    # ---- linearisation + quadratic form
    def build(self):
        e, Xc = self.errors()
        chi = np.einsum("ei,eij,ej->e", e, self.info, e)
        _, rho1 = self.robust(chi)
        x, y, z = Xc.T
        z2 = z * z
        fx, fy = self.intr[:, 0], self.intr[:, 1]
        tmp = np.zeros((self.ne, 2, 3))
        tmp[:, 0, 0] = fx; tmp[:, 0, 2] = -x / z * fx
        tmp[:, 1, 1] = fy; tmp[:, 1, 2] = -y / z * fy
        Ji = -1.0 / z[:, None, None] * np.einsum("eij,ejk->eik", tmp, self.R[self.e_cam])       # d error / d point
        Jj = np.zeros((self.ne, 2, 6))                                                        # d error / d pose (omega, upsilon)
        Jj[:, 0, 0] = x * y / z2 * fx; Jj[:, 0, 1] = -(1 + x * x / z2) * fx; Jj[:, 0, 2] = y / z * fx
        Jj[:, 0, 3] = -1.0 / z * fx; Jj[:, 0, 5] = x / z2 * fx
        Jj[:, 1, 0] = (1 + y * y / z2) * fy; Jj[:, 1, 1] = -x * y / z2 * fy; Jj[:, 1, 2] = -x / z * fy
        Jj[:, 1, 4] = -1.0 / z * fy; Jj[:, 1, 5] = y / z2 * fy
        W = rho1[:, None, None] * self.info
        omega_r = -rho1[:, None] * np.einsum("eij,ej->ei", self.info, e)
        Hcam = np.zeros((self.nc, 6, 6)); bcam = np.zeros((self.nc, 6))
        Hpt = np.zeros((self.npt, 3, 3)); bpt = np.zeros((self.npt, 3))
        Hpl = np.einsum("eki,ekl,elj->eij", Jj, W, Ji)                                        # pose rows, point columns (6 x 3)
        np.add.at(Hcam, self.e_cam, np.einsum("eki,ekl,elj->eij", Jj, W, Jj))
        np.add.at(bcam, self.e_cam, np.einsum("eki,ek->ei", Jj, omega_r))
        np.add.at(Hpt, self.e_pt, np.einsum("eki,ekl,elj->eij", Ji, W, Ji))
        np.add.at(bpt, self.e_pt, np.einsum("eki,ek->ei", Ji, omega_r))
        Hcam[self.cam_fixed] = 0; bcam[self.cam_fixed] = 0
        Hpl[self.cam_fixed[self.e_cam]] = 0
        self.sys = (Hcam, bcam, Hpt, bpt, Hpl)
        return self.sys

    # ---- Schur complement solve with lambda on every diagonal
    def solve(self, lam):
        Hcam, bcam, Hpt, bpt, Hpl = self.sys
        n = self.n_pose
        S = np.zeros((n, n)); r = np.zeros(n)
        for c in range(self.nc):
            k = self.cam_col[c]
            if k >= 0:
                S[k:k + 6, k:k + 6] = Hcam[c] + lam * np.eye(6)
                r[k:k + 6] = bcam[c]
        Dinv = np.linalg.inv(Hpt + lam * np.eye(3))
        order = np.argsort(self.e_pt, kind="stable")
        bounds = np.searchsorted(self.e_pt[order], np.arange(self.npt + 1))
        for p in range(self.npt):
            es = order[bounds[p]:bounds[p + 1]]
            es = es[self.cam_col[self.e_cam[es]] >= 0]
            for a in es:
                ka = self.cam_col[self.e_cam[a]]
                WD = Hpl[a] @ Dinv[p]
                r[ka:ka + 6] -= WD @ bpt[p]
                for b in es:
                    kb = self.cam_col[self.e_cam[b]]
                    S[ka:ka + 6, kb:kb + 6] -= WD @ Hpl[b].T
        try:
            np.linalg.cholesky(S)
        except np.linalg.LinAlgError:
            return False, None, None
        xp = np.linalg.solve(S, r)
        cl = bpt.copy()
        for e in range(self.ne):
            k = self.cam_col[self.e_cam[e]]
            if k >= 0:
                cl[self.e_pt[e]] -= Hpl[e].T @ xp[k:k + 6]
        xl = np.einsum("pij,pj->pi", Dinv, cl)
        return True, xp, xl

    def update(self, xp, xl):
        for c in range(self.nc):
            k = self.cam_col[c]
            if k >= 0:
                dR, dt = se3_exp(xp[k:k + 6])
                self.t[c] = dt + dR @ self.t[c]
                self.R[c] = quat_to_R(R_to_quat_raw(dR @ self.R[c]))      # the product is re-normalised (se3quat.h:346-351)
        self.X += xl

    def optimize(self, iters):
        chi_h, lam_h, tr_h = [], [], []
        lam, ni, n_bad = 0.0, 2.0, 0
        for it in range(iters):
            cur = self.chi2()
            ini = cur
            Hcam, bcam, Hpt, bpt, _ = self.build()
            if it == 0:
                md = 0.0
                for c in range(self.nc):
                    if self.cam_col[c] >= 0:
                        md = max(md, np.abs(np.diag(Hcam[c])).max())
                md = max(md, np.abs(np.einsum("pii->pi", Hpt)).max())
                lam, ni, n_bad = 1e-5 * md, 2.0, 0
            rho, q = 0.0, 0
            while True:
                saved = (self.R.copy(), self.t.copy(), self.X.copy())
                ok, xp, xl = self.solve(lam)
                scale = 0.0
                if ok:
                    self.update(xp, xl)
                    b_all = np.concatenate([bcam[~self.cam_fixed].ravel(), bpt.ravel()])
                    x_all = np.concatenate([xp, xl.ravel()])
                    scale = float(np.sum(x_all * (lam * x_all + b_all)))
                tmp = self.chi2() if ok else np.finfo(float).max
                rho = (cur - tmp) / (scale + 1e-3)
                if rho > 0 and np.isfinite(tmp):
                    alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                    lam *= max(1.0 / 3.0, alpha)
                    ni = 2.0
                    cur = tmp
                else:
                    lam *= ni
                    ni *= 2
                    self.R, self.t, self.X = saved
                q += 1
                if not (rho < 0 and q < 10):
                    break
            chi_h.append(cur); lam_h.append(lam); tr_h.append(q)
            if q == 10 or rho == 0:
                break
            n_bad = n_bad + 1 if (ini - cur) * 1e3 < ini else 0
            if n_bad >= 3:
                break
        return np.array(chi_h), np.array(lam_h), np.array(tr_h, np.int32)

    def cams7(self):
        return np.concatenate([self.t, np.array([R_to_quat(R) for R in self.R])], 1)


def main():
    from cube_slam_wu_amd import synth_ba
    pr = synth_ba.make_problem(n_cams=30, n_points=900, n_cuboids=0, seed=2024)
    G = Graph(pr["cams"], pr["cam_fixed"], pr["points"], pr["e_pt"], pr["e_cam"], pr["e_uv"], pr["e_info"], pr["e_intr"], pr["e_huber"])
    chi0 = G.chi2()
    Hcam, bcam, Hpt, bpt, Hpl = G.build()
    ok, xp, xl = G.solve(7.5)
    assert ok
    chi_h, lam_h, tr_h = G.optimize(5)
    out = os.path.join(ROOT, "tests", "golden", "ba_proj_schur_30.npz")
    np.savez_compressed(
        out, cams=pr["cams"], cam_fixed=pr["cam_fixed"], points=pr["points"], pt_fixed=pr["pt_fixed"], e_pt=pr["e_pt"], e_cam=pr["e_cam"], e_uv=pr["e_uv"],
        e_info=pr["e_info"], e_intr=pr["e_intr"], e_huber=pr["e_huber"],
        chi2_initial=chi0, Hcam=Hcam.reshape(-1, 36), bcam=bcam, Hpt=Hpt.reshape(-1, 9), bpt=bpt, Hpl=Hpl.reshape(-1, 18),
        solve_lambda=7.5, solve_xp=xp, solve_xl=xl,
        chi2_hist=chi_h, lambda_hist=lam_h, trials_hist=tr_h, final_cams=G.cams7(), final_points=G.X)
    print("wrote %s: %d cameras, %d points, %d edges; chi2 %.6g -> %s, trials %s" % (out, G.nc, G.npt, G.ne, chi0, chi_h, tr_h))


if __name__ == "__main__":
    main()
