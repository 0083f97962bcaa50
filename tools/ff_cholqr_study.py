"""Would the full-frame Jacobi gain from running on the L x L Cholesky factor of A A^T instead of the L x M rows
?  Numerical experiment in float32 on BASELINE config 2's shape (1080 x 1920, uniform noise - the
bench content - and camera-like content): how orthogonal are the rows of B = R A when the rotations R come from the
factor, i.e. how many clean-up sweeps on the full-length rows would remain?
    python tools/ff_cholqr_study.py"""
import numpy as np, scipy.linalg as sl, time


def study(name, A):
    L, M = A.shape
    A32 = A.astype(np.float32)
    G = (A32 @ A32.T).astype(np.float32)                       # the Gram product as the MFMA SGEMM would form it
    s = np.linalg.svd(A.astype(np.float64), compute_uv=False)
    print(f"{name}: sigma_1 / sigma_L = {s[0] / s[-1]:.1f}, eps * kappa^2 = {6e-8 * (s[0] / s[-1]) ** 2:.2e}")
    try:
        Lc = np.linalg.cholesky(G.astype(np.float64)).astype(np.float32)   # float64 factorisation of the float32 Gram: the best case
    except np.linalg.LinAlgError:
        print("   Cholesky of the float32 Gram matrix breaks down"); return
    # "converged Jacobi on the factor": exact left singular vectors of Lc (float64), applied to A in float32
    U, sl_, _ = np.linalg.svd(Lc.astype(np.float64))
    R = U.T.astype(np.float32)
    B = (R @ A32).astype(np.float64)
    n = np.linalg.norm(B, axis=1)
    C = (B @ B.T) / np.outer(n, n)
    np.fill_diagonal(C, 0)
    c = np.abs(C)
    print(f"   rows of B = R A: max |cos| {c.max():.2e}, pairs above the stopping cosine 2e-4: {(c > 2e-4).sum() // 2} of {L * (L - 1) // 2}, "
          f"above 1e-2: {(c > 1e-2).sum() // 2}, rows involved: {(c.max(1) > 2e-4).sum()} of {L}")
    print(f"   sigma from the factor vs A: max rel err {np.max(np.abs(sl_ - s) / s):.2e} (at sigma/sigma_1 = {s[np.argmax(np.abs(sl_ - s) / s)] / s[0]:.1e})")
    # quadratic convergence from there: sweeps until max cos < 2e-4, by the c -> c^2 rule with the row-cyclic constant ~1
    cc, sweeps = c.max(), 0
    while cc > 2e-4 and sweeps < 10:
        cc = cc * cc * (s[0] / s[-1]) if cc * (s[0] / s[-1]) < 1 else cc * 0.5     # graded rows: the worst pair improves slower
        sweeps += 1
    print(f"   clean-up sweeps on the full-length rows by the quadratic rule: >= {max(sweeps, 1)}")


rng = np.random.default_rng(1234)
H, W = 1080, 1920
study("uniform noise (bench content)", rng.integers(0, 256, (H, W)).astype(np.float64))
yy, xx = np.mgrid[0:H, 0:W]
cam = np.clip(128 + 70 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 40 * np.sin((xx + 2 * yy) / 91.0) + rng.normal(0, 2.0, (H, W)), 0, 255).round()
study("smooth field + sensor noise (camera-like)", cam)
