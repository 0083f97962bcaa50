"""Ordering study: flat block tournament vs 2-level (super-block) tournament, cyclic one-sided Jacobi on rows.
Sub-block r rows; counts sweeps until the largest cosine seen in a sweep (before rotating) < tol."""
import numpy as np, sys, time

def rot_pair(X, p, q, stat):
    a = X[p] @ X[p]; b = X[q] @ X[q]; g = X[p] @ X[q]
    if a <= 0 or b <= 0: return
    c = abs(g) / np.sqrt(a * b)
    stat[0] = max(stat[0], c)
    if c < stat[1]: return
    tau = b - a
    t2 = 2 * g
    h = np.hypot(tau, t2)
    x = 0.5 + 0.5 * abs(tau) / h
    cs = np.sqrt(x); sn = (g / h) / cs
    # de Rijk: larger to p
    if tau > 0:
        C, S = sn, -cs
    else:
        C, S = cs, -sn
    xp = C * X[p] - S * X[q]; xq = S * X[p] + C * X[q]
    X[p] = xp; X[q] = xq

def tournament(n):
    idx = list(range(n if n % 2 == 0 else n + 1))
    m = len(idx)
    steps = []
    for s in range(m - 1):
        pairs = []
        for k in range(m // 2):
            a, b = idx[k], idx[m - 1 - k]
            if a < n and b < n: pairs.append((min(a, b), max(a, b)))
        steps.append(pairs)
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return steps

def cross(X, ra, rb, stat):
    r = len(ra)
    for step in range(max(len(ra), len(rb))):
        for k in range(len(ra)):
            j = (k + step) % len(rb)
            rot_pair(X, ra[k], rb[j], stat)

def full(X, rows, stat):
    n = len(rows)
    for st in tournament(n):
        for a, b in st: rot_pair(X, rows[a], rows[b], stat)

def sweep_flat(X, r, stat):
    nb = X.shape[0] // r
    rows = [list(range(b * r, (b + 1) * r)) for b in range(nb)]
    for s, st in enumerate(tournament(nb)):
        for a, b in st:
            if s == 0: full(X, rows[a] + rows[b], stat)
            else: cross(X, rows[a], rows[b], stat)

def sweep_2level(X, r, sizes, stat):
    nb = X.shape[0] // r
    assert sum(sizes) == nb
    rows = [list(range(b * r, (b + 1) * r)) for b in range(nb)]
    sbs, o = [], 0
    for s in sizes: sbs.append(list(range(o, o + s))); o += s
    for s1, st in enumerate(tournament(len(sbs))):
        for A, B in st:
            sa, sb = sbs[A], sbs[B]
            if s1 == 0:
                allb = sa + sb
                for s2, st2 in enumerate(tournament(len(allb))):
                    for a, b in st2:
                        if s2 == 0: full(X, rows[allb[a]] + rows[allb[b]], stat)
                        else: cross(X, rows[allb[a]], rows[allb[b]], stat)
            else:
                for t in range(max(len(sa), len(sb))):
                    used = set()
                    for i in range(len(sa)):
                        j = (i + t) % len(sb)
                        cross(X, rows[sa[i]], rows[sb[j]], stat)

def run(kind, A, r, sizes, tol=2e-4):
    X = A.copy()
    for sw in range(40):
        stat = [0.0, 0.25 * tol]
        if kind == "flat": sweep_flat(X, r, stat)
        else: sweep_2level(X, r, sizes, stat)
        print(f"  {kind} sweep {sw + 1}: max cos {stat[0]:.3e}", flush=True)
        if stat[0] < tol: return sw + 1, X
    return -1, X

if __name__ == "__main__":
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    nb = 34
    L = nb * r; M = int(L * 1920 / 1088)
    rng = np.random.default_rng(1)
    for name, A in (("noise", rng.integers(0, 256, (L, M)).astype(np.float64)),
                    ("smooth", (np.outer(np.linspace(0, 1, L), np.ones(M)) * 100 + 50 * np.sin(np.outer(np.arange(L), np.arange(M)) / 37.0) + rng.normal(0, 2, (L, M))))):
        print(name, L, M)
        s0 = np.linalg.svd(A, compute_uv=False)
        for kind, sizes in (("flat", None), ("2level", [6, 4, 4, 4, 4, 4, 4, 4]), ("2level", [6, 6, 6, 6, 6, 4])):
            t = time.time()
            n, X = run(kind, A, r, sizes)
            s = np.sort(np.linalg.norm(X, axis=1))[::-1]
            print(f" {kind} {sizes}: sweeps {n}  sigma err {np.max(np.abs(s - s0)) / s0[0]:.2e}  ({time.time() - t:.1f}s)")
