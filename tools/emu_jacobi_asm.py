#!/usr/bin/env python3
"""Scalar float32 emulator of the instruction stream tools/gen_jacobi_asm.py generates (one lane = one tile,
any number of lanes side by side as NumPy vectors; VCC / branches are wave-uniform exactly like on the GPU).

Test infrastructure: lets the CPU suite check the hand-written gfx950 stream (arithmetic, operand selects,
sweep control) against float64 LAPACK without a GPU.  Only the ~20 opcodes the stream uses are modelled;
v_rsq_f32 is 1/sqrt in float32 (the hardware's is 1 ulp), denormal INPUTS of v_rsq_f32 read as zero.
"""
import re

import numpy as np

import gen_jacobi_asm as G

F = np.float32


def _interp(st, n, named, lo=None, hi=None, v_init=None, n_vregs=128):
    """Runs the stream on n lanes; returns (vector registers [n_vregs, n], more mask, sweeps)."""
    v = np.zeros((n_vregs, n), F)
    if v_init:
        for r_, val in v_init.items():
            v[r_] = val
    s = {}
    vcc = np.zeros(n, bool)
    ex = np.ones(n, bool)            # exec mask: VALU results land in active lanes only
    scc = False
    more = np.zeros(n, bool)
    lines = st.lines
    labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}

    def wr(reg, val):
        v[reg] = np.where(ex, val, v[reg])

    def sval(tok):
        tok = tok.strip()
        if tok in named:
            return named[tok]
        if tok.startswith("0x"):
            return int(tok, 16)
        if re.fullmatch(r"-?\d+", tok):
            return int(tok)
        if tok.startswith("s["):
            return s.get(tok, 0)
        return s.get(tok, 0)

    def as_f32(x):
        if isinstance(x, (int, np.integer)):
            return np.frombuffer(np.uint32(x & 0xFFFFFFFF).tobytes(), F)[0]
        return F(x)

    def src(tok):
        """32-bit VALU source operand -> float32 vector"""
        tok = tok.strip()
        neg = tok.startswith("-")
        tok = tok.lstrip("-")
        ab = tok.startswith("|")
        tok = tok.strip("|")
        if tok.startswith("v"):
            x = v[int(tok[1:])]
        elif tok.startswith("s"):
            x = np.full(n, as_f32(s[tok]), F)
        elif tok.startswith("0x"):                    # 32-bit literal: the bit pattern of a float
            x = np.full(n, as_f32(int(tok, 16)), F)
        else:
            x = np.full(n, F(float(tok)), F)
        if ab:
            x = np.abs(x)
        return -x if neg else x

    def pair(tok):
        a, b = tok.strip()[2:-1].split(":")
        return int(a), int(b)

    def mods(text, name, nops, default):
        m = re.search(name + r":\[([01,]+)\]", text)
        if not m:
            return [default] * nops
        vals = [int(t) for t in m.group(1).split(",")]
        return vals + [default] * (nops - len(vals))

    pc, sweeps, guard = 0, 0, 0
    while pc < len(lines):
        guard += 1
        assert guard < 400000, "emulator ran away"
        ln = lines[pc]
        pc += 1
        if ln.endswith(":"):
            continue
        mn, rest = ln.split(None, 1)
        opnds = [o.strip() for o in re.split(r"\s+(?:op_sel|neg_)", rest)[0].split(",")]
        with np.errstate(all="ignore"):
            if mn.startswith("v_cvt_f32_ubyte"):
                b = int(mn[len("v_cvt_f32_ubyte")])
                m = re.fullmatch(r"%\[(lo|hi)(\d)\]", opnds[1])
                w = (lo if m.group(1) == "lo" else hi)[int(m.group(2))]
                wr(int(opnds[0][1:]), ((w >> np.uint32(8 * b)) & np.uint32(255)).astype(F))
            elif mn.startswith("v_mov_b32"):
                wr(int(opnds[0][1:]), src(opnds[1]))
            elif mn in ("v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32"):
                k = 3 if mn == "v_pk_fma_f32" else 2
                osl, osh = mods(ln, "op_sel", k, 0), mods(ln, "op_sel_hi", k, 1)
                ngl, ngh = mods(ln, "neg_lo", k, 0), mods(ln, "neg_hi", k, 0)
                d0, _ = pair(opnds[0])
                ps = [pair(o)[0] for o in opnds[1:1 + k]]
                L = [(-1 if ngl[i] else 1) * v[ps[i] + osl[i]] for i in range(k)]
                Hh = [(-1 if ngh[i] else 1) * v[ps[i] + osh[i]] for i in range(k)]
                if mn == "v_pk_mul_f32":
                    rl, rh = L[0] * L[1], Hh[0] * Hh[1]
                elif mn == "v_pk_add_f32":
                    rl, rh = L[0] + L[1], Hh[0] + Hh[1]
                else:   # fused multiply-add, one rounding
                    rl = (L[0].astype(np.float64) * L[1].astype(np.float64) + L[2].astype(np.float64)).astype(F)
                    rh = (Hh[0].astype(np.float64) * Hh[1].astype(np.float64) + Hh[2].astype(np.float64)).astype(F)
                wr(d0, rl.astype(F)); wr(d0 + 1, rh.astype(F))
            elif mn.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_max_f32", "v_min_f32")):
                a, b = src(opnds[1]), src(opnds[2])
                op = mn[2:5]
                r = {"mul": a * b, "add": a + b, "sub": a - b, "max": np.maximum(a, b), "min": np.minimum(a, b)}[op]
                wr(int(opnds[0][1:]), r.astype(F))
            elif mn == "v_fma_f32":
                a, b, c = (src(o).astype(np.float64) for o in opnds[1:4])
                wr(int(opnds[0][1:]), (a * b + c).astype(F))
            elif mn.startswith("v_rsq_f32"):
                xx = src(opnds[1]).copy()
                xx[np.abs(xx) < np.finfo(F).tiny] = 0          # v_rsq_f32 does not take denormals
                wr(int(opnds[0][1:]), (F(1) / np.sqrt(xx.astype(F))).astype(F))
            elif mn.startswith("v_cmp_gt_f32"):
                vcc = (src(opnds[1]) > src(opnds[2])) & ex     # inactive lanes read 0
            elif mn.startswith("v_cmp_lt_f32"):
                vcc = (src(opnds[1]) < src(opnds[2])) & ex
            elif mn.startswith("v_cndmask_b32"):
                wr(int(opnds[0][1:]), np.where(vcc, src(opnds[2]), src(opnds[1])))
            elif mn == "s_mov_b32":
                if opnds[0] == "%[sweeps]":
                    sweeps = sval(opnds[1])
                else:
                    s[opnds[0]] = sval(opnds[1])
            elif mn == "s_mov_b64":
                val = np.zeros(n, bool) if opnds[1] == "0" else ex.copy() if opnds[1] == "exec" else s[opnds[1]]
                if opnds[0] == "%[more]":
                    more = val
                elif opnds[0] == "exec":
                    ex = val.copy()
                else:
                    s[opnds[0]] = val
            elif mn == "s_and_b64":
                get = lambda t_: vcc if t_ == "vcc" else ex if t_ == "exec" else s[t_]
                val = get(opnds[1]) & get(opnds[2])
                scc = bool(np.any(val))
                if opnds[0] == "exec":
                    ex = val.copy()
                else:
                    s[opnds[0]] = val
            elif mn == "s_or_b64":
                get = lambda t_: vcc if t_ == "vcc" else s[t_]
                s[opnds[0]] = get(opnds[1]) | get(opnds[2])
                scc = bool(np.any(s[opnds[0]]))
            elif mn == "s_bitcmp1_b32":
                scc = bool((sval(opnds[0]) >> sval(opnds[1])) & 1)
            elif mn == "s_cmp_ge_i32":
                scc = sval(opnds[0]) >= sval(opnds[1])
            elif mn == "s_cmp_lt_i32":
                scc = sval(opnds[0]) < sval(opnds[1])
            elif mn == "s_cmp_eq_u64":
                scc = not bool(np.any(s[opnds[0]]))
            elif mn == "s_cselect_b64":
                pick = opnds[1] if scc else opnds[2]
                s[opnds[0]] = np.ones(n, bool) if pick == "-1" else np.zeros(n, bool)
            elif mn == "s_cbranch_scc0":
                if not scc:
                    pc = labels[opnds[0]]
            elif mn == "s_cselect_b32":
                s[opnds[0]] = sval(opnds[1]) if scc else sval(opnds[2])
            elif mn == "s_add_i32":
                s[opnds[0]] = sval(opnds[1]) + sval(opnds[2])
                if opnds[0] == G.SW:
                    sweeps = s[opnds[0]]
            elif mn == "s_branch":
                pc = labels[opnds[0]]
            elif mn == "s_cbranch_scc1":
                if scc:
                    pc = labels[opnds[0]]
            elif mn == "s_cbranch_vccz":
                if not vcc.any():
                    pc = labels[opnds[0]]
            elif mn == "s_nop":
                pass
            else:
                raise NotImplementedError(ln)
    return v, more, sweeps


def _pack_words(raw_tiles):
    n = raw_tiles.shape[0]
    lo = [np.zeros(n, np.uint32) for _ in range(8)]
    hi = [np.zeros(n, np.uint32) for _ in range(8)]
    for r in range(8):
        for c in range(4):
            lo[r] |= raw_tiles[:, r, c].astype(np.uint32) << np.uint32(8 * c)
            hi[r] |= raw_tiles[:, r, 4 + c].astype(np.uint32) << np.uint32(8 * c)
    return lo, hi


def _unpack(v, base, n):
    a = np.zeros((n, 4, 8, 2), F)
    for rp in range(4):
        for c in range(8):
            b = base + 2 * (8 * rp + c)
            a[:, rp, c, 0], a[:, rp, c, 1] = v[b], v[b + 1]
    return a


def run(raw_tiles: np.ndarray, conv2: float, skip2: float, min_sweeps: int, skip_from: int):
    """raw_tiles uint8 [n, 8, 8] -> (a [n, 4, 8, 2] float32, n2 [n, 8], more mask (bool [n]), sweeps)."""
    n = raw_tiles.shape[0]
    lo, hi = _pack_words(raw_tiles)
    named = {"%[conv]": F(conv2), "%[skip]": F(skip2), "%[minsw]": int(min_sweeps), "%[skipfrom]": int(skip_from)}
    v, more, sweeps = _interp(G.build(), n, named, lo, hi)
    n2 = np.stack([v[G.N0 + c] for c in range(8)], axis=1)
    return _unpack(v, G.A0, n), n2, more, sweeps


def run_v(tiles_f32: np.ndarray, conv2: float):
    """The stream with V (jacobi_cols_v_gfx950).  tiles_f32 float32 [n, 8, 8] (rows r, columns c) ->
    (b [n, 8, 8], v [n, 8, 8], n2 [n, 8], vn2 [n, 8], more mask, sweeps): B = A V, columns orthogonal."""
    L = G.LAY_V
    n = tiles_f32.shape[0]
    init = {}
    for rp in range(4):
        for c in range(8):
            b = L.A0 + 2 * (8 * rp + c)
            init[b], init[b + 1] = tiles_f32[:, 2 * rp, c].astype(F), tiles_f32[:, 2 * rp + 1, c].astype(F)
    vr, more, sweeps = _interp(G.build_v(), n, {"%[conv]": F(conv2)}, v_init=init, n_vregs=256)
    to_rc = lambda x: x.transpose(0, 1, 3, 2).reshape(n, 8, 8)          # [n][rp][c][half] -> [n][row][column]
    n2 = np.stack([vr[L.N0 + c] for c in range(8)], axis=1)
    vn2 = np.stack([vr[L.VN0 + c] for c in range(8)], axis=1)
    return to_rc(_unpack(vr, L.A0, n)), to_rc(_unpack(vr, L.V0, n)), n2, vn2, more, sweeps


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    tiles = rng.integers(0, 256, (64, 8, 8), dtype=np.uint8)
    yy, xx = np.mgrid[0:8, 0:8]
    tiles[0] = (yy + xx) % 2 * 255
    tiles[1] = 9
    tiles[2] = 0
    for conv, skip, mn, sf in ((1e-7, 1e-12, 4, 4), (1e-3, 0.0, 3, 1 << 20)):
        a, n2, more, sw = run(tiles, conv, skip, mn, sf)
        ref = np.linalg.svd(tiles.astype(np.float64), compute_uv=False)
        err = np.abs(np.sqrt(np.maximum(n2, 0)) - ref) / ref[:, :1].clip(1)
        print(f"conv {conv:g}: {sw} sweeps, not-converged lanes {int(np.sum(more))}, max sigma err / s1 {err.max():.2e}, "
              f"finite {bool(np.isfinite(n2).all())}")
