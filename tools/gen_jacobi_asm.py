#!/usr/bin/env python3
"""Generates csrc/wm_jacobi_gfx950.inc: the packed one-sided Jacobi of the tile kernels
(wm_tile_math.h: raw_to_pk + jacobi_cols_pk + the final col_norms2_pk) as ONE hand-scheduled gfx950
instruction stream in a single inline-asm statement with pinned registers.

Why not leave it to hipcc (profiles/r02_embed_variants.md): for the C++ form it SLP-packs the rotation's
scalar angle arithmetic into v_pk ops stitched together with v_mov, spends an `s_nop` after most packed
instructions (gfx940 forwarding hazard of VOP3P results), keeps ~140 VGPRs live (3 waves per SIMD) and
spills as soon as the sweep is split into basic blocks.  Here the whole iteration lives in a fixed
register file - B in v[40:103], the column norms in v[104:111], 16 temporaries in v[112:127] - so the
kernel fits 128 VGPRs (4 waves per SIMD), and every hazard the stream has is resolved by ORDER:
  * a VOP3P result is never read by the next instruction            (DstSel forwarding hazard, 1 state)
  * a v_rsq_f32 result is never read by the next instruction        (trans-use hazard, 1 state)
  * VCC written by v_cmp is read by v_cndmask two instructions later (VALU SGPR write -> VALU read, 2 states)
The generator checks these three rules on the stream it emits and pads with s_nop where the order
does not already satisfy them (it never needs to inside a rotation).

Same arithmetic as jacobi_rot_pk (two-rsq angle, de Rijk swap, cancellation-free norm update, wave-uniform
skip of a pair that is below the skip threshold in all 64 tiles), same pair order (row-cyclic), same sweep
control (norms recomputed before odd sweeps, convergence test on cos^2, sweep bound 12).

A second stream, csrc/wm_jacobi_v_gfx950.inc (build_v), is jacobi_cols_pk_v: the same rotation applied to the
stacked rows of B and V (v[40:103], v[104:167]), every pair tested and rotated, a converged lane leaving the
exec mask - the literal path of rank-deficient tiles and the watermark-side SVD.

    python tools/gen_jacobi_asm.py          # rewrites both .inc files next to the kernels
"""
import os
import struct
import sys

MAX_SWEEPS = 12


class Lay:
    """Register file of one stream: B at v[a0 ..], optionally V at v[v0 ..], norms, 16 temporaries at v[t ..]."""

    def __init__(self, a0, n0, t, v0=None, vn0=None):
        self.A0, self.N0, self.T, self.V0, self.VN0 = a0, n0, t, v0, vn0
        pr = lambda k: f"v[{t + k}:{t + k + 1}]"
        self.GV, self.T1, self.T2, self.CS, self.T3 = pr(0), pr(2), pr(4), pr(6), pr(14)
        self.GVl, self.GVh, self.C, self.S = f"v{t}", f"v{t + 1}", f"v{t + 6}", f"v{t + 7}"
        (self.g, self.gg, self.ab, self.tau, self.ta, self.t1, self.ih, self.x) = (f"v{t + k}" for k in range(8, 16))

    def A(self, rp, c, base=None):
        b = (self.A0 if base is None else base) + 2 * (8 * rp + c)
        return f"v[{b}:{b + 1}]"

    def V(self, rp, c):
        return self.A(rp, c, self.V0)

    def Alo(self, rp, c):
        return f"v{self.A0 + 2 * (8 * rp + c)}"

    def Ahi(self, rp, c):
        return f"v{self.A0 + 2 * (8 * rp + c) + 1}"

    def N(self, c, base=None):
        return f"v{(self.N0 if base is None else base) + c}"


# the V-free stream (embed / sigma kernels): B in v[40:103], n2 in v[104:111], temporaries v[112:127]
LAY = Lay(40, 104, 112)
A0, N0, T = LAY.A0, LAY.N0, LAY.T
# the stream with V (fallback embed, watermark-side SVD): B v[40:103], V v[104:167], |b|^2 v[168:175], |v|^2 v[176:183],
# temporaries v[184:199]
LAY_V = Lay(40, 168, 184, v0=104, vn0=176)
# SGPRs (clobbered): mask, sweep counter, constants
M, SW, EPS, CONV, SKIPC, MINSW, SKIPFROM = "s[80:81]", "s82", "s83", "s84", "s85", "s86", "s87"
TMPM, NOSKIP, TMPS = "s[88:89]", "s[90:91]", "s92"
DEFI_FROM = 6           # from this many sweeps on, a lane whose tile is rank deficient no longer keeps its wave iterating
SAVE = "s[88:89]"       # with-V stream: the caller's exec mask (TMPM / NOSKIP are unused there)


class Stream:
    """Instruction list with the three ordering rules checked as it grows."""

    def __init__(self):
        self.lines = []
        self.hist = []          # (kind, set of written regs) of the last few VALU-slot instructions

    @staticmethod
    def regs(op):
        op = op.strip().lstrip("|-").rstrip("|")
        if op.startswith("v["):
            a, b = op[2:-1].split(":")
            return {f"v{i}" for i in range(int(a), int(b) + 1)}
        if op.startswith("v") and op[1:].isdigit():
            return {op}
        if op == "vcc":
            return {"vcc"}
        return set()

    def emit(self, text, kind="valu", reads_vcc=False):
        if text.endswith(":") or kind in ("salu", "label"):
            self.lines.append(text)
            if kind == "salu":
                self.hist.append(("salu", set()))
            return
        mn, rest = text.split(None, 1)
        ops = [o.strip() for o in rest.split(" op_sel")[0].split(" neg_")[0].split(",")]
        dst, srcs = ops[0], ops[1:]
        read = set()
        for s_ in srcs:
            read |= self.regs(s_)
        if reads_vcc:
            read.add("vcc")
        need = 0
        for d, (k, w) in enumerate(reversed(self.hist[-3:]), start=1):
            if not (w & read):
                continue
            if k in ("pk", "trans") and d < 2:
                need = max(need, 2 - d)
            if k == "vcmp" and "vcc" in (w & read) and mn.startswith("v_") and d < 3:
                need = max(need, 3 - d)
        if need:
            self.lines.append(f"s_nop {need - 1}")
            for _ in range(need):
                self.hist.append(("nop", set()))
        k = "pk" if mn.startswith("v_pk_") else "trans" if mn.startswith("v_rsq") else "vcmp" if mn.startswith("v_cmp") else "valu"
        self.lines.append(text)
        self.hist.append((k, self.regs(dst) | ({"vcc"} if k == "vcmp" else set())))

    def nops(self):
        return sum(1 for ln in self.lines if ln.startswith("s_nop"))


def col_norms(st, L=LAY, of_v=False):
    """n2[c] = sum over rows of a[.][c]^2: four independent chains at a time (no packed result is read
    by the instruction after its producer).  of_v: the norms of V's columns into vn2."""
    tmps = [L.GV, L.T1, L.T2, L.CS]
    mat = (lambda rp, c: L.V(rp, c)) if of_v else (lambda rp, c: L.A(rp, c))
    nb = L.VN0 if of_v else L.N0
    for c0 in (0, 4):
        for i in range(4):
            st.emit(f"v_pk_mul_f32 {tmps[i]}, {mat(0, c0 + i)}, {mat(0, c0 + i)}")
        for rp in (1, 2, 3):
            for i in range(4):
                st.emit(f"v_pk_fma_f32 {tmps[i]}, {mat(rp, c0 + i)}, {mat(rp, c0 + i)}, {tmps[i]}")
        for i in range(4):
            lo = tmps[i][2:-1].split(":")[0]
            st.emit(f"v_add_f32_e32 {L.N(c0 + i, nb)}, v{lo}, v{int(lo) + 1}")


def rotation(st, p, q, uid, plain=False, L=LAY, with_v=False):
    """One Jacobi rotation of columns p < q (jacobi_rot_pk<CHECK, SKIP>); plain: no convergence test, no skip
    (the sweeps that can never be the last one); with_v: tested, never skipped, V's columns rotated too
    (jacobi_rot_pk_v<1>)."""
    A, N = L.A, L.N
    GV, T1, T2, CS, T3, GVl, GVh, C, S = L.GV, L.T1, L.T2, L.CS, L.T3, L.GVl, L.GVh, L.C, L.S
    g, gg, ab, tau, ta, t1, ih, x = L.g, L.gg, L.ab, L.tau, L.ta, L.t1, L.ih, L.x
    Np, Nq = N(p), N(q)
    st.emit(f"v_pk_mul_f32 {GV}, {A(0, p)}, {A(0, q)}")
    st.emit(f"v_max_f32_e32 {ab}, {Np}, {Nq}" if plain else f"v_mul_f32_e32 {ab}, {Np}, {Nq}")
    st.emit(f"v_pk_fma_f32 {GV}, {A(1, p)}, {A(1, q)}, {GV}")
    st.emit(f"v_sub_f32_e32 {tau}, {Nq}, {Np}")                       # tau = be - al
    st.emit(f"v_pk_fma_f32 {GV}, {A(2, p)}, {A(2, q)}, {GV}")
    st.emit(f"v_add_f32_e64 {ta}, |{tau}|, {EPS}")                     # |tau| + 1e-18
    st.emit(f"v_pk_fma_f32 {GV}, {A(3, p)}, {A(3, q)}, {GV}")
    st.emit(f"v_mul_f32_e32 {t1}, {ta}, {ta}")
    st.emit(f"v_add_f32_e32 {g}, {GVl}, {GVh}")                        # g = a_p . a_q
    if plain:
        st.emit(f"v_mul_f32_e32 {gg}, {g}, {g}")
    elif with_v:
        st.emit(f"v_mul_f32_e32 {x}, {CONV}, {ab}")
        st.emit(f"v_mul_f32_e32 {gg}, {g}, {g}")
        st.emit(f"v_cmp_gt_f32_e32 vcc, {gg}, {x}")                    # notconv |= g^2 > conv2 al be (active lanes only)
        st.emit(f"s_or_b64 {M}, {M}, vcc", kind="salu")
    else:
        st.emit(f"v_mul_f32_e32 {x}, {CONV}, {ab}")
        st.emit(f"v_mul_f32_e32 {gg}, {g}, {g}")
        st.emit(f"v_mul_f32_e32 {ih}, {SKIPC}, {ab}")
        st.emit(f"v_cmp_gt_f32_e32 vcc, {gg}, {x}")                    # notconv |= g^2 > conv2 al be
        st.emit(f"s_or_b64 {M}, {M}, vcc", kind="salu")
        st.emit(f"v_cmp_gt_f32_e32 vcc, {gg}, {ih}")                   # any tile above the skip threshold ...
        st.emit(f"s_or_b64 {TMPM}, vcc, {NOSKIP}", kind="salu")        # ... or a sweep that rotates every pair (SCC = result != 0)
        st.emit(f"s_cbranch_scc0 .Lwmj_skip_{uid}_%=", kind="salu")
    st.emit(f"v_fma_f32 {t1}, {gg}, 4.0, {t1}")                        # h^2 = tau^2 + 4 g^2
    st.emit(f"v_rsq_f32_e32 {ih}, {t1}")                               # 1/h
    st.emit(f"v_mul_f32_e32 {x}, 0.5, {ta}")
    st.emit(f"v_fma_f32 {x}, {x}, {ih}, 0.5")                          # cos^2 in [0.5, 1]
    st.emit(f"v_rsq_f32_e32 {t1}, {x}")                                # rx
    st.emit(f"v_mul_f32_e32 {GVh}, {g}, {ih}")
    st.emit(f"v_mul_f32_e32 {GVl}, {x}, {t1}")                         # c0 = cos
    st.emit(f"v_mul_f32_e32 {GVh}, {GVh}, {t1}")                       # s0 = sin * sign(g)
    st.emit(f"v_cmp_lt_f32_e32 vcc, 0, {tau}")                         # de Rijk: |a_q| > |a_p| -> swap roles
    st.emit(f"v_mul_f32_e32 {gg}, {GVh}, {t1}")                        # t = s0 / c0 ...
    if plain:
        st.emit(f"v_min_f32_e32 {ta}, {Np}, {Nq}")                     # (ta is dead after cos^2)
    else:
        st.emit(f"v_max_f32_e32 {ab}, {Np}, {Nq}")
    st.emit(f"v_cndmask_b32_e32 {C}, {GVl}, {GVh}, vcc", reads_vcc=True)   # C = sw ? s0 : c0
    st.emit(f"v_cndmask_b32_e32 {S}, {GVh}, {GVl}, vcc", reads_vcc=True)   # S = sw ? c0 : s0
    st.emit(f"v_mul_f32_e32 {gg}, {gg}, {g}")                          # ... w = t g  (|w| below)
    if not plain:
        st.emit(f"v_min_f32_e32 {ta}, {Np}, {Nq}")
    st.emit(f"v_add_f32_e64 {Np}, {ab}, |{gg}|")                       # larger norm grows by |t g|
    st.emit(f"v_sub_f32_e64 {Nq}, {ta}, |{gg}|")
    # columns: a_p <- C a_p + S a_q ; a_q <- C a_q - S a_p   (C, S broadcast from the halves of CS by op_sel)
    mats = [L.A] + ([L.V] if with_v else [])
    for mat in mats:
        for rp0 in (0, 2):
            tt = ((T1, T2), (GV, T3))
            for i in (0, 1):
                rp = rp0 + i
                st.emit(f"v_pk_mul_f32 {tt[i][0]}, {CS}, {mat(rp, q)} op_sel:[1,0]")
                st.emit(f"v_pk_mul_f32 {tt[i][1]}, {CS}, {mat(rp, p)} op_sel:[1,0]")
            for i in (0, 1):
                rp = rp0 + i
                st.emit(f"v_pk_fma_f32 {mat(rp, p)}, {CS}, {mat(rp, p)}, {tt[i][0]} op_sel_hi:[0,1,1]")
                st.emit(f"v_pk_fma_f32 {mat(rp, q)}, {CS}, {mat(rp, q)}, {tt[i][1]} op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]")
    if not plain and not with_v:
        st.emit(f".Lwmj_skip_{uid}_%=:", kind="label")
        st.hist.append(("nop", set()))       # a taken branch lands here: nothing before it may be assumed


def eps_literal():
    return struct.unpack("<I", struct.pack("<f", 1e-18))[0]  # keeps 0/0 out; its square (1e-36) is a normal float


def build():
    st = Stream()
    e = st.emit
    Alo, Ahi = LAY.Alo, LAY.Ahi
    # inputs: %[lo0..lo7], %[hi0..hi7] raw row words; %[conv] %[skip] %[minsw] %[skipfrom]
    e(f"s_mov_b32 {SW}, 0", kind="salu")
    e(f"s_mov_b32 {EPS}, 0x{eps_literal():08x}", kind="salu")
    e(f"s_mov_b32 {CONV}, %[conv]", kind="salu")
    e(f"s_mov_b32 {SKIPC}, %[skip]", kind="salu")
    e(f"s_mov_b32 {MINSW}, %[minsw]", kind="salu")
    e(f"s_mov_b32 {SKIPFROM}, %[skipfrom]", kind="salu")
    for r in range(8):
        for c in range(8):
            src = f"%[lo{r}]" if c < 4 else f"%[hi{r}]"
            dst = Alo(r >> 1, c) if (r & 1) == 0 else Ahi(r >> 1, c)
            e(f"v_cvt_f32_ubyte{c & 3}_e32 {dst}, {src}")
    e(".Lwmj_sweep_%=:", kind="label")
    e(f"s_mov_b64 {M}, 0", kind="salu")
    e(f"s_bitcmp1_b32 {SW}, 0", kind="salu")                 # norms are recomputed before odd-numbered sweeps
    e("s_cbranch_scc1 .Lwmj_nonorm_%=", kind="salu")
    col_norms(st)
    e(".Lwmj_nonorm_%=:", kind="label")
    st.hist.append(("nop", set()))
    e(f"s_cmp_ge_i32 {SW}, {SKIPFROM}", kind="salu")         # pairs are skipped from sweep `skipfrom` on (0-based);
    e(f"s_cselect_b64 {NOSKIP}, 0, -1", kind="salu")         # before that every pair is rotated (zero columns must still be sorted)
    # sweeps 0 .. min_sweeps - 2 can never be the last one: they run the body without test and skip logic
    e(f"s_add_i32 {TMPS}, {SW}, 1", kind="salu")
    e(f"s_cmp_lt_i32 {TMPS}, {MINSW}", kind="salu")
    e("s_cbranch_scc0 .Lwmj_tested_%=", kind="salu")
    for p in range(7):
        for q in range(p + 1, 8):
            rotation(st, p, q, -1, plain=True)
    e("s_branch .Lwmj_swept_%=", kind="salu")
    e(".Lwmj_tested_%=:", kind="label")
    st.hist.append(("nop", set()))
    k = 0
    for p in range(7):
        for q in range(p + 1, 8):
            rotation(st, p, q, k)
            k += 1
    e(".Lwmj_swept_%=:", kind="label")
    st.hist.append(("nop", set()))
    e(f"s_add_i32 {SW}, {SW}, 1", kind="salu")
    e(f"s_cmp_lt_i32 {SW}, {MINSW}", kind="salu")             # the first sweeps are never the last
    e("s_cbranch_scc1 .Lwmj_sweep_%=", kind="salu")
    # A rank-deficient tile's null columns are rounding residue: their mutual cosines never fall and their TRACKED norms
    # cancel to garbage, so such a lane would hold its wave to the sweep bound and report non-convergence on a perfectly
    # good image (flat-and-gradient UI content: tests/test_gpu_parity.py::test_random_scenes_every_tile_class).  After
    # DEFI_FROM sweeps - every full-rank tile is long done - a lane with n2[7] <= 1e-10 n2[0] (the tile is flagged and
    # completed from its good columns anyway) leaves the convergence mask.
    e(f"s_cmp_lt_i32 {SW}, {DEFI_FROM}", kind="salu")
    e("s_cbranch_scc1 .Lwmj_nodef_%=", kind="salu")
    e(f"v_mul_f32_e32 {LAY.x}, 0x{struct.unpack('<I', struct.pack('<f', 1e-10))[0]:08x}, {LAY.N(0)}")
    e(f"v_mov_b32_e32 {LAY.ih}, {LAY.N(7)}")                    # (keeps the compare's operands one instruction apart)
    e(f"v_cmp_gt_f32_e32 vcc, {LAY.ih}, {LAY.x}")               # n2[7] > 1e-10 n2[0]: a full-rank tile
    e(f"s_and_b64 {M}, {M}, vcc", kind="salu")
    e(".Lwmj_nodef_%=:", kind="label")
    st.hist.append(("nop", set()))
    e(f"s_cmp_eq_u64 {M}, 0", kind="salu")
    e("s_cbranch_scc1 .Lwmj_done_%=", kind="salu")
    e(f"s_cmp_lt_i32 {SW}, {MAX_SWEEPS}", kind="salu")
    e("s_cbranch_scc1 .Lwmj_sweep_%=", kind="salu")
    e(".Lwmj_done_%=:", kind="label")
    st.hist.append(("nop", set()))
    col_norms(st)
    e(f"s_mov_b64 %[more], {M}", kind="salu")
    return st


def build_v():
    """jacobi_cols_pk_v: B = A V with V accumulated from the identity, norms recomputed before every sweep, every
    pair tested and rotated; a lane (tile) whose own sweep saw nothing to rotate leaves the exec mask and sits out
    the sweeps its wave neighbours still need, so its result does not depend on which tiles share the wave."""
    L = LAY_V
    st = Stream()
    e = st.emit
    e(f"s_mov_b64 {SAVE}, exec", kind="salu")
    e(f"s_mov_b32 {SW}, 0", kind="salu")
    e(f"s_mov_b32 {EPS}, 0x{eps_literal():08x}", kind="salu")
    e(f"s_mov_b32 {CONV}, %[conv]", kind="salu")
    for rp in range(4):
        for c in range(8):
            b = L.V0 + 2 * (8 * rp + c)
            e(f"v_mov_b32_e32 v{b}, {'1.0' if 2 * rp == c else '0'}")
            e(f"v_mov_b32_e32 v{b + 1}, {'1.0' if 2 * rp + 1 == c else '0'}")
    e(".Lwmv_sweep_%=:", kind="label")
    st.hist.append(("nop", set()))
    e(f"s_mov_b64 {M}, 0", kind="salu")
    col_norms(st, L)
    for p in range(7):
        for q in range(p + 1, 8):
            rotation(st, p, q, -1, L=L, with_v=True)
    e(f"s_add_i32 {SW}, {SW}, 1", kind="salu")
    e(f"s_and_b64 exec, exec, {M}", kind="salu")              # lanes that rotated something stay (SCC = any left)
    e("s_cbranch_scc0 .Lwmv_done_%=", kind="salu")
    e(f"s_cmp_lt_i32 {SW}, {MAX_SWEEPS}", kind="salu")
    e("s_cbranch_scc1 .Lwmv_sweep_%=", kind="salu")
    e(".Lwmv_done_%=:", kind="label")
    st.hist.append(("nop", set()))
    e(f"s_mov_b64 %[more], exec", kind="salu")                # non-zero only when the sweep bound was hit
    e(f"s_mov_b32 %[sweeps], {SW}", kind="salu")
    e(f"s_mov_b64 exec, {SAVE}", kind="salu")
    col_norms(st, L)
    col_norms(st, L, of_v=True)
    return st


HEADER = """// GENERATED by tools/gen_jacobi_asm.py - do not edit.  The packed one-sided Jacobi of the tile kernels
// (raw_to_pk + jacobi_cols_pk + final col_norms2_pk of wm_tile_math.h) as one gfx950 instruction stream with
// pinned registers: B = X V in v[40:103] (a[rp][c] = v[40 + 2 (8 rp + c)] : rows 2 rp, 2 rp + 1),
// |b_c|^2 in v[104:111], temporaries v[112:127], control in s[80:92].  {n_inst} instructions, {n_nop} s_nop.
//   conv2:     a sweep that saw no pair with cos^2 > conv2 in any tile of the wave is the last one
//   skip2:     from sweep `skip_from` (0-based) on, a pair below skip2 in every tile of the wave is left alone
//   min_sweeps: sweeps that run regardless of the test (the first ones never pass it)
// Returns the wave's not-converged mask after the last sweep (non-zero only when the bound of {max_sw} is hit).
__device__ __forceinline__ unsigned long long jacobi_cols_gfx950(const uint32_t (&lo)[8], const uint32_t (&hi)[8],
                                                                 wm::v2f (&a)[4][8], float (&n2)[8], const float conv2,
                                                                 const float skip2, const int min_sweeps,
                                                                 const int skip_from) {{
  unsigned long long more;
  asm volatile(
"""

HEADER_V = """// GENERATED by tools/gen_jacobi_asm.py - do not edit.  jacobi_cols_pk_v of wm_tile_math.h (one-sided Jacobi WITH V:
// the fallback embed of rank-deficient tiles and the watermark-side SVD) as one gfx950 instruction stream with pinned
// registers: B = A V in v[40:103], V in v[104:167] (both [rp][c] -> base + 2 (8 rp + c), rows 2 rp, 2 rp + 1),
// |b_c|^2 in v[168:175], |v_c|^2 in v[176:183], temporaries v[184:199], control in s[80:89].
// {n_inst} instructions, {n_nop} s_nop.  Every pair is tested (cos^2 > conv2) and rotated; norms are recomputed
// before every sweep; a lane whose own sweep rotated nothing leaves the exec mask (restored at the end).
// Included inside namespace wm.  Returns sweeps, negated when the bound of {max_sw} was hit with lanes still active.
__device__ __forceinline__ int jacobi_cols_v_gfx950(v2f (&a)[4][8], v2f (&v)[4][8], float (&n2)[8], float (&vn2)[8],
                                                    const float conv2) {{
  unsigned long long more;
  int sweeps;
  asm volatile(
"""


def csrc_path(name):
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                        "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd", "csrc", name)


def n_instructions(st):
    return sum(1 for ln in st.lines if not ln.endswith(":"))


def render():
    st = build()
    body = HEADER.format(n_inst=n_instructions(st), n_nop=st.nops(), max_sw=MAX_SWEEPS)
    for ln in st.lines:
        body += f'      "{ln}\\n\\t"\n'
    outs = []
    for rp in range(4):
        for c in range(8):
            b = A0 + 2 * (8 * rp + c)
            outs.append(f'"=&{{v[{b}:{b + 1}]}}"(a[{rp}][{c}])')
    for c in range(8):
        outs.append(f'"=&{{v{N0 + c}}}"(n2[{c}])')
    outs.append('[more] "=&s"(more)')
    ins = [f'[lo{r}] "v"(lo[{r}])' for r in range(8)] + [f'[hi{r}] "v"(hi[{r}])' for r in range(8)]
    ins += ['[conv] "s"(conv2)', '[skip] "s"(skip2)', '[minsw] "s"(min_sweeps)', '[skipfrom] "s"(skip_from)']
    clob = [f'"v{i}"' for i in range(T, 128)] + [f'"s{i}"' for i in range(80, 93)] + ['"vcc"', '"scc"']
    body += "      : " + ",\n        ".join(outs) + "\n"
    body += "      : " + ",\n        ".join(ins) + "\n"
    body += "      : " + ", ".join(clob) + ");\n  return more;\n}\n"
    return st, body


def render_v():
    L = LAY_V
    st = build_v()
    body = HEADER_V.format(n_inst=n_instructions(st), n_nop=st.nops(), max_sw=MAX_SWEEPS)
    for ln in st.lines:
        body += f'      "{ln}\\n\\t"\n'
    outs = []
    for rp in range(4):
        for c in range(8):
            b = L.A0 + 2 * (8 * rp + c)
            outs.append(f'"+{{v[{b}:{b + 1}]}}"(a[{rp}][{c}])')
    for rp in range(4):
        for c in range(8):
            b = L.V0 + 2 * (8 * rp + c)
            outs.append(f'"=&{{v[{b}:{b + 1}]}}"(v[{rp}][{c}])')
    for c in range(8):
        outs.append(f'"=&{{v{L.N0 + c}}}"(n2[{c}])')
    for c in range(8):
        outs.append(f'"=&{{v{L.VN0 + c}}}"(vn2[{c}])')
    outs += ['[more] "=&s"(more)', '[sweeps] "=&s"(sweeps)']
    ins = ['[conv] "s"(conv2)']
    clob = [f'"v{i}"' for i in range(L.T, L.T + 16)] + [f'"s{i}"' for i in range(80, 90)] + ['"vcc"', '"scc"']
    body += "      : " + ",\n        ".join(outs) + "\n"
    body += "      : " + ",\n        ".join(ins) + "\n"
    body += "      : " + ", ".join(clob) + ");\n  return more ? -sweeps : sweeps;\n}\n"
    return st, body


def main():
    import argparse
    ap = argparse.ArgumentParser(description="Generate the gfx950 Jacobi instruction streams (csrc/wm_jacobi*_gfx950.inc). "
                                 "Default: compare with the committed files and write nothing.")
    ap.add_argument("--write", action="store_true", help="rewrite a file whose content differs (an identical file is "
                    "never touched: its mtime would otherwise make build() recompile libwmhip.so)")
    ap.add_argument("--check", action="store_true", help="compare only (the default); exit status 1 on a difference")
    args = ap.parse_args()
    differs = False
    for name, fn in (("wm_jacobi_gfx950.inc", render), ("wm_jacobi_v_gfx950.inc", render_v)):
        st, body = fn()
        out = csrc_path(name)
        old = open(out).read() if os.path.exists(out) else None
        what = f"{os.path.normpath(out)}: {n_instructions(st)} instructions, {st.nops()} s_nop"
        if old == body:
            print("up to date  " + what, file=sys.stderr)
        elif args.write and not args.check:
            open(out, "w").write(body)
            print("wrote       " + what, file=sys.stderr)
        else:
            differs = True
            print("DIFFERS     " + what + "   (run with --write to regenerate)", file=sys.stderr)
    return 1 if differs else 0


if __name__ == "__main__":
    sys.exit(main())
