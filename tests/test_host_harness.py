"""CPU checks of the exact per-tile functions the gfx950 kernels are built
from (csrc/wm_tile_math.h compiled with g++, tests/host_harness.cpp) against
the oracle.  The harness is test infrastructure; the product never loads it."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import wm_oracle as o

SIGMA_RTOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hh():
    return C.CDLL(ge.build_host_harness())


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _inputs(H, W, seed=1234):
    host = np.random.default_rng(seed).integers(0, 256, (H, W), dtype=np.uint8)
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    return host, wys


def _rel(a, b):
    a = a.reshape(-1, 8); b = b.reshape(-1, 8)
    return float(np.max(np.abs(a - b) / np.maximum(b[:, :1], 1e-30)))


def test_dct8x8_matches_closed_form(hh):
    t = np.random.default_rng(0).uniform(0, 255, (8, 8)).astype(np.float32)
    a = t.copy(); hh.hh_dct8x8(vp(a), 0)
    D = o.dct_basis(8)
    assert np.abs(a - D @ t @ D.T).max() < 1e-3
    hh.hh_dct8x8(vp(a), 1)
    assert np.abs(a - t).max() < 1e-3


def test_dct8x8_against_the_published_jpeg_example(hh):
    """The kernels' 8-point DCT (wm_tile_math.h, CPU build) on the JPEG literature's worked example
    (tests/golden/external/jpeg_dct_example.npz: an external known answer, two printed decimals)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "external", "jpeg_dct_example.npz"))
    a = g["block"].astype(np.float32).copy()
    hh.hh_dct8x8(vp(a), 0)
    assert np.abs(a - g["dct"]).max() < 6e-3
    hh.hh_dct8x8(vp(a), 1)
    assert np.abs(a - g["block"]).max() < 1e-3


@pytest.mark.parametrize("variant", ["literal", "packed"])
@pytest.mark.parametrize("H,W", [(64, 96), (256, 256)])
def test_embed_tile_math_vs_oracle(hh, variant, H, W):
    alpha = 0.15
    host, wys = _inputs(H, W)
    ref = o.embed_plane(host.astype(np.float32), wys, alpha, 0.6, tile=8)
    nb = (H // 8) * (W // 8)
    sw = np.ascontiguousarray(ref["Sw"].reshape(nb, 8))
    stego = np.empty((H, W), np.uint8); sc = np.empty((nb, 8), np.float32); yw = np.empty((H, W), np.float32)
    ms = C.c_int(0); nf = C.c_int(0)
    if variant == "literal":
        hh.hh_embed_tiles_u8(vp(host), vp(sw), vp(stego), vp(sc), vp(yw), H, W, W, C.c_float(alpha), 8, C.byref(ms))
    else:
        hh.hh_embed_tiles_u8_pk(vp(host), vp(sw), vp(stego), vp(sc), vp(yw), H, W, W, C.c_float(alpha), 8,
                                C.byref(ms), C.byref(nf))
    assert 0 < ms.value <= 7
    assert _rel(sc, ref["Sc"]) < SIGMA_RTOL
    d = np.abs(stego.astype(int) - ref["stego"].astype(int))
    assert d.max() <= 1 and np.mean(d != 0) < 1e-3
    assert np.abs(yw - ref["Yw"]).max() < 3e-2


def test_packed_embed_property_random_parameters(hh):
    """Hypothesis sweep over what the fixed cases do not vary: alpha, the band K, plane size and
    content (iid noise / smooth field + sensor noise).  Packed production math vs the oracle."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
    @given(nby=st.integers(1, 6), nbx=st.integers(1, 8), alpha=st.floats(0.0, 0.3), K=st.integers(0, 8),
           smooth=st.booleans(), seed=st.integers(0, 2 ** 31 - 1))
    def prop(nby, nbx, alpha, K, smooth, seed):
        H, W = 8 * nby, 8 * nbx
        rng = np.random.default_rng(seed)
        if smooth:
            yy, xx = np.mgrid[0:H, 0:W]
            host = np.clip(120 + 60 * np.sin(xx / 9.0 + seed % 7) * np.cos(yy / 7.0) + rng.normal(0, 1.5, (H, W)), 0, 255).astype(np.uint8)
        else:
            host = rng.integers(0, 256, (H, W), dtype=np.uint8)
        wys = rng.integers(0, 256, (H, W)).astype(np.float32)
        ref = o.embed_plane(host.astype(np.float32), wys, alpha, kfrac=0.0, tile=8, k_floor=K)
        nb = nby * nbx
        sw = np.ascontiguousarray(ref["Sw"].reshape(nb, 8))
        stego = np.empty((H, W), np.uint8); sc = np.empty((nb, 8), np.float32); yw = np.empty((H, W), np.float32)
        ms = C.c_int(0); nf = C.c_int(0)
        hh.hh_embed_tiles_u8_pk(vp(host), vp(sw), vp(stego), vp(sc), vp(yw), H, W, W, C.c_float(alpha), K,
                                C.byref(ms), C.byref(nf))
        assert 0 < ms.value <= 9
        assert _rel(sc, ref["Sc"]) < SIGMA_RTOL
        ok = np.kron(ref["Sc"][..., 7] > 1e-5 * ref["Sc"][..., 0], np.ones((8, 8), bool))   # tiles with defined vectors
        d = np.abs(stego.astype(int) - ref["stego"].astype(int))
        assert d[ok].max(initial=0) <= 1
        assert np.abs(yw - ref["Yw"])[ok].max(initial=0.0) < 5e-2

    prop()


def test_sigma_svd_extract_tile_math_vs_oracle(hh):
    H, W, alpha = 128, 160, 0.15
    host, wys = _inputs(H, W)
    ref = o.embed_plane(host.astype(np.float32), wys, alpha, 0.6, tile=8)
    nb = (H // 8) * (W // 8)
    for fn, extra in (("hh_sigma_tiles_u8", ()), ("hh_sigma_tiles_u8_pk", (None,))):
        s = np.empty((nb, 8), np.float32)
        getattr(hh, fn)(vp(ref["stego"]), vp(s), H, W, W, *extra)
        assert _rel(s, o.stego_sigma(ref["stego"].astype(np.float32), 8)) < SIGMA_RTOL
    U = np.empty((nb, 8, 8), np.float32); S = np.empty((nb, 8), np.float32); Vt = np.empty((nb, 8, 8), np.float32)
    hh.hh_svd_tiles_f32(vp(wys), vp(U), vp(S), vp(Vt), H, W, W)
    assert _rel(S, ref["Sw"]) < SIGMA_RTOL
    rec = np.matmul(U * S[:, None, :], Vt)
    reco = np.matmul(ref["Uw"].reshape(nb, 8, 8) * ref["Sw"].reshape(nb, 1, 8), ref["Vwt"].reshape(nb, 8, 8))
    assert np.abs(rec - reco).max() < 5e-3
    out = np.empty((H, W), np.float32)
    hh.hh_extract_tiles_u8(vp(ref["stego"]), vp(np.ascontiguousarray(ref["Sc"])), vp(np.ascontiguousarray(ref["Uw"])),
                           vp(np.ascontiguousarray(ref["Vwt"])), vp(out), H, W, W, C.c_float(alpha), 8)
    wo = o.extract_plane(ref["stego"].astype(np.float32), ref["Sc"], ref["Uw"], ref["Vwt"], alpha, 0.6, H, W, 8)
    assert np.abs(out - wo).max() < 2e-2


def _degenerate_image(H=128, W=128):
    """flat, saturated, rank-1, rank-2 and smooth-noisy regions in one plane"""
    yy, xx = np.mgrid[0:H, 0:W]
    img = (128 + 60 * np.sin(xx / 37.0) + 50 * np.cos(yy / 23.0)
           + np.random.default_rng(5).normal(0, 1.5, (H, W))).clip(0, 255).astype(np.uint8)
    img[:32, :32] = 200; img[32:64, :32] = 0; img[64:96, :32] = 255
    img[96:, :32] = (np.arange(32, dtype=np.uint8) * 3)[None, :]                       # rank 1
    img[96:, 32:64] = (np.arange(32)[:, None] * 5 + np.arange(32)[None, :] * 2).astype(np.uint8)   # rank 2
    mask = np.zeros((H, W), bool); mask[:, :32] = True; mask[96:, 32:64] = True
    return img, mask


def check_completion_properties(img, mask, wys, alpha, stego, sc, yw, ref):
    """Rank-deficient tiles have no unique singular vectors, so parity with LAPACK's
    arbitrary completion is undefined; what IS defined is checked instead:
    sigma(Yw tile) = Sc + alpha*Sw, Sc = the tile's true singular values, the
    injected energy, and 1-LSB parity everywhere else."""
    H, W = img.shape
    nb = (H // 8) * (W // 8)
    sw = ref["Sw"].reshape(nb, 8).astype(np.float64)
    assert np.all(np.isfinite(yw)) and np.all(np.isfinite(sc))
    d = np.abs(stego.astype(int) - ref["stego"].astype(int))
    assert d[~mask].max() <= 1
    T = o.to_tiles(yw).reshape(nb, 8, 8).astype(np.float64)
    X = o.to_tiles(img.astype(np.float64)).reshape(nb, 8, 8)
    sx = np.linalg.svd(X, compute_uv=False)
    scf = sc.reshape(nb, 8).astype(np.float64)
    # exact zeros come out as delta * sigma(pattern) <= 4 * 2^-14 by construction
    assert np.max((np.abs(scf - sx) - 4 * 2.0 ** -14) / np.maximum(sx[:, :1], 1.0)) < 1e-5
    want = np.sort(scf + alpha * sw, axis=1)[:, ::-1]
    got = np.linalg.svd(T, compute_uv=False)
    assert np.max(np.abs(got - want) / np.maximum(want[:, :1], 1.0)) < 1e-4
    inj = np.sqrt(((T - X) ** 2).sum((1, 2)))
    assert np.max(np.abs(inj - alpha * np.sqrt((sw ** 2).sum(1))) / inj) < 1e-3
    assert np.array_equal(stego, np.clip(yw, 0, 255).astype(np.uint8))


def test_rank_deficient_tiles_get_an_orthonormal_completion(hh):
    img, mask = _degenerate_image()
    H, W = img.shape
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    ref = o.embed_plane(img.astype(np.float32), wys, 0.15, 0.6, 8)
    nb = (H // 8) * (W // 8)
    sw = np.ascontiguousarray(ref["Sw"].reshape(nb, 8))
    stego = np.empty((H, W), np.uint8); sc = np.empty((nb, 8), np.float32); yw = np.empty((H, W), np.float32)
    ms = C.c_int(0); nf = C.c_int(0)
    hh.hh_embed_tiles_u8_pk(vp(img), vp(sw), vp(stego), vp(sc), vp(yw), H, W, W, C.c_float(0.15), 8,
                            C.byref(ms), C.byref(nf))
    assert nf.value >= 16 * 4 + 8                      # 4 flat/rank-1 column blocks + rank-2 block
    check_completion_properties(img, mask, wys, 0.15, stego, sc, yw, ref)
    # watermark-side SVD of a degenerate plane: U, Vt stay orthonormal and reconstruct the DCT tile
    wflat = np.zeros((16, 16), np.float32); wflat[:8, :8] = 255; wflat[8:, 8:] = np.arange(8)[None, :]
    U = np.empty((4, 8, 8), np.float32); S = np.empty((4, 8), np.float32); Vt = np.empty((4, 8, 8), np.float32)
    hh.hh_svd_tiles_f32(vp(wflat), vp(U), vp(S), vp(Vt), 16, 16, 16)
    I = np.eye(8)
    assert np.abs(np.matmul(U.transpose(0, 2, 1), U) - I).max() < 1e-5
    assert np.abs(np.matmul(Vt, Vt.transpose(0, 2, 1)) - I).max() < 1e-5
    C0 = o._dct_tiles(o.to_tiles(wflat)).reshape(4, 8, 8)
    assert np.abs(np.matmul(U * S[:, None, :], Vt) - C0).max() < 1e-3


def test_constant_tile_closed_form_equals_the_literal_chain(hh):
    """embed_tile_constant (tabulated completion of a constant tile) against embed_tile_completed on the same tile:
    Yw within 2e-2 grey levels, Sc within 1e-6 * max(8 v, 1) - for black, dark, mid, white tiles, K = 8 and K = 3 -
    and the committed tables are the generator's output."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_completion_tables as gt
    inc = os.path.join(ROOT, "digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd", "csrc",
                       "wm_completion_tables.inc")
    assert open(inc).read() == gt.render()
    rng = np.random.default_rng(3)
    for v in (0.0, 1.0, 16.0, 128.0, 235.0, 255.0):
        for K in (8, 3):
            sw = np.sort(rng.uniform(1.0, 2000.0, 8))[::-1].astype(np.float32).copy()
            ak = np.array([0.15 if i < K else 0.0 for i in range(8)], np.float32)
            yc = np.empty((8, 8), np.float32); yl = np.empty((8, 8), np.float32)
            scc = np.empty(8, np.float32); scl = np.empty(8, np.float32)
            hh.hh_constant_tile_both_ways(C.c_float(v), vp(sw), vp(ak), vp(yc), vp(scc), vp(yl), vp(scl))
            assert np.max(np.abs(yc - yl)) < 2e-2, (v, K, np.max(np.abs(yc - yl)))
            assert np.max(np.abs(scc - scl)) < 1e-6 * max(8 * v, 1.0), (v, K, scc, scl)
            # and the reference's invariant: svd(Yw) = Sc + alpha Sw on the first K (the completion is orthonormal)
            got = np.linalg.svd(yc.astype(np.float64), compute_uv=False)
            want = np.sort(scc.astype(np.float64) + ak * sw)[::-1]
            assert np.max(np.abs(got - want)) < 1e-3 * max(want[0], 1.0)


def test_tile_math_under_sanitizers(tmp_path):
    """AddressSanitizer + UndefinedBehaviorSanitizer over the whole CPU build of the tile arithmetic and its shim
    (tests/host_harness.cpp = csrc/wm_tile_math.h behind the same entry-point shapes as the C ABI) - CPU only: GPU
    sanitizers are not available on this pool.  One driver walks EVERY harness entry
    point - the packed fast path with its flagged-tile classes, the literal chain, sigma-only, the watermark-side SVD,
    extract, the closed forms (constant / rank-1 tiles) and the DCT - over noise, flat, saturated, striped (rank-1) and
    ragged (H, W not multiples of 8, row_stride > W) planes; any out-of-bounds access, misaligned or overflowing
    operation aborts the run (-fno-sanitize-recover=all)."""
    exe = str(tmp_path / "hh_asan")
    main = str(tmp_path / "main.cpp")
    open(main, "w").write(r'''
#include <vector>
#include <cstdio>
#include <cstdint>
#include <cmath>
extern "C" {
int hh_embed_tiles_u8(const uint8_t*, const float*, uint8_t*, float*, float*, int, int, int, float, int, int*);
int hh_sigma_tiles_u8(const uint8_t*, float*, int, int, int);
int hh_svd_tiles_f32(const float*, float*, float*, float*, int, int, int);
int hh_extract_tiles_u8(const uint8_t*, const float*, const float*, const float*, float*, int, int, int, float, int);
int hh_embed_tiles_u8_pk(const uint8_t*, const float*, uint8_t*, float*, float*, int, int, int, float, int, int*, int*);
int hh_sigma_tiles_u8_pk(const uint8_t*, float*, int, int, int, int*);
void hh_constant_tile_both_ways(float, const float*, const float*, float*, float*, float*, float*);
int hh_rank1_tile_both_ways(const uint8_t*, const float*, const float*, float*, float*, float*, float*);
void hh_dct8x8(float*, int);
}
static unsigned lcg(unsigned& x) { x = x * 1664525u + 1013904223u; return x >> 24; }
int main() {
  unsigned seed = 12345;
  int total = 0;
  const int geo[4][3] = {{40, 56, 56}, {43, 61, 64}, {8, 8, 8}, {24, 100, 128}};     // H, W, row_stride (ragged ones too)
  for (int content = 0; content < 5; ++content)
    for (auto& g : geo) {
      const int H = g[0], W = g[1], RS = g[2], nt = (H / 8) * (W / 8);
      std::vector<uint8_t> h((size_t)H * RS), s((size_t)H * RS), s2((size_t)H * RS);
      for (int r = 0; r < H; ++r)
        for (int c = 0; c < RS; ++c) {
          uint8_t v;
          switch (content) {
            case 0: v = (uint8_t)lcg(seed); break;                       // noise
            case 1: v = 77; break;                                       // flat
            case 2: v = (c / 8 % 2) ? 255 : 0; break;                    // saturated blocks
            case 3: v = (uint8_t)(10 + 20 * (c % 8)); break;             // every row equal: rank-1 tiles
            default: v = (r % 8 == 3) ? 200 : 30; break;                 // one-pixel rules: rank-1 tiles of the other kind
          }
          h[(size_t)r * RS + c] = v;
        }
      std::vector<float> sw((size_t)nt * 8 + 8), sc(sw.size()), sc2(sw.size()), yw((size_t)H * W), wm((size_t)H * W);
      for (size_t i = 0; i < sw.size(); ++i) sw[i] = 40.0f / (1 + i % 8);
      for (int K : {8, 3}) {
        int ms = 0, nf = 0, hist[16] = {0};
        total += hh_embed_tiles_u8_pk(h.data(), sw.data(), s.data(), sc.data(), yw.data(), H, W, RS, 0.15f, K, &ms, &nf);
        total += hh_embed_tiles_u8(h.data(), sw.data(), s2.data(), sc2.data(), nullptr, H, W, RS, 0.15f, K, &ms);
        total += hh_sigma_tiles_u8_pk(s.data(), sc2.data(), H, W, RS, hist);
        total += hh_sigma_tiles_u8(s2.data(), sc2.data(), H, W, RS);
        std::vector<float> plane((size_t)H * W), U((size_t)nt * 64 + 64), S((size_t)nt * 8 + 8), Vt((size_t)nt * 64 + 64);
        for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) plane[(size_t)r * W + c] = (float)h[(size_t)r * RS + c];
        total += hh_svd_tiles_f32(plane.data(), U.data(), S.data(), Vt.data(), H, W, W);
        total += hh_extract_tiles_u8(s.data(), sc.data(), U.data(), Vt.data(), wm.data(), H, W, RS, 0.15f, K);
      }
    }
  float swt[8] = {50, 40, 30, 20, 10, 5, 2, 1}, ak[8] = {.15f, .15f, .15f, .15f, .15f, 0, 0, 0};
  float y1[64], y2[64], s1[8], s2[8];
  for (float v : {0.0f, 1.0f, 128.0f, 255.0f}) hh_constant_tile_both_ways(v, swt, ak, y1, s1, y2, s2);
  uint8_t tile[64];
  for (int kind = 0; kind < 3; ++kind) {
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) tile[r * 8 + c] = kind == 0 ? (uint8_t)(3 * r + 1) : kind == 1 ? (uint8_t)(c == 4 ? 250 : 9) : (uint8_t)lcg(seed);
    total += hh_rank1_tile_both_ways(tile, swt, ak, y1, s1, y2, s2);
  }
  float d[64]; for (int i = 0; i < 64; ++i) d[i] = (float)lcg(seed);
  hh_dct8x8(d, 0); hh_dct8x8(d, 1);
  std::printf("ok %d\n", total); return 0; }
''')
    src = os.path.join(ge.ROOT, "tests", "host_harness.cpp")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        src, main, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]


def test_rank1_closed_form(hh):
    """embed_tile_rank1 (csrc/wm_tile_math.h): rank-1 tiles X = a b^T (edges of flat rectangles, rules and their
    crossings, one-axis gradients) are finished analytically with two Householder reflections instead of the Jacobi
    with V.  LAPACK's completion of the seven null directions is arbitrary, so pixels are not comparable with the oracle;
    what the reference guarantees whatever the completion is (single:174-176): svd(Yw) = Sc + alpha Sw with Sc the tile's
    true singular values, and the injected energy alpha |Sw[:K]|.  The literal chain agrees on both.  raw_is_rank1 decides
    exactly on the integers: rank-2 tiles and full-rank tiles are refused, the zero tile too (it is a constant tile)."""
    rng = np.random.default_rng(11)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)
    ones = np.ones(8)
    edge = np.array([10, 10, 10, 200, 200, 200, 200, 200.0])
    line = np.array([1, 1, 1, 0, 1, 1, 1, 1.0])
    tiles = [np.outer(ones, edge), np.outer(edge, ones), np.outer(ones, [0, 0, 0, 0, 255, 0, 0, 0]), np.outer([255, 0, 0, 0, 0, 0, 0, 0], ones),
             240 * np.outer(line, np.roll(line, 3)),                       # a rule crossing on a flat background: rank 1, no equal rows / columns
             np.outer([1, 2, 3, 4, 5, 6, 7, 8], [30, 20, 10, 5, 0, 1, 2, 3]), np.outer(np.arange(8), np.arange(8)) * 5,
             np.zeros((8, 8)) + np.eye(8)[0][:, None] * np.eye(8)[7][None, :] * 3, np.full((8, 8), 77.0)]
    for X in tiles:
        for K in (8, 3):
            sw = f32(np.sort(rng.uniform(1, 2000, 8))[::-1]); ak = f32([0.15 if i < K else 0 for i in range(8)])
            yc = np.empty((8, 8), np.float32); sc_c = np.empty(8, np.float32); yl = np.empty((8, 8), np.float32); sc_l = np.empty(8, np.float32)
            assert hh.hh_rank1_tile_both_ways(vp(u8(X)), vp(sw), vp(ak), vp(yc), vp(sc_c), vp(yl), vp(sc_l)) == 1
            s_true = np.linalg.svd(X, compute_uv=False)
            assert abs(sc_c[0] - s_true[0]) < 2e-6 * s_true[0] and not sc_c[1:].any()
            w = (ak * sw).astype(np.float64)
            want = np.sort(sc_c.astype(np.float64) + w)[::-1]
            got = np.linalg.svd(yc.astype(np.float64), compute_uv=False)
            assert np.abs(got - want).max() < 2e-5 * max(want[0], 1.0), (X, K)
            inj = np.sqrt(((yc - X) ** 2).sum())
            assert abs(inj - np.sqrt((w ** 2).sum())) < 1e-4 * inj
            # the literal chain agrees on what is defined: Sc (up to its delta-sized completion values), the energy
            assert abs(sc_l[0] - sc_c[0]) < 1e-5 * sc_c[0] + 1e-3 and sc_l[1:].max() < 4 * 2.0 ** -14 + 1e-6 * sc_c[0]
            assert abs(np.sqrt(((yl - X) ** 2).sum()) - inj) < 2e-3 * inj
    dummy = [np.empty((8, 8), np.float32), np.empty(8, np.float32), np.empty((8, 8), np.float32), np.empty(8, np.float32)]
    sw = f32(np.arange(8, 0, -1)); ak = f32([0.15] * 8)
    rank2 = np.outer(ones, edge) + np.outer([0, 0, 0, 5, 5, 5, 5, 5], ones)
    off_by_one = np.outer([1, 2, 3, 4, 5, 6, 7, 8], [30, 20, 10, 5, 0, 1, 2, 3]); off_by_one[5, 2] += 1
    for X in (rank2, off_by_one, rng.integers(0, 256, (8, 8)), np.zeros((8, 8)), np.eye(8) * 9):
        assert hh.hh_rank1_tile_both_ways(vp(u8(X)), vp(sw), vp(ak), *[vp(d) for d in dummy]) == 0


def test_rank1_tiles_take_the_closed_form_in_the_plane_path(hh):
    """hh_embed_tiles_u8_pk (the CPU mirror of k_embed_tiles + k_embed_fallback) routes constant tiles and rank-1 tiles
    to their closed forms and everything else that is rank deficient to the literal chain; the completion properties
    hold on all of them."""
    img, mask = _degenerate_image()
    img[8:24, 40:52] = 17; img[8:24, 52:64] = 230                  # a vertical edge inside the tiles of columns 48..55: rows-equal tiles
    img[40:44, 64:96] = 5; img[44:48, 64:96] = 250                 # a horizontal edge: columns-equal tiles
    mask[8:24, 40:64] = True; mask[40:48, 64:96] = True
    H, W = img.shape
    wys = np.random.default_rng(4321).integers(0, 256, (H, W)).astype(np.float32)
    ref = o.embed_plane(img.astype(np.float32), wys, 0.15, 0.6, 8)
    nb = (H // 8) * (W // 8)
    sw = np.ascontiguousarray(ref["Sw"].reshape(nb, 8))
    stego = np.empty((H, W), np.uint8); sc = np.empty((nb, 8), np.float32); yw = np.empty((H, W), np.float32)
    ms = C.c_int(0); nf = C.c_int(0)
    hh.hh_embed_tiles_u8_pk(vp(img), vp(sw), vp(stego), vp(sc), vp(yw), H, W, W, C.c_float(0.15), 8, C.byref(ms), C.byref(nf))
    check_completion_properties(img, mask, wys, 0.15, stego, sc, yw, ref)
    tiles = img.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3)
    rows_eq = (tiles == tiles[:, :, :1, :]).all(axis=(2, 3)); cols_eq = (tiles == tiles[:, :, :, :1]).all(axis=(2, 3))
    r1 = (rows_eq | cols_eq) & ~(rows_eq & cols_eq)
    assert r1.sum() >= 16 + 2 + 4                                  # the rank-1 block of _degenerate_image + the two edges
    assert not sc.reshape(H // 8, W // 8, 8)[r1][:, 1:].any()      # closed form: exact zeros, not the pattern's delta-sized values
    rank2 = np.zeros_like(r1); rank2[12:, 4:8] = True              # _degenerate_image's rank-2 block: completed from B, Sc = the rounding-sized norms of B's null columns
    assert (sc.reshape(H // 8, W // 8, 8)[rank2][:, 2:] > 0).all()


def _one_small_tiles(n_want=24, seed=5):
    """uint8 tiles with exactly one singular value below 1e-5 sigma_1: random tiles that happen to be that close to
    singular (what noise frames contain, ~4 in 10 000) and exactly rank-7 ones (two equal rows, a zero column, a column
    that is the sum of two others)."""
    rng = np.random.default_rng(seed)
    found = []
    while len(found) < n_want:
        t = rng.integers(0, 256, (200000, 8, 8)).astype(np.float64)
        s = np.linalg.svd(t, compute_uv=False)
        sel = (s[:, 7] < 1e-5 * s[:, 0]) & (s[:, 6] > 1e-3 * s[:, 0])
        found += [x.astype(np.uint8) for x in t[sel]]
    near = found[:n_want]
    r7 = []
    for k in range(8):
        a = rng.integers(0, 256, (8, 8)).astype(np.uint8)
        if k % 3 == 0: a[5] = a[2]
        elif k % 3 == 1: a[:, 3] = 0
        else: a[:, 6] = (a[:, 0].astype(int) // 2 + a[:, 1].astype(int) // 2).astype(np.uint8); a[:, 0] = (a[:, 0] // 2) * 2; a[:, 1] = (a[:, 1] // 2) * 2; a[:, 6] = a[:, 0] // 2 + a[:, 1] // 2
        r7.append(a)
    return near, r7


def test_one_small_singular_value_is_completed_without_v(hh):
    """embed_tile_one_small (csrc/wm_tile_math.h): tiles whose eighth singular value is out of the V-free form's reach
    (or exactly 0) are finished from the fast path's own B = X V - seven right vectors as usual, the eighth as the
    orthogonal complement of the seven - instead of the Jacobi with V.  Near-singular FULL-rank tiles have unique
    singular vectors, so they must match the float64 oracle to 1 LSB; rank-7 tiles (sign of the null pair arbitrary)
    must satisfy the reference's invariant svd(Yw) = Sc + alpha Sw."""
    near, r7 = _one_small_tiles()
    tiles = near + r7
    H, W = 8, 8 * len(tiles)
    img = np.concatenate(tiles, axis=1)
    rng = np.random.default_rng(2)
    sw = np.sort(rng.uniform(1, 1500, (len(tiles), 8)).astype(np.float32), axis=-1)[..., ::-1].copy()
    stego = np.empty((H, W), np.uint8); sc = np.empty((len(tiles), 8), np.float32); yw = np.empty((H, W), np.float32)
    ms = C.c_int(0); nf = C.c_int(0)
    hh.hh_embed_tiles_u8_pk(vp(img), vp(sw), vp(stego), vp(sc), vp(yw), H, W, W, C.c_float(0.15), 8, C.byref(ms), C.byref(nf))
    assert nf.value == len(tiles)                                   # every one of them is flagged by the fast path
    T = yw.reshape(8, len(tiles), 8).transpose(1, 0, 2).astype(np.float64)
    X = np.stack(tiles).astype(np.float64)
    sx = np.linalg.svd(X, compute_uv=False)
    assert np.max(np.abs(sc - sx) / sx[:, :1]) < 2e-6
    want = np.sort(sc.astype(np.float64) + 0.15 * sw, axis=1)[:, ::-1]
    got = np.linalg.svd(T, compute_uv=False)
    assert np.max(np.abs(got - want) / want[:, :1]) < 2e-5
    assert np.array_equal(stego, np.clip(yw, 0, 255).astype(np.uint8))
    # the near-singular full-rank ones against float64 LAPACK: U diag(S + alpha Sw) V^T
    U, S, Vt = np.linalg.svd(X[:len(near)])
    ref = (U * (S + 0.15 * sw[:len(near)])[:, None, :]) @ Vt
    d = np.abs(T[:len(near)] - ref)
    assert d.max() < 0.05, d.max()                                  # grey levels; the literal chain manages ~1e-2 here
    q = np.abs(np.clip(ref, 0, 255).astype(np.uint8).astype(int) - np.clip(T[:len(near)], 0, 255).astype(np.uint8).astype(int))
    assert q.max() <= 1
