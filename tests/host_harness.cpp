// CPU build of the per-tile arithmetic in csrc/wm_tile_math.h, for tests only.
// It lets the CPU test-suite (no GPU in the build container) check the exact
// functions the gfx950 kernels are made of against the oracle, and lets
// sanitizers run over them.  Nothing in the product loads this file.
//
//   g++ -O2 -shared -fPIC -o tests/_build/libwm_hostharness.so tests/host_harness.cpp
#include <string.h>
#include "../digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd/csrc/wm_tile_math.h"

using namespace wm;

static inline void load_u8(const uint8_t* p, int stride, float (&a)[8][8]) {
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) a[r][c] = (float)p[r * stride + c];
}

extern "C" {

// same meaning as wm_embed_tiles_u8 (include/wmhip.h), single plane
int hh_embed_tiles_u8(const uint8_t* host, const float* sigma_w, uint8_t* stego, float* sigma_c,
                      float* yw, int H, int W, int row_stride, float alpha, int K, int* max_sweeps) {
  const int nby = H / 8, nbx = W / 8;
  float alpha_k[8];
  for (int i = 0; i < 8; ++i) alpha_k[i] = (i < K) ? alpha : 0.0f;
  int ms = 0;
  for (int r = 0; r < H; ++r) memcpy(stego + (size_t)r * row_stride, host + (size_t)r * row_stride, W);
  if (yw)
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < W; ++c) yw[(size_t)r * W + c] = (float)host[(size_t)r * row_stride + c];
  for (int ty = 0; ty < nby; ++ty)
    for (int tx = 0; tx < nbx; ++tx) {
      const size_t t = (size_t)ty * nbx + tx;
      float a[8][8], sw[8], sc[8];
      load_u8(host + (size_t)ty * 8 * row_stride + tx * 8, row_stride, a);
      for (int i = 0; i < 8; ++i) sw[i] = sigma_w[t * 8 + i];
      const int s = embed_tile(a, sw, alpha_k, sc);
      if (s > ms) ms = s;
      for (int i = 0; i < 8; ++i) sigma_c[t * 8 + i] = sc[i];
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
          stego[(size_t)(ty * 8 + r) * row_stride + tx * 8 + c] = (uint8_t)quant_u8(a[r][c]);
          if (yw) yw[(size_t)(ty * 8 + r) * W + tx * 8 + c] = a[r][c];
        }
    }
  if (max_sweeps) *max_sweeps = ms;
  return 0;
}

int hh_sigma_tiles_u8(const uint8_t* plane, float* sigma, int H, int W, int row_stride) {
  const int nby = H / 8, nbx = W / 8;
  for (int ty = 0; ty < nby; ++ty)
    for (int tx = 0; tx < nbx; ++tx) {
      float a[8][8], s[8];
      load_u8(plane + (size_t)ty * 8 * row_stride + tx * 8, row_stride, a);
      sigma_tile(a, s);
      for (int i = 0; i < 8; ++i) sigma[((size_t)ty * nbx + tx) * 8 + i] = s[i];
    }
  return 0;
}

int hh_svd_tiles_f32(const float* plane, float* U, float* S, float* Vt, int H, int W, int row_stride) {
  const int nby = H / 8, nbx = W / 8;
  for (int ty = 0; ty < nby; ++ty)
    for (int tx = 0; tx < nbx; ++tx) {
      const size_t t = (size_t)ty * nbx + tx;
      float a[8][8], s[8], vt[8][8];
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) a[r][c] = plane[(size_t)(ty * 8 + r) * row_stride + tx * 8 + c];
      svd_tile(a, s, vt);
      if (!(s[7] > 1e-5f * s[0])) {   // rank-deficient: orthonormal completion, like k_svd_tiles
        for (int r = 0; r < 8; ++r)
          for (int c = 0; c < 8; ++c) a[r][c] = plane[(size_t)(ty * 8 + r) * row_stride + tx * 8 + c];
        svd_tile(a, s, vt, true);
      }
      for (int i = 0; i < 8; ++i) S[t * 8 + i] = s[i];
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
          U[t * 64 + r * 8 + c] = a[r][c];
          Vt[t * 64 + r * 8 + c] = vt[r][c];
        }
    }
  return 0;
}

int hh_extract_tiles_u8(const uint8_t* stego, const float* sigma_c, const float* Uw, const float* Vwt,
                        float* out, int H, int W, int row_stride, float alpha, int K) {
  const int nby = H / 8, nbx = W / 8;
  const float inv_alpha = 1.0f / fmaxf(alpha, 1e-8f);
  float keep[8];
  for (int i = 0; i < 8; ++i) keep[i] = (i < K) ? 1.0f : 0.0f;
  for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = 0.0f;
  for (int ty = 0; ty < nby; ++ty)
    for (int tx = 0; tx < nbx; ++tx) {
      const size_t t = (size_t)ty * nbx + tx;
      float a[8][8], s[8], sc[8], uw[8][8], vwt[8][8], o[8][8];
      load_u8(stego + (size_t)ty * 8 * row_stride + tx * 8, row_stride, a);
      sigma_tile(a, s);
      for (int i = 0; i < 8; ++i) sc[i] = sigma_c[t * 8 + i];
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
          uw[r][c] = Uw[t * 64 + r * 8 + c];
          vwt[r][c] = Vwt[t * 64 + r * 8 + c];
        }
      extract_tile(s, sc, inv_alpha, keep, uw, vwt, o);
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) out[(size_t)(ty * 8 + r) * W + tx * 8 + c] = o[r][c];
    }
  return 0;
}


static inline void load_raw(const uint8_t* p, int stride, RawTile& t) {
  for (int r = 0; r < 8; ++r) {
    const uint8_t* q = p + (size_t)r * stride;
    t.lo[r] = q[0] | (q[1] << 8) | (q[2] << 16) | ((uint32_t)q[3] << 24);
    t.hi[r] = q[4] | (q[5] << 8) | (q[6] << 16) | ((uint32_t)q[7] << 24);
  }
}

// production formulation (packed, V-free, pixel domain) with the literal fallback,
// mirroring k_embed_tiles
int hh_embed_tiles_u8_pk(const uint8_t* host, const float* sigma_w, uint8_t* stego, float* sigma_c,
                         float* yw, int H, int W, int row_stride, float alpha, int K, int* max_sweeps,
                         int* n_fallback) {
  const int nby = H / 8, nbx = W / 8;
  float alpha_k[8];
  for (int i = 0; i < 8; ++i) alpha_k[i] = (i < K) ? alpha : 0.0f;
  int ms = 0, nf = 0;
  for (int r = 0; r < H; ++r) memcpy(stego + (size_t)r * row_stride, host + (size_t)r * row_stride, W);
  if (yw)
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < W; ++c) yw[(size_t)r * W + c] = (float)host[(size_t)r * row_stride + c];
  for (int ty = 0; ty < nby; ++ty)
    for (int tx = 0; tx < nbx; ++tx) {
      const size_t t = (size_t)ty * nbx + tx;
      RawTile raw, out;
      float sw[8], sc[8], y[8][8];
      load_raw(host + (size_t)ty * 8 * row_stride + tx * 8, row_stride, raw);
      for (int i = 0; i < 8; ++i) sw[i] = sigma_w[t * 8 + i];
      bool deficient = false;
      v2f ab[4][8];
      float n2[8];
      int s = embed_jacobi_pk(raw, ab, n2);                      // = embed_tile_pk, with B and the column norms kept
      embed_finish_pk<true>(raw, ab, n2, sw, alpha_k, sc, out, &y[0][0], 8, deficient);
      if (deficient) {
        ++nf;
        raw_to_f32(raw, y);
        if (raw_is_constant(raw)) { embed_tile_constant(y[0][0], sw, alpha_k, sc, y); s = 1; }   // like k_embed_fallback
        else if (raw_rank1_pretest(raw) && raw_is_rank1(raw)) { embed_tile_rank1(y, sw, alpha_k, sc, y); s = 1; }   // closed form as well
        else {                                                      // everything else that is flagged: completed from B, no V
          float bb[8][8];
          for (int r = 0; r < 8; ++r) for (int i = 0; i < 8; ++i) bb[r][i] = ab[r >> 1][i][r & 1];
          if (n2_one_small(n2)) embed_tile_one_small(y, bb, sw, alpha_k, sc, y);     // one missing pair: its sign is defined
          else embed_tile_from_b(y, bb, sw, alpha_k, sc, y);                        // ranks 2 .. 6
        }
        for (int r = 0; r < 8; ++r) {
          out.lo[r] = quant_u8(y[r][0]) | (quant_u8(y[r][1]) << 8) | (quant_u8(y[r][2]) << 16) | (quant_u8(y[r][3]) << 24);
          out.hi[r] = quant_u8(y[r][4]) | (quant_u8(y[r][5]) << 8) | (quant_u8(y[r][6]) << 16) | (quant_u8(y[r][7]) << 24);
        }
      }
      if (s > ms) ms = s;
      for (int i = 0; i < 8; ++i) sigma_c[t * 8 + i] = sc[i];
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
          const uint32_t w = (c < 4) ? out.lo[r] : out.hi[r];
          stego[(size_t)(ty * 8 + r) * row_stride + tx * 8 + c] = (uint8_t)(w >> (8 * (c & 3)));
          if (yw) yw[(size_t)(ty * 8 + r) * W + tx * 8 + c] = y[r][c];
        }
    }
  if (max_sweeps) *max_sweeps = ms;
  if (n_fallback) *n_fallback = nf;
  return 0;
}

int hh_sigma_tiles_u8_pk(const uint8_t* plane, float* sigma, int H, int W, int row_stride, int* sweep_hist) {
  const int nby = H / 8, nbx = W / 8;
  for (int ty = 0; ty < nby; ++ty)
    for (int tx = 0; tx < nbx; ++tx) {
      RawTile raw;
      float s[8];
      load_raw(plane + (size_t)ty * 8 * row_stride + tx * 8, row_stride, raw);
      int sw = sigma_tile_pk(raw, s);
      if (sweep_hist) { if (sw < 0) sw = 15; sweep_hist[sw > 15 ? 15 : sw]++; }
      for (int i = 0; i < 8; ++i) sigma[((size_t)ty * nbx + tx) * 8 + i] = s[i];
    }
  return 0;
}

// one constant tile of value v through the closed form and through the literal chain (both Yw [8][8] and Sc [8])
void hh_constant_tile_both_ways(float v, const float* sw, const float* alpha_k, float* yw_closed, float* sc_closed,
                                float* yw_literal, float* sc_literal) {
  float swa[8], ak[8], sc[8], a[8][8];
  for (int i = 0; i < 8; ++i) { swa[i] = sw[i]; ak[i] = alpha_k[i]; }
  embed_tile_constant(v, swa, ak, sc, a);
  memcpy(yw_closed, a, sizeof(a)); memcpy(sc_closed, sc, sizeof(sc));
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) a[r][c] = v;
  embed_tile_completed(a, swa, ak, sc);
  memcpy(yw_literal, a, sizeof(a)); memcpy(sc_literal, sc, sizeof(sc));
}

// one rank-1 tile (uint8 values in tile[64]) through the closed form and the literal chain; returns raw_is_rank1's verdict
int hh_rank1_tile_both_ways(const uint8_t* tile, const float* sw, const float* alpha_k, float* yw_closed, float* sc_closed,
                            float* yw_literal, float* sc_literal) {
  float swa[8], ak[8], sc[8], a[8][8], x[8][8];
  RawTile raw;
  load_raw(tile, 8, raw);
  for (int i = 0; i < 8; ++i) { swa[i] = sw[i]; ak[i] = alpha_k[i]; }
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) x[r][c] = (float)tile[r * 8 + c];
  const int is1 = (raw_rank1_pretest(raw) && raw_is_rank1(raw)) ? 1 : 0;
  if (is1) {
    embed_tile_rank1(x, swa, ak, sc, a);
    memcpy(yw_closed, a, sizeof(a)); memcpy(sc_closed, sc, sizeof(sc));
  }
  memcpy(a, x, sizeof(a));
  embed_tile_completed(a, swa, ak, sc);
  memcpy(yw_literal, a, sizeof(a)); memcpy(sc_literal, sc, sizeof(sc));
  return is1;
}

void hh_dct8x8(float* tile, int inverse) {
  float a[8][8];
  memcpy(a, tile, sizeof(a));
  if (inverse) idct8x8(a); else dct8x8(a);
  memcpy(tile, a, sizeof(a));
}

}  // extern "C"
