// Development study (host build): one-sided Jacobi with SCALED ("fast") rotations and no
// de Rijk swap, against the production packed sweep, on 8x8 uint8 tiles with the kernels'
// wave-uniform termination emulated.  Reports sweeps, sigma error vs float64, final max cos.
//   g++ -O2 -o tools/bin/fastrot_study tools/fastrot_study.cpp && tools/bin/fastrot_study
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd/csrc/wm_tile_math.h"
using namespace wm;

struct Tile { float s[8][8]; float AL[8]; float D[8]; };

static inline float rsqf(float x) { return 1.0f / sqrtf(x); }

// one scaled rotation of stored columns p,q; returns cos^2 seen
static float fast_rot(Tile& t, int p, int q) {
  float g = 0;
  for (int r = 0; r < 8; ++r) g = fmaf(t.s[r][p], t.s[r][q], g);
  const float Dp = t.D[p], Dq = t.D[q], al = t.AL[p], be = t.AL[q];
  const float g2 = g + g, u2 = g2 * Dp, v2 = g2 * Dq, G4 = u2 * v2;
  const float tau = be - al, ta = fabsf(tau) + 1e-18f;
  const float h2 = fmaf(ta, ta, G4);
  const float ih = rsqf(h2), h = h2 * ih;
  float r = 1.0f / (ta + h);
  r = (tau < 0.0f) ? -r : r;
  const float tp = v2 * r, tq = u2 * r;
  const float c2 = fmaf(0.5f * ta, ih, 0.5f);
  t.D[p] = Dp * c2; t.D[q] = Dq * c2;
  const float w = 0.5f * G4 * r;                 // t * g_true, sign of tau
  t.AL[p] = al - w; t.AL[q] = be + w;
  for (int rr = 0; rr < 8; ++rr) {
    const float X = t.s[rr][p], Y = t.s[rr][q];
    t.s[rr][p] = fmaf(-tp, Y, X);
    t.s[rr][q] = fmaf(tq, X, Y);
  }
  return 0.25f * G4 / fmaxf(al * be, 1e-30f);
}

static void sort_cols(Tile& t) {   // physical sort of stored columns by true norm, descending
  for (int i = 0; i < 8; ++i) for (int j = i + 1; j < 8; ++j) if (t.AL[j] > t.AL[i]) {
    std::swap(t.AL[i], t.AL[j]); std::swap(t.D[i], t.D[j]);
    for (int r = 0; r < 8; ++r) std::swap(t.s[r][i], t.s[r][j]);
  }
}
static int SORT_UNTIL = 0;
static void true_norms(Tile& t) {
  for (int c = 0; c < 8; ++c) { float n = 0; for (int r = 0; r < 8; ++r) n = fmaf(t.s[r][c], t.s[r][c], n); t.AL[c] = n * t.D[c]; }
}

static void svd_f64(const double (&x)[8][8], double (&s)[8]) {
  double a[8][8];
  for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) a[r][c] = x[r][c];
  for (int sw = 0; sw < 30; ++sw)
    for (int p = 0; p < 7; ++p)
      for (int q = p + 1; q < 8; ++q) {
        double al = 0, be = 0, g = 0;
        for (int r = 0; r < 8; ++r) { al += a[r][p] * a[r][p]; be += a[r][q] * a[r][q]; g += a[r][p] * a[r][q]; }
        if (fabs(g) < 1e-300) continue;
        const double z = (be - al) / (2 * g), tt = (z >= 0 ? 1 : -1) / (fabs(z) + sqrt(1 + z * z));
        const double c = 1 / sqrt(1 + tt * tt), s_ = c * tt;
        for (int r = 0; r < 8; ++r) { const double X = a[r][p], Y = a[r][q]; a[r][p] = c * X - s_ * Y; a[r][q] = s_ * X + c * Y; }
      }
  for (int c = 0; c < 8; ++c) { double n = 0; for (int r = 0; r < 8; ++r) n += a[r][c] * a[r][c]; s[c] = sqrt(n); }
  std::sort(s, s + 8, [](double u, double v) { return u > v; });
}

int main(int argc, char** argv) {
  if (argc > 1) SORT_UNTIL = atoi(argv[1]);
  const int NW = 400;
  const char* kinds[] = {"noise", "natural"};
  const float thr[] = {1e-7f, 1e-3f};
  for (int kind = 0; kind < 2; ++kind) {
    srand(1234);
    std::vector<uint8_t> px((size_t)NW * 64 * 64);
    for (int t = 0; t < NW * 64; ++t) {
      const double b0 = 20 + rand() % 200, gx = (rand() % 200 - 100) / 25.0, gy = (rand() % 200 - 100) / 25.0, cxy = (rand() % 200 - 100) / 400.0;
      for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) {
        double v;
        if (kind == 0) v = rand() % 256;
        else { double n = 0; for (int k = 0; k < 4; ++k) n += (rand() % 1000) / 1000.0 - 0.5; v = b0 + gx * c + gy * r + cxy * r * c + 3.5 * n; }
        px[(size_t)t * 64 + r * 8 + c] = (uint8_t)fmin(fmax(v, 0.0), 255.0);
      }
    }
    for (float T : thr) {
      double sum_sweeps = 0, max_err = 0, sum_err = 0, max_cos = 0, max_rel = 0; long nerr = 0; int hist[16] = {0};
      double growth = 1;
      for (int w = 0; w < NW; ++w) {
        static Tile tl[64];
        for (int l = 0; l < 64; ++l) {
          for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) tl[l].s[r][c] = px[((size_t)w * 64 + l) * 64 + r * 8 + c];
          for (int c = 0; c < 8; ++c) tl[l].D[c] = 1.0f;
          true_norms(tl[l]);
        }
        int sweep = 0; bool more = true;
        while (more && sweep < 12) {
          more = false;
          for (int l = 0; l < 64; ++l) {
            if (sweep >= 2 && (sweep & 1) == 0) true_norms(tl[l]);
            if (sweep < SORT_UNTIL) { true_norms(tl[l]); sort_cols(tl[l]); }
            float m = 0;
            for (int p = 0; p < 7; ++p) for (int q = p + 1; q < 8; ++q) m = fmaxf(m, fast_rot(tl[l], p, q));
            if (sweep >= 2 && m > T) more = true;
          }
          if (sweep < 2) more = true;
          ++sweep;
        }
        sum_sweeps += sweep; hist[sweep]++;
        for (int l = 0; l < 64; ++l) {
          true_norms(tl[l]);
          double x[8][8], s[8], mine[8];
          for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) x[r][c] = px[((size_t)w * 64 + l) * 64 + r * 8 + c];
          svd_f64(x, s);
          for (int i = 0; i < 8; ++i) { mine[i] = sqrt((double)tl[l].AL[i]); growth = std::max(growth, 1.0 / tl[l].D[i]); }
          std::sort(mine, mine + 8, [](double u, double v) { return u > v; });
          for (int i = 0; i < 8; ++i) {
            const double e = fabs(mine[i] - s[i]) / s[0]; max_err = std::max(max_err, e); sum_err += e; ++nerr;
            if (s[i] > 1e-3 * s[0]) max_rel = std::max(max_rel, fabs(mine[i] - s[i]) / s[i]);
          }
          for (int p = 0; p < 7; ++p) for (int q = p + 1; q < 8; ++q) {
            double g = 0, np = 0, nq = 0;
            for (int r = 0; r < 8; ++r) { g += (double)tl[l].s[r][p] * tl[l].s[r][q]; np += (double)tl[l].s[r][p] * tl[l].s[r][p]; nq += (double)tl[l].s[r][q] * tl[l].s[r][q]; }
            if (np > 0 && nq > 0) max_cos = std::max(max_cos, fabs(g) / sqrt(np * nq));
          }
        }
      }
      printf("fastrot %-8s T=%.0e  wave sweeps avg %.3f [3:%d 4:%d 5:%d 6:%d 7+:%d]  sigma err/s1 max %.2e mean %.2e  rel(s_i>1e-3 s1) max %.2e  final max cos %.2e  max 1/D %.1f\n",
             kinds[kind], T, sum_sweeps / NW, hist[3], hist[4], hist[5], hist[6], hist[7] + hist[8] + hist[9] + hist[10] + hist[11] + hist[12],
             max_err, sum_err / nerr, max_rel, max_cos, growth);
    }
  }
  return 0;
}
