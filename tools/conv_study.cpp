// Development study (host build): how the Jacobi stopping threshold trades sweeps for
// accuracy on 8x8 uint8 tiles, with the wave-uniform termination of the kernels emulated
// (64 consecutive tiles sweep in lock-step until none of them saw cos^2 > T).
//   g++ -O2 -o tools/bin/conv_study tools/conv_study.cpp && tools/bin/conv_study
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd/csrc/wm_tile_math.h"
using namespace wm;

// jacobi_rot_pk with the largest cos^2 it saw reported instead of a fixed threshold test
static void rot(v2f (&a)[4][8], float (&n2)[8], int p, int q, float& maxc2) {
  v2f gv = a[0][p] * a[0][q];
  for (int rp = 1; rp < 4; ++rp) gv = fma2(a[rp][p], a[rp][q], gv);
  const float g = gv[0] + gv[1];
  const float al = n2[p], be = n2[q];
  const float c2 = g * g / fmaxf(al * be, 1e-30f);
  if (c2 > maxc2) maxc2 = c2;
  bool dummy = false;
  jacobi_rot_pk<0>(a, n2, p, q, dummy);
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
        const double z = (be - al) / (2 * g), t = (z >= 0 ? 1 : -1) / (fabs(z) + sqrt(1 + z * z));
        const double c = 1 / sqrt(1 + t * t), s_ = c * t;
        for (int r = 0; r < 8; ++r) { const double X = a[r][p], Y = a[r][q]; a[r][p] = c * X - s_ * Y; a[r][q] = s_ * X + c * Y; }
      }
  for (int c = 0; c < 8; ++c) { double n = 0; for (int r = 0; r < 8; ++r) n += a[r][c] * a[r][c]; s[c] = sqrt(n); }
  std::sort(s, s + 8, [](double u, double v) { return u > v; });
}

int main(int argc, char** argv) {
  const int NW = 400;   // waves of 64 tiles
  const char* kinds[] = {"noise", "natural"};
  const float thr[] = {1e-7f, 1e-6f, 1e-5f, 1e-4f, 1e-3f};
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
      double sum_sweeps = 0, max_err = 0, sum_err = 0, max_cos = 0, sum_tile_sweeps = 0; long nerr = 0;
      int hist[16] = {0};
      for (int w = 0; w < NW; ++w) {
        static v2f a[64][4][8]; static float n2[64][8];
        int tile_done[64];
        for (int l = 0; l < 64; ++l) {
          for (int rp = 0; rp < 4; ++rp) for (int c = 0; c < 8; ++c) {
            v2f v = {(float)px[((size_t)w * 64 + l) * 64 + (2 * rp) * 8 + c], (float)px[((size_t)w * 64 + l) * 64 + (2 * rp + 1) * 8 + c]};
            a[l][rp][c] = v;
          }
          col_norms2_pk(a[l], n2[l]); tile_done[l] = 0;
          if (argc > 1 && atoi(argv[1]) == 1) {      // experiment: columns sorted by norm (descending) before the first sweep
            int ord[8]; for (int c = 0; c < 8; ++c) ord[c] = c;
            std::sort(ord, ord + 8, [&](int u, int v) { return n2[l][u] > n2[l][v]; });
            v2f b[4][8]; float m2[8];
            for (int c = 0; c < 8; ++c) { for (int rp = 0; rp < 4; ++rp) b[rp][c] = a[l][rp][ord[c]]; m2[c] = n2[l][ord[c]]; }
            for (int c = 0; c < 8; ++c) { for (int rp = 0; rp < 4; ++rp) a[l][rp][c] = b[rp][c]; n2[l][c] = m2[c]; }
          }
          if (argc > 1 && atoi(argv[1]) == 2) {      // experiment: the ROWS' mean removed first?  (no: changes the matrix) - transpose instead
            v2f b[4][8];
            float t_[8][8];
            for (int rp = 0; rp < 4; ++rp) for (int c = 0; c < 8; ++c) { t_[2 * rp][c] = a[l][rp][c][0]; t_[2 * rp + 1][c] = a[l][rp][c][1]; }
            for (int rp = 0; rp < 4; ++rp) for (int c = 0; c < 8; ++c) { v2f v = {t_[c][2 * rp], t_[c][2 * rp + 1]}; b[rp][c] = v; }
            for (int rp = 0; rp < 4; ++rp) for (int c = 0; c < 8; ++c) a[l][rp][c] = b[rp][c];
            col_norms2_pk(a[l], n2[l]);
          }
        }
        int sweep = 0; bool more = true;
        while (more && sweep < 12) {
          more = false;
          for (int l = 0; l < 64; ++l) {
            if (sweep >= 2 && (sweep & 1) == 0) col_norms2_pk(a[l], n2[l]);
            float m = 0;
            for (int p = 0; p < 7; ++p) for (int q = p + 1; q < 8; ++q) rot(a[l], n2[l], p, q, m);
            if (sweep >= 2 && m > T) more = true;
            if (!(m > T) && !tile_done[l]) tile_done[l] = sweep + 1;
            if (m > T) tile_done[l] = 0;
          }
          if (sweep < 2) more = true;
          ++sweep;
        }
        sum_sweeps += sweep; hist[sweep]++;
        for (int l = 0; l < 64; ++l) {
          sum_tile_sweeps += tile_done[l] ? std::max(tile_done[l], 3) : sweep;
          col_norms2_pk(a[l], n2[l]);
          double x[8][8], s[8];
          for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) x[r][c] = px[((size_t)w * 64 + l) * 64 + r * 8 + c];
          svd_f64(x, s);
          for (int i = 0; i < 8; ++i) { const double e = fabs(sqrt((double)n2[l][i]) - s[i]) / s[0]; max_err = std::max(max_err, e); sum_err += e; ++nerr; }
          for (int p = 0; p < 7; ++p) for (int q = p + 1; q < 8; ++q) {
            double g = 0; for (int rp = 0; rp < 4; ++rp) g += (double)a[l][rp][p][0] * a[l][rp][q][0] + (double)a[l][rp][p][1] * a[l][rp][q][1];
            max_cos = std::max(max_cos, fabs(g) / sqrt((double)n2[l][p] * n2[l][q] + 1e-300));
          }
        }
      }
      printf("%-8s T=%.0e  wave sweeps avg %.3f [3:%d 4:%d 5:%d 6:%d 7+:%d]  per-tile avg %.3f  sigma err/s1 max %.2e mean %.2e  final max cos %.2e\n",
             kinds[kind], T, sum_sweeps / NW, hist[3], hist[4], hist[5], hist[6], hist[7] + hist[8] + hist[9] + hist[10] + hist[11] + hist[12],
             sum_tile_sweeps / (NW * 64.0), max_err, sum_err / nerr, max_cos);
    }
  }
  return 0;
}
