// Development study (host build): for every sweep of the packed one-sided Jacobi, how many column pairs
// are below a cos^2 threshold in ALL 64 tiles of a wave at the moment the pair is visited - i.e. how much a
// wave-uniform "leave this pair alone" test can save (embed-kernel diet).
//   g++ -O2 -o tools/bin/skip_study tools/skip_study.cpp && tools/bin/skip_study
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../digital-watermarking-for-image-video-using-dct-svd-singular-value-decomposition_amd/csrc/wm_tile_math.h"
using namespace wm;

int main() {
  const int NW = 300, NS = 6;
  const char* kinds[] = {"noise", "natural"};
  const float thr[] = {1e-13f, 1e-12f, 1e-11f, 1e-10f, 1e-8f, 1e-7f, 1e-6f, 1e-5f, 1e-4f, 1e-3f};
  const int NT = sizeof(thr) / sizeof(thr[0]);
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
    // below[s][k]: pairs whose wave-max cos^2 < thr[k] in sweep s; wavemax[s][k]: waves whose sweep-s maximum over all pairs < thr[k]
    double below[NS][16] = {{0}}, wavemax[NS][16] = {{0}}, lanebelow[NS][16] = {{0}};
    for (int w = 0; w < NW; ++w) {
      static v2f a[64][4][8]; static float n2[64][8];
      for (int l = 0; l < 64; ++l) {
        for (int rp = 0; rp < 4; ++rp) for (int c = 0; c < 8; ++c) {
          v2f v = {(float)px[((size_t)w * 64 + l) * 64 + (2 * rp) * 8 + c], (float)px[((size_t)w * 64 + l) * 64 + (2 * rp + 1) * 8 + c]};
          a[l][rp][c] = v;
        }
        col_norms2_pk(a[l], n2[l]);
      }
      for (int s = 0; s < NS; ++s) {
        float sweepmax = 0;
        float lanemax[64] = {0};
        if (s >= 2 && (s & 1) == 0) for (int l = 0; l < 64; ++l) col_norms2_pk(a[l], n2[l]);
        for (int p = 0; p < 7; ++p) for (int q = p + 1; q < 8; ++q) {
          float m = 0;
          for (int l = 0; l < 64; ++l) {
            v2f gv = a[l][0][p] * a[l][0][q];
            for (int rp = 1; rp < 4; ++rp) gv = fma2(a[l][rp][p], a[l][rp][q], gv);
            const float g = gv[0] + gv[1];
            const float c2 = g * g / fmaxf(n2[l][p] * n2[l][q], 1e-30f);
            m = fmaxf(m, c2); lanemax[l] = fmaxf(lanemax[l], c2);
            bool dummy = false;
            jacobi_rot_pk<0>(a[l], n2[l], p, q, dummy);
          }
          sweepmax = fmaxf(sweepmax, m);
          for (int k = 0; k < NT; ++k) if (m < thr[k]) below[s][k] += 1;
        }
        for (int k = 0; k < NT; ++k) {
          if (sweepmax < thr[k]) wavemax[s][k] += 1;
          for (int l = 0; l < 64; ++l) if (lanemax[l] < thr[k]) lanebelow[s][k] += 1;
        }
      }
    }
    printf("== %s: fraction of (wave, pair) visits whose cos^2 is below T in all 64 tiles | fraction of WAVES whose whole sweep is below T | fraction of TILES\n", kinds[kind]);
    printf("   T:      "); for (int k = 0; k < NT; ++k) printf(" %7.0e", thr[k]); printf("\n");
    for (int s = 2; s < NS; ++s) {
      printf("sweep %d pair:", s + 1); for (int k = 0; k < NT; ++k) printf(" %7.3f", below[s][k] / (NW * 28.0)); printf("\n");
      printf("        wave:"); for (int k = 0; k < NT; ++k) printf(" %7.3f", wavemax[s][k] / NW); printf("\n");
      printf("        tile:"); for (int k = 0; k < NT; ++k) printf(" %7.3f", lanebelow[s][k] / (NW * 64.0)); printf("\n");
    }
  }
  return 0;
}
