/* spec_chain.c -- zero-GPU feasibility study of the SPECULATIVE SEGMENT-PARALLEL ordered chain (VERDICT round 2, next #2c).
 *
 * Mixer::Mix's dot product is p = (((0 + x0) + x1) + ... + x2077) in f32, products x_i = RN(in_i * w_i) (mixer.cpp:40-52). With the
 * chain cut into S segments, segment k needs s_k, the f32 running sum at its first term. Idea: a spare wavefront runs segment k from 64
 * candidate starts at once; s -> RN(s + x) is monotone non-decreasing, so is the whole segment map F_k; when the true s_k arrives,
 *   - a lane whose candidate equals s_k bit for bit has the exact result, and
 *   - two candidates c_j <= s_k <= c_{j+1} with F_k(c_j) == F_k(c_{j+1}) bracket it: F_k(s_k) is that common value, exactly.
 * A miss falls back to the serial path. The estimate the candidates are centred on is the f64 sum of the products before the
 * segment, rounded to f32 (a tree sum, available early). This module is fed every layer-0 Mix of an oracle run through the
 * orc_mix_probe hook and counts, per candidate scheme, how often each speculative segment would hit.
 *
 * Candidate schemes (64 lanes each):
 *   0: consecutive floats around the estimate (est -32 ulp .. est +31 ulp)
 *   1: est + j * d, j = -32..31, d = ulp(max |prefix sum| seen in f64 over the preceding terms)          (absolute grid, coarse)
 *   2: est + j * d/2                                                                                     (same grid, half step)
 *   3: est + j * d/4
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NSCHEME 4
#define MAXSEG 16
static int g_nseg = 4;
static uint64_t g_calls, g_hit[NSCHEME][MAXSEG], g_hit_exact[NSCHEME][MAXSEG], g_tot[MAXSEG], g_allhit[NSCHEME], g_wrong;
static uint64_t g_err_hist[64];   /* log2 bucket of |s_k - est| in ulp(est) */
static double g_err_ulp_max_sum, g_err_ulp_max_sq; static uint64_t g_err_n;
static uint64_t g_sample_every = 1, g_seen;

static float ulpf(float x) { x = fabsf(x); if (x < 1e-30f) x = 1e-30f; return nextafterf(x, INFINITY) - x; }

static float seg_run(float s, const float* x, int n) { for (int i = 0; i < n; ++i) s += x[i]; return s; }

void spec_probe(const void* mixer, const float* in, const float* w, int n_in) {
  (void)mixer;
  if (n_in != 2078) return;
  if ((g_seen++ % g_sample_every) != 0) return;
  static float x[2078];
  for (int i = 0; i < n_in; ++i) { volatile float t = in[i] * w[i]; x[i] = t; }
  const int S = g_nseg;
  int edge[MAXSEG + 1];
  for (int k = 0; k <= S; ++k) edge[k] = (int)((long)n_in * k / S);
  /* true starts and f64 prefix data */
  float s = 0; double d = 0, dmax = 0;
  int all[NSCHEME]; for (int c = 0; c < NSCHEME; ++c) all[c] = 1;
  g_calls++;
  for (int k = 0; k < S; ++k) {
    if (k > 0) {
      const float est = (float)d;
      const float strue = s;
      const float want = seg_run(strue, x + edge[k], edge[k + 1] - edge[k]);
      /* error statistics */
      double e = fabs((double)strue - (double)est) / ulpf(est);
      int b = e < 0.5 ? 0 : (int)ceil(log2(e + 1)); if (b > 63) b = 63;
      g_err_hist[b]++;
      double em = fabs((double)strue - d) / ulpf((float)dmax); g_err_ulp_max_sum += em; g_err_ulp_max_sq += em * em; g_err_n++;
      g_tot[k]++;
      for (int sch = 0; sch < NSCHEME; ++sch) {
        float c[64], f[64];
        if (sch == 0) {
          c[32] = est;
          for (int j = 33; j < 64; ++j) c[j] = nextafterf(c[j - 1], INFINITY);
          for (int j = 31; j >= 0; --j) c[j] = nextafterf(c[j + 1], -INFINITY);
        } else {
          const float step = ulpf((float)dmax) / (float)(1 << (sch - 1));
          for (int j = 0; j < 64; ++j) c[j] = est + (float)(j - 32) * step;
        }
        for (int j = 0; j < 64; ++j) f[j] = c[j];
        for (int i = edge[k]; i < edge[k + 1]; ++i) { const float xi = x[i]; for (int j = 0; j < 64; ++j) f[j] += xi; }
        int hit = 0, exact = 0; float got = 0;
        for (int j = 0; j < 64 && !hit; ++j) if (c[j] == strue) { hit = exact = 1; got = f[j]; }
        for (int j = 0; j + 1 < 64 && !hit; ++j) if (c[j] <= strue && strue <= c[j + 1] && f[j] == f[j + 1]) { hit = 1; got = f[j]; }
        if (hit) { g_hit[sch][k]++; g_hit_exact[sch][k] += exact; if (memcmp(&got, &want, 4)) g_wrong++; }
        else all[sch] = 0;
      }
    }
    for (int i = edge[k]; i < edge[k + 1]; ++i) { s += x[i]; d += (double)x[i]; if (fabs(d) > dmax) dmax = fabs(d); }
  }
  for (int c = 0; c < NSCHEME; ++c) g_allhit[c] += all[c];
}

void spec_config(int nseg, int sample_every) { g_nseg = nseg; g_sample_every = sample_every > 0 ? sample_every : 1; }
void spec_reset(void) { g_calls = g_seen = g_wrong = 0; memset(g_hit, 0, sizeof g_hit); memset(g_hit_exact, 0, sizeof g_hit_exact); memset(g_tot, 0, sizeof g_tot);
  memset(g_allhit, 0, sizeof g_allhit); memset(g_err_hist, 0, sizeof g_err_hist); g_err_ulp_max_sum = g_err_ulp_max_sq = 0; g_err_n = 0; }
void spec_report(FILE* f_unused) {
  (void)f_unused;
  printf("segments %d, mixes sampled %llu, wrong results among hits: %llu (must be 0: monotonicity)\n", g_nseg, (unsigned long long)g_calls, (unsigned long long)g_wrong);
  const char* nm[NSCHEME] = {"consecutive floats around est", "est + j*ulp(max|prefix|)", "est + j*ulp(max|prefix|)/2", "est + j*ulp(max|prefix|)/4"};
  for (int sch = 0; sch < NSCHEME; ++sch) {
    printf("scheme %d (%s):", sch, nm[sch]);
    for (int k = 1; k < g_nseg; ++k) printf("  seg%d %.4f (exact %.4f)", k, (double)g_hit[sch][k] / (double)g_tot[k], (double)g_hit_exact[sch][k] / (double)g_tot[k]);
    printf("  | all %d speculative segments of a mix hit: %.4f\n", g_nseg - 1, (double)g_allhit[sch] / (double)g_calls);
  }
  printf("|true start - f64 estimate| in ulp(estimate), log2 buckets (0: <0.5, b: < 2^b):");
  for (int b = 0; b < 24; ++b) printf(" %llu", (unsigned long long)g_err_hist[b]);
  printf("\nsame in ulp(max |prefix|): mean %.2f rms %.2f\n", g_err_ulp_max_sum / (double)g_err_n, sqrt(g_err_ulp_max_sq / (double)g_err_n));
}
