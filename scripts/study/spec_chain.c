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
 *   4..7: a dense core around the estimate (consecutive floats, |k| <= D) and geometrically growing steps behind it, in ulp(est): the far
 *         candidates only ever hit through the bracket rule (spec_offset below). Round 4: what the helpers' waves could use instead of scheme 0.
 * Also counted per BIT (26 consecutive layer-0 mixes = the 26 helper workgroups of one bit, which the gather wave waits for together): how
 * often at least one of the 78 speculative segments misses -- a miss re-runs a 512-term chain while the other 25 helpers idle.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NSCHEME 8
#define MAXSEG 16
static int g_nseg = 4;
static uint64_t g_calls, g_hit[NSCHEME][MAXSEG], g_hit_exact[NSCHEME][MAXSEG], g_tot[MAXSEG], g_allhit[NSCHEME], g_wrong;
static uint64_t g_err_hist[64];   /* log2 bucket of |s_k - est| in ulp(est) */
static double g_err_ulp_max_sum, g_err_ulp_max_sq; static uint64_t g_err_n;
static uint64_t g_sample_every = 1, g_seen;

/* schemes 4..7: candidate k = -32..31 sits off(k) ulps from the estimate; dense for |k| <= D, then D + G * (2^(|k| - D) - 1) */
static const int g_core[NSCHEME] = {0, 0, 0, 0, 16, 24, 20, 12};
static const int g_grow[NSCHEME] = {0, 0, 0, 0, 2, 4, 1, 1};
static long spec_offset(int sch, int k) {
  const int a = k < 0 ? -k : k, D = g_core[sch];
  long o = a <= D ? a : D + (long)g_grow[sch] * ((1L << (a - D)) - 1);
  return k < 0 ? -o : o;
}
static uint32_t f2ord(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static float ord2f(uint32_t o) { uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; memcpy(&f, &u, 4); return f; }
static uint64_t g_wrong_s[NSCHEME];
static uint64_t g_bit_calls, g_bit_anymiss[NSCHEME], g_bit_misses[NSCHEME], g_in_bit, g_bit_miss_now[NSCHEME];

static float ulpf(float x) { x = fabsf(x); if (x < 1e-30f) x = 1e-30f; return nextafterf(x, INFINITY) - x; }

static float seg_run(float s, const float* x, int n) { for (int i = 0; i < n; ++i) s += x[i]; return s; }

void spec_probe(const void* mixer, const float* in, const float* w, int n_in) {
  (void)mixer;
  if (n_in != 2078) return;
  if (((g_seen++ / 26) % g_sample_every) != 0) return;   /* whole bits are sampled: the 26 mixers of a bit together */
  static float x[2078];
  for (int i = 0; i < n_in; ++i) { volatile float t = in[i] * w[i]; x[i] = t; }
  const int S = g_nseg;
  int edge[MAXSEG + 1];
  for (int k = 0; k <= S; ++k) edge[k] = (int)((long)n_in * k / S);
  if (S == 4) { edge[1] = 512; edge[2] = 1024; edge[3] = 1536; }
  if (S == 8) for (int k = 1; k < 8; ++k) edge[k] = 256 * k;   /* as the helper workgroups cut the chain (mixnet_chunk.hip helper_role) */
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
        } else if (sch < 4) {
          const float step = ulpf((float)dmax) / (float)(1 << (sch - 1));
          for (int j = 0; j < 64; ++j) c[j] = est + (float)(j - 32) * step;
        } else {
          for (int j = 0; j < 64; ++j) c[j] = ord2f((uint32_t)((long)f2ord(est) + spec_offset(sch, j - 32)));
        }
        for (int j = 0; j < 64; ++j) f[j] = c[j];
        for (int i = edge[k]; i < edge[k + 1]; ++i) { const float xi = x[i]; for (int j = 0; j < 64; ++j) f[j] += xi; }
        int hit = 0, exact = 0; float got = 0;
        if (sch < 4) {
          for (int j = 0; j < 64 && !hit; ++j) if (c[j] == strue) { hit = exact = 1; got = f[j]; }
          for (int j = 0; j + 1 < 64 && !hit; ++j) if (c[j] <= strue && strue <= c[j + 1] && f[j] == f[j + 1]) { hit = 1; got = f[j]; }
        } else {   /* as a wave would do it: everything in the ordered-integer image of the floats (-0 and +0 are two candidates), results compared bit for bit */
          const uint32_t so = f2ord(strue);
          int below = 0;
          for (int j = 0; j < 64; ++j) below += f2ord(c[j]) <= so;
          if (below >= 1 && f2ord(c[below - 1]) == so) { hit = exact = 1; got = f[below - 1]; }
          else if (below >= 1 && below < 64 && !memcmp(&f[below - 1], &f[below], 4)) { hit = 1; got = f[below - 1]; }
        }
        if (hit) { g_hit[sch][k]++; g_hit_exact[sch][k] += exact; if (memcmp(&got, &want, 4)) { g_wrong++; g_wrong_s[sch]++; } }
        else { all[sch] = 0; g_bit_miss_now[sch]++; }
      }
    }
    for (int i = edge[k]; i < edge[k + 1]; ++i) { s += x[i]; d += (double)x[i]; if (fabs(d) > dmax) dmax = fabs(d); }
  }
  for (int c = 0; c < NSCHEME; ++c) g_allhit[c] += all[c];
  if (++g_in_bit == 26) {   /* the 26 layer-0 mixers of one bit (predictor.cpp:395-400) */
    g_in_bit = 0; g_bit_calls++;
    for (int c = 0; c < NSCHEME; ++c) { g_bit_anymiss[c] += g_bit_miss_now[c] != 0; g_bit_misses[c] += g_bit_miss_now[c]; g_bit_miss_now[c] = 0; }
  }
}

void spec_config(int nseg, int sample_every) { g_nseg = nseg; g_sample_every = sample_every > 0 ? sample_every : 1; }
void spec_reset(void) { g_calls = g_seen = g_wrong = 0; memset(g_hit, 0, sizeof g_hit); memset(g_hit_exact, 0, sizeof g_hit_exact); memset(g_tot, 0, sizeof g_tot);
  memset(g_allhit, 0, sizeof g_allhit); g_bit_calls = g_in_bit = 0; memset(g_bit_anymiss, 0, sizeof g_bit_anymiss); memset(g_bit_misses, 0, sizeof g_bit_misses); memset(g_bit_miss_now, 0, sizeof g_bit_miss_now); memset(g_err_hist, 0, sizeof g_err_hist); g_err_ulp_max_sum = g_err_ulp_max_sq = 0; g_err_n = 0; }
void spec_report(FILE* f_unused) {
  (void)f_unused;
  printf("segments %d, mixes sampled %llu, wrong results among hits: %llu (must be 0: monotonicity)\n", g_nseg, (unsigned long long)g_calls, (unsigned long long)g_wrong);
  const char* nm[NSCHEME] = {"consecutive floats around est", "est + j*ulp(max|prefix|)", "est + j*ulp(max|prefix|)/2", "est + j*ulp(max|prefix|)/4",
                             "core 16 + 2(2^m-1) ulp", "core 24 + 4(2^m-1) ulp", "core 20 + (2^m-1) ulp", "core 12 + (2^m-1) ulp"};
  for (int sch = 0; sch < NSCHEME; ++sch) {
    printf("scheme %d (%s):", sch, nm[sch]);
    for (int k = 1; k < g_nseg; ++k) printf("  seg%d %.4f (exact %.4f)", k, (double)g_hit[sch][k] / (double)g_tot[k], (double)g_hit_exact[sch][k] / (double)g_tot[k]);
    printf("  | all %d speculative segments of a mix hit: %.4f | bits with a miss in any of the 26 mixers: %.4f (%.2f misses per bit)\n", g_nseg - 1, (double)g_allhit[sch] / (double)g_calls,
           (double)g_bit_anymiss[sch] / (double)(g_bit_calls ? g_bit_calls : 1), (double)g_bit_misses[sch] / (double)(g_bit_calls ? g_bit_calls : 1));
  }
  for (int c = 0; c < NSCHEME; ++c) printf("wrong[%d] %llu ", c, (unsigned long long)g_wrong_s[c]);
  printf("\n");
  printf("|true start - f64 estimate| in ulp(estimate), log2 buckets (0: <0.5, b: < 2^b):");
  for (int b = 0; b < 24; ++b) printf(" %llu", (unsigned long long)g_err_hist[b]);
  printf("\nsame in ulp(max |prefix|): mean %.2f rms %.2f\n", g_err_ulp_max_sum / (double)g_err_n, sqrt(g_err_ulp_max_sq / (double)g_err_n));
}
