/* ORACLE — test infrastructure only. Never linked, loaded or called by the product path (neuralsampleid_amd/).
 *
 * Plain-C restatement (gcc, no BLAS, no torch) of the arithmetic of the hot-path kernels K1-K6 of SURVEY.md section 8, behind a C ABI
 * whose argument conventions are those of include/nsid.h (node-major rows, clip-local int32 neighbour ids, sizes as ints, 0 on
 * success) — the "CPU restatement behind the same signatures" SURVEY section 8b asks for. It exists beside oracle/ref_torch.py (the
 * torch restatement the GPU tests use) as an INDEPENDENT second statement: ref_torch.py leans on torch's own kernels (bmm, topk,
 * logsumexp, conv arithmetic), this file spells every loop out. Parity status: PINNED — tests/test_oracle_c.py checks every function
 * here against the golden vectors produced by the reference's own modules (tests/golden/make_golden.py) and against ref_torch.py.
 *
 * Only tests/ may load the library this file builds (oracle/_build/libnsid_oracle.so; recipe: oracle/c/Makefile, driven by
 * __graft_entry__.build()).
 *
 * Reference sites restated (paths relative to the reference repo):
 *   oracle_knn_graph        encoder/gcn_lib/torch_edge.py:7-18 (pairwise_distance), 70-103 (dense_knn_matrix), 245-255 ([::dilation]),
 *                           270-284 (F.normalize in front)
 *   oracle_mr_aggregate_*   encoder/gcn_lib/torch_vertex.py:21-32 + torch_nn.py:79-98 (batched_index_select)
 *   oracle_linear_fwd       every Conv2d 1x1 / Linear (torch_vertex.py:152-162, graph_encoder.py:74-77; groups: torch_nn.py:56)
 *   oracle_bn_fwd           nn.BatchNorm2d over (B, C, N, 1) = over the rows of (B*N, C)
 *   oracle_downsample3_fwd  encoder/graph_encoder.py:44-50 (Conv2d 3x3 s2 p1 on a width-1 map)
 *   oracle_peak_patchify    peak_extractor.py:45-70
 *   oracle_ntxent           simclr/ntxent.py:5-30
 *
 * Accumulations run in double and are rounded to float once: the reference's fp32 BLAS / vectorised reductions have no fixed
 * summation order to copy, and a double sum is within half an ulp of whatever order they took.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_EINVAL (-1)

/* K1. y: (B, N, C) node-major fp32. idx: (B, N, k) clip-local ids, ascending distance (ties: lower index first, as a stable
 * arg-sort; torch.topk's own tie order is unspecified, so tests compare neighbour SETS outside recorded near-ties). */
int oracle_knn_graph(const float* y, int B, int N, int C, int k, int dilation, int32_t* idx) {
  if (!y || !idx || B <= 0 || N <= 0 || C <= 0 || k <= 0 || dilation <= 0 || (long)k * dilation > N) return ORACLE_EINVAL;
  const int kd = k * dilation;
  float* yn = (float*)malloc(sizeof(float) * (size_t)N * C);
  float* sq = (float*)malloc(sizeof(float) * (size_t)N);
  float* d = (float*)malloc(sizeof(float) * (size_t)N);
  int32_t* best = (int32_t*)malloc(sizeof(int32_t) * (size_t)kd);
  if (!yn || !sq || !d || !best) { free(yn); free(sq); free(d); free(best); return ORACLE_EINVAL; }
  for (int b = 0; b < B; ++b) {
    const float* yb = y + (size_t)b * N * C;
    for (int n = 0; n < N; ++n) {               /* F.normalize(p=2, dim=channels, eps=1e-12) */
      double s = 0.0;
      for (int c = 0; c < C; ++c) s += (double)yb[(size_t)n * C + c] * yb[(size_t)n * C + c];
      float nrm = (float)sqrt(s);
      if (nrm < 1e-12f) nrm = 1e-12f;
      double s2 = 0.0;
      for (int c = 0; c < C; ++c) {
        const float v = yb[(size_t)n * C + c] / nrm;
        yn[(size_t)n * C + c] = v;
        s2 += (double)v * v;
      }
      sq[n] = (float)s2;                        /* x_square = sum(x*x) */
    }
    for (int i = 0; i < N; ++i) {
      for (int j = 0; j < N; ++j) {             /* dist = x_square + (-2 x x^T) + x_square^T, each step rounded to fp32 */
        double dot = 0.0;
        for (int c = 0; c < C; ++c) dot += (double)yn[(size_t)i * C + c] * yn[(size_t)j * C + c];
        const float inner = -2.0f * (float)dot;
        d[j] = (sq[i] + inner) + sq[j];
      }
      for (int p = 0; p < kd; ++p) {            /* kd smallest by selection: (distance, index) lexicographic */
        int arg = -1;
        for (int j = 0; j < N; ++j) {
          int taken = 0;
          for (int q = 0; q < p; ++q) taken |= best[q] == j;
          if (taken) continue;
          if (arg < 0 || d[j] < d[arg]) arg = j;
        }
        best[p] = arg;
      }
      for (int p = 0; p < k; ++p) idx[((size_t)b * N + i) * k + p] = best[p * dilation];
    }
  }
  free(yn); free(sq); free(d); free(best);
  return ORACLE_OK;
}

/* K2 forward. y: (B*N, C); idx: (B, N, k); u: (B*N, 2C) with u[2c] = y[c], u[2c+1] = max_j (y[idx_j][c] - y[c]); argmax (B*N, C)
 * optional: the FIRST j attaining the maximum (torch.max). */
int oracle_mr_aggregate_fwd(const float* y, const int32_t* idx, int B, int N, int C, int k, float* u, uint8_t* argmax) {
  if (!y || !idx || !u || B <= 0 || N <= 0 || C <= 0 || k <= 0 || k > 255) return ORACLE_EINVAL;
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      const size_t row = (size_t)b * N + n;
      for (int c = 0; c < C; ++c) {
        const float own = y[row * C + c];
        float best = -INFINITY;
        int arg = 0;
        for (int j = 0; j < k; ++j) {
          int m = idx[row * k + j];
          m = m < 0 ? 0 : (m >= N ? N - 1 : m);
          const float dd = y[((size_t)b * N + m) * C + c] - own;
          if (dd > best) { best = dd; arg = j; }
        }
        u[row * 2 * C + 2 * c] = own;
        u[row * 2 * C + 2 * c + 1] = best;
        if (argmax) argmax[row * C + c] = (uint8_t)arg;
      }
    }
  return ORACLE_OK;
}

/* K2 backward. du: (B*N, 2C); dy[n][c] = du[n][2c] - du[n][2c+1] + sum over (m, c) whose arg-max neighbour is n of du[m][2c+1]. */
int oracle_mr_aggregate_bwd(const float* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C, int k, float* dy) {
  if (!du || !idx || !argmax || !dy || B <= 0 || N <= 0 || C <= 0 || k <= 0) return ORACLE_EINVAL;
  double* acc = (double*)malloc(sizeof(double) * (size_t)N * C);
  if (!acc) return ORACLE_EINVAL;
  for (int b = 0; b < B; ++b) {
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c) {
        const size_t row = (size_t)b * N + n;
        acc[(size_t)n * C + c] = (double)du[row * 2 * C + 2 * c] - (double)du[row * 2 * C + 2 * c + 1];
      }
    for (int m = 0; m < N; ++m)
      for (int c = 0; c < C; ++c) {
        const size_t row = (size_t)b * N + m;
        int j = argmax[row * C + c];
        if (j >= k) j = k - 1;
        int t = idx[row * k + j];
        t = t < 0 ? 0 : (t >= N ? N - 1 : t);
        acc[(size_t)t * C + c] += (double)du[row * 2 * C + 2 * c + 1];
      }
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < C; ++c) dy[((size_t)b * N + n) * C + c] = (float)acc[(size_t)n * C + c];
  }
  free(acc);
  return ORACLE_OK;
}

/* K3 / K4. out (M, groups*Nout) = x (M, groups*K) W^T (+ bias), weight (groups*Nout, K): group g maps columns [gK, (g+1)K) to
 * [gNout, (g+1)Nout). */
int oracle_linear_fwd(const float* x, const float* w, const float* bias, int M, int Nout, int K, int groups, float* out) {
  if (!x || !w || !out || M <= 0 || Nout <= 0 || K <= 0 || groups <= 0) return ORACLE_EINVAL;
  for (int m = 0; m < M; ++m)
    for (int g = 0; g < groups; ++g)
      for (int o = 0; o < Nout; ++o) {
        double s = 0.0;
        const float* xr = x + (size_t)m * groups * K + (size_t)g * K;
        const float* wr = w + ((size_t)g * Nout + o) * K;
        for (int c = 0; c < K; ++c) s += (double)xr[c] * wr[c];
        float v = (float)s;
        if (bias) v += bias[g * Nout + o];
        out[(size_t)m * groups * Nout + (size_t)g * Nout + o] = v;
      }
  return ORACLE_OK;
}

/* nn.BatchNorm2d over the rows of x (M, C). training != 0: batch statistics (biased variance to normalise), running statistics
 * updated in place with momentum 0.1 and the UNBIASED variance; training == 0: running statistics. eps = 1e-5. */
int oracle_bn_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                  int training, float* out) {
  if (!x || !gamma || !beta || !running_mean || !running_var || !out || M <= 0 || C <= 0) return ORACLE_EINVAL;
  for (int c = 0; c < C; ++c) {
    float mean, var;
    if (training) {
      double s = 0.0;
      for (int m = 0; m < M; ++m) s += x[(size_t)m * C + c];
      const double mu = s / M;
      double v = 0.0;
      for (int m = 0; m < M; ++m) { const double dd = x[(size_t)m * C + c] - mu; v += dd * dd; }
      mean = (float)mu;
      var = (float)(v / M);
      const float unb = (float)(v / (M > 1 ? M - 1 : 1));
      running_mean[c] = (1.0f - 0.1f) * running_mean[c] + 0.1f * mean;
      running_var[c] = (1.0f - 0.1f) * running_var[c] + 0.1f * unb;
    } else {
      mean = running_mean[c];
      var = running_var[c];
    }
    const float inv = 1.0f / sqrtf(var + 1e-5f);
    for (int m = 0; m < M; ++m) out[(size_t)m * C + c] = (x[(size_t)m * C + c] - mean) * inv * gamma[c] + beta[c];
  }
  return ORACLE_OK;
}

/* K5. x: (B, N, C) node-major; w: (Co, C, 3, 3) — only w[:, :, t, 1] meets data on a width-1 map; out (B, No, Co),
 * No = (N + 2 - 3) / 2 + 1, out[n'] = sum_t W_t x[2 n' - 1 + t] + bias (zero node either side). */
int oracle_downsample3_fwd(const float* x, const float* w, const float* bias, int B, int N, int C, int Co, float* out) {
  if (!x || !w || !out || B <= 0 || N <= 0 || C <= 0 || Co <= 0) return ORACLE_EINVAL;
  const int No = (N + 2 - 3) / 2 + 1;
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < No; ++n)
      for (int o = 0; o < Co; ++o) {
        double s = 0.0;
        for (int t = 0; t < 3; ++t) {
          const int src = 2 * n - 1 + t;
          if (src < 0 || src >= N) continue;
          const float* xr = x + ((size_t)b * N + src) * C;
          for (int c = 0; c < C; ++c) s += (double)xr[c] * w[(((size_t)o * C + c) * 3 + t) * 3 + 1];
        }
        out[((size_t)b * No + n) * Co + o] = (float)s + (bias ? bias[o] : 0.0f);
      }
  return ORACLE_OK;
}

/* torch.linspace(0, 1, steps)[i] */
static float linspace01(int i, int steps) {
  const float step = 1.0f / (float)(steps - 1);
  return i < steps / 2 ? (float)i * step : 1.0f - (float)(steps - 1 - i) * step;
}

/* a1. spec (B, H, W); w (F, 3, pb, pf); out (B * NP, F) node-major, node = (mel patch, frame patch) row-major; planes time, freq,
 * normalised spectrogram. */
int oracle_peak_patchify_fwd(const float* spec, const float* w, const float* bias, int B, int H, int W, int pb, int pf, int F,
                             float* out) {
  if (!spec || !w || !bias || !out || B <= 0 || H <= 1 || W <= 1 || pb <= 0 || pf <= 0 || H % pb || W % pf || F <= 0) return ORACLE_EINVAL;
  const int Hp = H / pb, Wp = W / pf;
  for (int b = 0; b < B; ++b) {
    const float* x = spec + (size_t)b * H * W;
    float lo = x[0], hi = x[0];
    for (int i = 1; i < H * W; ++i) { if (x[i] < lo) lo = x[i]; if (x[i] > hi) hi = x[i]; }
    const float range = hi - lo;
    for (int ph = 0; ph < Hp; ++ph)
      for (int pw = 0; pw < Wp; ++pw)
        for (int f = 0; f < F; ++f) {
          double s = 0.0;
          for (int i = 0; i < pb; ++i)
            for (int j = 0; j < pf; ++j) {
              const int hh = ph * pb + i, ww = pw * pf + j;
              const float* wf = w + (size_t)f * 3 * pb * pf;
              s += (double)wf[i * pf + j] * linspace01(ww, W);
              s += (double)wf[pb * pf + i * pf + j] * linspace01(hh, H);
              s += (double)wf[2 * pb * pf + i * pf + j] * ((x[hh * W + ww] - lo) / range);
            }
          const float v = (float)s + bias[f];
          out[((size_t)b * Hp * Wp + (size_t)ph * Wp + pw) * F + f] = v < 0.0f ? 0.0f : v;   /* NaN (constant clip) propagates */
        }
  }
  return ORACLE_OK;
}

/* K6. z_i, z_j: (B, d). Rows 2p, 2p+1 of the interleaved z are the two views of pair p; the positive of row r is r ^ 1;
 * a = z z^T / tau with the diagonal masked; loss = mean_r (logsumexp_c a[r][c] - a[r][r^1]); dz_* = d loss / d z_* (may be NULL). */
int oracle_ntxent(const float* z_i, const float* z_j, int B, int d, float tau, float* loss, float* dz_i, float* dz_j) {
  if (!z_i || !z_j || !loss || B <= 0 || d <= 0 || !(tau > 0.0f)) return ORACLE_EINVAL;
  const int M = 2 * B;
  double* a = (double*)malloc(sizeof(double) * (size_t)M * M);
  double* p = (double*)malloc(sizeof(double) * (size_t)M * M);
  double* g = (double*)calloc((size_t)M * d, sizeof(double));
  if (!a || !p || !g) { free(a); free(p); free(g); return ORACLE_EINVAL; }
#define ZROW(r) (((r) & 1) ? z_j + (size_t)((r) >> 1) * d : z_i + (size_t)((r) >> 1) * d)
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < M; ++c) {
      double s = 0.0;
      const float* zr = ZROW(r);
      const float* zc = ZROW(c);
      for (int e = 0; e < d; ++e) s += (double)zr[e] * zc[e];
      a[(size_t)r * M + c] = (double)((float)s / tau);            /* the logits are fp32 in the reference */
    }
  double total = 0.0;
  for (int r = 0; r < M; ++r) {
    double mx = -INFINITY;
    for (int c = 0; c < M; ++c) if (c != r && a[(size_t)r * M + c] > mx) mx = a[(size_t)r * M + c];
    double se = 0.0;
    for (int c = 0; c < M; ++c) if (c != r) se += exp(a[(size_t)r * M + c] - mx);
    const double lse = mx + log(se);
    total += lse - a[(size_t)r * M + (r ^ 1)];
    for (int c = 0; c < M; ++c) p[(size_t)r * M + c] = c == r ? 0.0 : exp(a[(size_t)r * M + c] - lse);
  }
  *loss = (float)(total / M);
  if (dz_i && dz_j) {
    /* dL/da[r][c] = (p[r][c] - [c == r^1]) / M; a = z z^T / tau: dz[r] += sum_c G[r][c] z[c] / tau, dz[c] += G[r][c] z[r] / tau */
    for (int r = 0; r < M; ++r)
      for (int c = 0; c < M; ++c) {
        if (c == r) continue;
        const double G = (p[(size_t)r * M + c] - (c == (r ^ 1) ? 1.0 : 0.0)) / M / tau;
        const float* zr = ZROW(r);
        const float* zc = ZROW(c);
        for (int e = 0; e < d; ++e) {
          g[(size_t)r * d + e] += G * zc[e];
          g[(size_t)c * d + e] += G * zr[e];
        }
      }
    for (int r = 0; r < M; ++r) {
      float* dst = (r & 1) ? dz_j + (size_t)(r >> 1) * d : dz_i + (size_t)(r >> 1) * d;
      for (int e = 0; e < d; ++e) dst[e] = (float)g[(size_t)r * d + e];
    }
  }
#undef ZROW
  free(a); free(p); free(g);
  return ORACLE_OK;
}

int oracle_version(void) { return 1; }
