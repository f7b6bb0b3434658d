// Training-mode BatchNorm2d split around the GEMMs (all HBM-bound streaming kernels, one 16-byte chunk per lane):
//   GEMM epilogue -> per-row-tile column sums  -> bn_finalize (fp64 combine, running stats)  -> scale/shift
//   consumer GEMM applies scale/shift(+act) on load; the residual stream is materialised by bn_apply.
// Backward: bn_bwd_reduce (two column reductions) -> bn_bwd_finalize -> bn_bwd_apply (dr, in place).
// Activation tensors are fp32 or bf16 (template parameter T); statistics, coefficients and arithmetic are fp32/fp64.
#include <cstdlib>
#include "nsid_common.h"

namespace {

// 1024 threads = 16 tile groups x 64 channels: lanes run along channels (coalesced 256-B rows of the partial
// matrix), tile groups stride over the row tiles; partial sums are combined in fp64 in a fixed order (deterministic).
// Round 6: a workgroup that covers 64 channels pulls tiles x 64 x 2 floats through ONE CU (262 KB for the 512 row tiles of the C = 64
// layers at B = 256: 2-3 us at what a CU draws from L2, on a dependent chain where a microsecond per finalize launch is 0.3 ms of the
// step). With many row tiles the same 1024 threads cover 16 channels x 64 tile groups instead: four times the workgroups (CUs), a quarter
// of the bytes each. FIN_CH is a template parameter of the two finalize kernels; bn_fin_ch() chooses.
constexpr int FIN_THREADS = 1024;

template <int FIN_CH>
__device__ __forceinline__ void tile_sums(const float* __restrict__ p0, const float* __restrict__ p1, int tiles,
                                          int C, int c, int tg, bool ok, double* red0, double* red1, double& sum,
                                          double& sq) {
  constexpr int FIN_TG = FIN_THREADS / FIN_CH;
  double a = 0.0, b = 0.0;
  if (ok) {
    // all loads of a batch are issued before the first add (the fp64 chain would otherwise serialise one L2 round trip
    // per tile: this kernel is pure latency); the summation order is unchanged
    constexpr int UB = 8;
    int t = tg;
    for (; t + (UB - 1) * FIN_TG < tiles; t += UB * FIN_TG) {
      float va[UB], vb[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        va[u] = p0[(long)(t + u * FIN_TG) * C + c];
        vb[u] = p1[(long)(t + u * FIN_TG) * C + c];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) { a += (double)va[u]; b += (double)vb[u]; }
    }
    for (; t < tiles; t += FIN_TG) {
      a += (double)p0[(long)t * C + c];
      b += (double)p1[(long)t * C + c];
    }
  }
  red0[threadIdx.x] = a;
  red1[threadIdx.x] = b;
  __syncthreads();
  sum = 0.0; sq = 0.0;
  if (tg == 0)
    for (int g = 0; g < FIN_TG; ++g) {
      sum += red0[g * FIN_CH + (threadIdx.x & (FIN_CH - 1))];
      sq += red1[g * FIN_CH + (threadIdx.x & (FIN_CH - 1))];
    }
}

// running = (1 - m) * running + m * stat with every operation rounded on its own (no fused multiply-add): the two kernels
// that apply it (ordered finalize, deferred update) then agree bit for bit, and with torch's elementwise expression
__device__ __forceinline__ float running_blend(float old, float stat, float m) {
#pragma clang fp contract(off)      // (HIP's __fmul_rn / __fadd_rn are plain operators: they contract like any other)
  const float a = (1.f - m) * old;
  const float b = m * stat;
  return a + b;
}

template <int FIN_CH>
__global__ __launch_bounds__(FIN_THREADS) void bn_finalize_kernel(
    const float* __restrict__ stat, int tiles, int C, int M, const float* gamma, const float* beta,
    float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, float* scale, float* shift,
    float* mean_out, float* invstd_out, float* uvar_out) {
  __shared__ double red0[FIN_THREADS], red1[FIN_THREADS];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), tg = threadIdx.x / FIN_CH;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
  double sum, sq;
  tile_sums<FIN_CH>(stat, stat + (long)tiles * C, tiles, C, c, tg, c < C, red0, red1, sum, sq);
  if (tg != 0 || c >= C) return;
  const double mean = sum / M;
  double var = sq / M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
  if (uvar_out != nullptr) uvar_out[c] = (float)unbiased;      // for a deferred running-statistics update
  if (running_mean != nullptr) {
    running_mean[c] = running_blend(running_mean[c], (float)mean, momentum);
    running_var[c] = running_blend(running_var[c], (float)unbiased, momentum);
  }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * rsqrtf(rv[c] + eps);   // torch: (x-mean)/sqrt(var+eps)*w+b
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

// out = act(scale*r+shift) + residual, one chunk per thread, grid-stride
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ r, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int act,
                                                       const T* __restrict__ residual, T* __restrict__ out,
                                                       long nchunks, int CV) {
  constexpr int N = Chunk<T>::N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * N;
    float v[N], o[N], sc[N], sh[N];
    Chunk<T>::load(r + i * N, v);
    load_channels<N>(scale, c, sc);
    load_channels<N>(shift, c, sh);
#pragma unroll
    for (int e = 0; e < N; ++e) o[e] = nsid_act(sc[e] * v[e] + sh[e], act);
    if (residual != nullptr) {
      float rs[N];
      Chunk<T>::load(residual + i * N, rs);
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] += rs[e];
    }
    Chunk<T>::store(out + i * N, o);
  }
}

// Column reductions over a 128-row tile x 64-channel chunk per block (grid = tiles x C/64, so the deep stages with
// few row tiles still fill the chip). 256 threads = (row groups) x (chunks of the 64 channels); every thread walks its
// rows with all loads issued before the first use; fixed-order LDS combine (deterministic).
// MODE 0: column sums of `dout` (bias gradients).  MODE 1: BatchNorm backward pair (sum g, sum g*xhat).
constexpr int CR_CH = 64;

template <int MODE, typename T>
__global__ __launch_bounds__(256) void col_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ r,
                                                         long ld, int M, int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float slope,
                                                         float* __restrict__ partial, int tiles) {
  constexpr int N = Chunk<T>::N;
  constexpr int CQ = CR_CH / N;            // chunks across the 64 channels: 16 (fp32) or 8 (bf16)
  constexpr int RG = 256 / CQ;             // row groups: 16 or 32
  constexpr int ROWS = NSID_ROW_TILE / RG; // rows per thread: 8 or 4
  __shared__ float red[2][RG][CR_CH];
  const int tile = blockIdx.x;
  const int row0 = tile * NSID_ROW_TILE;
  const int t = threadIdx.x, q = t % CQ, rgp = t / CQ;
  const int c = blockIdx.y * CR_CH + N * q;
  const bool cok = c < C;
  float s0[N], s1[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s0[e] = s1[e] = 0.f;
  if (cok) {
    float d[ROWS][N], x[ROWS][N];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const int row = row0 + rgp + RG * i;
      const long off = (long)(row < M ? row : row0) * ld + c;
      Chunk<T>::load(dout + off, d[i]);
      if (MODE == 1) Chunk<T>::load(r + off, x[i]);
    }
    float sc[N], sh[N], mu[N], is[N];
    if (MODE == 1) {
      load_channels<N>(scale, c, sc);
      load_channels<N>(shift, c, sh);
      load_channels<N>(mean, c, mu);
      load_channels<N>(invstd, c, is);
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (row0 + rgp + RG * i >= M) continue;
#pragma unroll
      for (int e = 0; e < N; ++e) {
        if (MODE == 0) {
          s0[e] += d[i][e];
        } else {
          const float g = (sc[e] * x[i][e] + sh[e]) > 0.f ? d[i][e] : d[i][e] * slope;
          s0[e] += g;
          s1[e] += g * ((x[i][e] - mu[e]) * is[e]);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) {
    red[0][rgp][N * q + e] = s0[e];
    red[1][rgp][N * q + e] = s1[e];
  }
  __syncthreads();
  if (t < CR_CH) {
    const int cc = blockIdx.y * CR_CH + t;
    if (cc < C) {
      float a0 = 0.f, a1 = 0.f;
      for (int g = 0; g < RG; ++g) { a0 += red[0][g][t]; a1 += red[1][g][t]; }
      partial[(long)tile * C + cc] = a0;
      if (MODE == 1) partial[(long)tiles * C + (long)tile * C + cc] = a1;
    }
  }
}

// coef4 (optional): the BatchNorm backward as ONE affine of its two sources for a consumer that evaluates it on its operand load
// (gemm.hip ABN): dr = sc*(g - c0 - xhat*c1) = sc*g + P*r + Q with P = -sc*c1*invstd, Q = sc*(c1*invstd*mean - c0);
// coef4[4][C] = {sc, sh, P, Q} (sc, sh: the forward affine, needed for the activation mask g = dy * act'(sc*r + sh)).
template <int FIN_CH>
__global__ __launch_bounds__(FIN_THREADS) void bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                           int tiles, int C, int M, float* dgamma,
                                                                           float* dbeta, float* coef,
                                                                           const float* __restrict__ scale,
                                                                           const float* __restrict__ shift,
                                                                           const float* __restrict__ mean,
                                                                           const float* __restrict__ invstd,
                                                                           float* __restrict__ coef4) {
  __shared__ double red0[FIN_THREADS], red1[FIN_THREADS];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), tg = threadIdx.x / FIN_CH;
  double sg, sgx;
  tile_sums<FIN_CH>(partial, partial + (long)tiles * C, tiles, C, c, tg, c < C, red0, red1, sg, sgx);
  if (tg != 0 || c >= C) return;
  // atomic: the two views of a step may run this concurrently on different streams for the same layer
  if (dbeta) atomicAdd(dbeta + c, (float)sg);
  if (dgamma) atomicAdd(dgamma + c, (float)sgx);
  const float c0 = (float)(sg / M), c1 = (float)(sgx / M);
  if (coef != nullptr) {
    coef[c] = c0;
    coef[C + c] = c1;
  }
  if (coef4 != nullptr) {
    const float sc = scale[c], is = invstd[c];
    coef4[c] = sc;
    coef4[C + c] = shift[c];
    coef4[2 * C + c] = -sc * c1 * is;
    coef4[3 * C + c] = sc * (c1 * is * mean[c] - c0);
  }
}

// Every thread keeps ONE column chunk for the whole kernel (the launch makes the thread count a multiple of the chunks per
// row), so the six per-channel vectors are loaded once and U rows are in flight per iteration. With the parameters re-read
// for every 16-byte chunk the kernel moved 4x more bytes through the L1 path than through HBM and ran at half the rate of
// bn_apply for the same traffic (10.2 us against 5 us per launch).
template <typename T, int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ r,
                                                           long rows, int CV, const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float slope,
                                                           const float* __restrict__ coef, T* __restrict__ dr) {
  constexpr int N = Chunk<T>::N;
  const int C = CV * N;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, total = (long)gridDim.x * blockDim.x;
  const int cq = (int)(t % CV), c = cq * N;
  const long rstep = total / CV;                  // total % CV == 0 (host)
  float sc[N], sh[N], mu[N], is[N], c0[N], c1[N];
  load_channels<N>(scale, c, sc);
  load_channels<N>(shift, c, sh);
  load_channels<N>(mean, c, mu);
  load_channels<N>(invstd, c, is);
  load_channels<N>(coef, c, c0);
  load_channels<N>(coef + C, c, c1);
  for (long row = t / CV; row < rows; row += U * rstep) {
    float d[U][N], x[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = row + u * rstep;
      if (rr < rows) {
        Chunk<T>::load(dout + (rr * CV + cq) * N, d[u]);
        Chunk<T>::load(r + (rr * CV + cq) * N, x[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = row + u * rstep;
      if (rr < rows) {
        float o[N];
#pragma unroll
        for (int e = 0; e < N; ++e) {
          const float g = (sc[e] * x[u][e] + sh[e]) > 0.f ? d[u][e] : d[u][e] * slope;
          const float xh = (x[u][e] - mu[e]) * is[e];
          o[e] = sc[e] * (g - c0[e] - xh * c1[e]);
        }
        Chunk<T>::store(dr + (rr * CV + cq) * N, o);
      }
    }
  }
}

// derivative of the activation on the non-positive side (1 on the positive side): none 1, ReLU 0, LeakyReLU 0.2
inline float bn_slope(int act) { return act == NSID_ACT_RELU ? 0.f : (act == NSID_ACT_LEAKY ? 0.2f : 1.f); }

inline int stream_grid(long nchunks) {
  long b = (nchunks + 255) / 256;
  const long cap = nsid_tune(NSID_T_stream_max_wg);
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));   // cap and grid-stride (guide §6 G11)
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_atomic_kernel(const T* __restrict__ x, long ld, int M, int C,
                                                            float* __restrict__ out) {
  constexpr int N = Chunk<T>::N;
  constexpr int CQ = CR_CH / N, RG = 256 / CQ, ROWS = NSID_ROW_TILE / RG;
  __shared__ float red[RG][CR_CH];
  const int row0 = blockIdx.x * NSID_ROW_TILE;
  const int t = threadIdx.x, q = t % CQ, rgp = t / CQ;
  const int c = blockIdx.y * CR_CH + N * q;
  float s[N];
#pragma unroll
  for (int e = 0; e < N; ++e) s[e] = 0.f;
  if (c < C)
    for (int i = 0; i < ROWS; ++i) {
      const int row = row0 + rgp + RG * i;
      if (row < M) {
        float d[N];
        Chunk<T>::load(x + (long)row * ld + c, d);
#pragma unroll
        for (int e = 0; e < N; ++e) s[e] += d[e];
      }
    }
#pragma unroll
  for (int e = 0; e < N; ++e) red[rgp][N * q + e] = s[e];
  __syncthreads();
  if (t < CR_CH && blockIdx.y * CR_CH + t < C) {
    float a = 0.f;
    for (int g = 0; g < RG; ++g) a += red[g][t];
    atomicAdd(out + blockIdx.y * CR_CH + t, a);
  }
}

// thread count of the column-chunk-keeping streaming kernels: a multiple of the chunks per row, about one thread per U chunks,
// at most max_wg workgroups of 256
inline long chunk_keeping_grid(long nchunks, int CV, int U, long max_wg, bool* capped) {
  int g0 = CV, d256 = 256;
  while (d256 % 2 == 0 && g0 % 2 == 0) { d256 /= 2; g0 /= 2; }          // g0 = CV / gcd(CV, 256)
  long want = (nchunks + 256L * U - 1) / (256L * U);
  if (capped) *capped = want > max_wg;
  if (want > max_wg) want = max_wg;
  if (want < 1) want = 1;
  return (want + g0 - 1) / g0 * g0;
}

// Channel width of a finalize workgroup: 16 once there are many row tiles (tuning key bn_fin_tiles, 0 = always 64).
inline int bn_fin_ch(int tiles) {
  const int t = (int)nsid_tune(NSID_T_bn_fin_tiles);
  return (t > 0 && tiles >= t) ? 16 : 64;
}
#define NSID_FIN_LAUNCH(kernel, tiles, C, stream, ...)                                                                      \
  do {                                                                                                                       \
    if (bn_fin_ch(tiles) == 16)                                                                                              \
      NSID_LAUNCH(kernel<16>, dim3(((C) + 15) / 16), dim3(FIN_THREADS), 0, static_cast<hipStream_t>(stream), __VA_ARGS__);   \
    else                                                                                                                     \
      NSID_LAUNCH(kernel<64>, dim3(((C) + 63) / 64), dim3(FIN_THREADS), 0, static_cast<hipStream_t>(stream), __VA_ARGS__);   \
  } while (0)

}  // namespace

extern "C" int nsid_bn_finalize(const float* stat, int tiles, int C, int M, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* invstd, void* stream) {
  NSID_REQUIRE(stat && gamma && beta && scale && shift && mean && invstd && C > 0 && M > 0);
  NSID_REQUIRE(tiles == nsid_row_tiles(M));
  NSID_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  NSID_FIN_LAUNCH(bn_finalize_kernel, tiles, C, stream, stat, tiles, C, M, gamma, beta, running_mean, running_var, nbt,
              momentum, eps, scale, shift, mean, invstd, static_cast<float*>(nullptr));
  return nsid_launch_status();
}

// The same finalize without touching the running statistics; it also returns the unbiased variance so that
// nsid_bn_running_update can apply the update later. Used when the two views of a step run on two streams: the reference
// updates the running statistics view i first, then view j (simclr.py:36,42) — ordering every layer's two finalize
// kernels across the streams costs one cross-branch graph edge per layer (0.2 ms per step); deferring the updates to ONE
// launch after both forwards gives the same numbers (same float expressions, same order) with no edge at all.
extern "C" int nsid_bn_finalize_deferred(const float* stat, int tiles, int C, int M, const float* gamma,
                                         const float* beta, float eps, float* scale, float* shift, float* mean,
                                         float* invstd, float* uvar, void* stream) {
  NSID_REQUIRE(stat && gamma && beta && scale && shift && mean && invstd && uvar && C > 0 && M > 0);
  NSID_REQUIRE(tiles == nsid_row_tiles(M));
  NSID_FIN_LAUNCH(bn_finalize_kernel, tiles, C, stream, stat, tiles, C, M, gamma, beta, static_cast<float*>(nullptr),
              static_cast<float*>(nullptr), static_cast<int64_t*>(nullptr), 0.f, eps, scale, shift, mean, invstd, uvar);
  return nsid_launch_status();
}

namespace {
constexpr int RU_LAYERS = 16;       // layers per launch: the pointer table travels in the kernel arguments (16 x 7 x 8 B)
struct RunUpdArgs {
  float* rm[RU_LAYERS]; float* rv[RU_LAYERS]; int64_t* nbt[RU_LAYERS];
  const float* mean_a[RU_LAYERS]; const float* uvar_a[RU_LAYERS];
  const float* mean_b[RU_LAYERS]; const float* uvar_b[RU_LAYERS];      // null: one update only
  int C[RU_LAYERS];
  float momentum;
};
__global__ __launch_bounds__(256) void bn_running_update_kernel(const RunUpdArgs a) {
  const int l = blockIdx.x;
  const float m = a.momentum;
  const bool two = a.mean_b[l] != nullptr;
  if (threadIdx.x == 0 && a.nbt[l] != nullptr) *a.nbt[l] += two ? 2 : 1;
  for (int c = threadIdx.x; c < a.C[l]; c += blockDim.x) {
    // exactly the two successive updates bn_finalize_kernel would have made (view a, then view b)
    float rm = running_blend(a.rm[l][c], a.mean_a[l][c], m);
    float rv = running_blend(a.rv[l][c], a.uvar_a[l][c], m);
    if (two) {
      rm = running_blend(rm, a.mean_b[l][c], m);
      rv = running_blend(rv, a.uvar_b[l][c], m);
    }
    a.rm[l][c] = rm;
    a.rv[l][c] = rv;
  }
}
}  // namespace

// host arrays of n pointers each (mean_b / uvar_b may be null arrays or hold null entries: a single update)
extern "C" int nsid_bn_running_update(int n, const int* C, float* const* running_mean, float* const* running_var,
                                      int64_t* const* num_batches_tracked, const float* const* mean_a,
                                      const float* const* uvar_a, const float* const* mean_b,
                                      const float* const* uvar_b, float momentum, void* stream) {
  NSID_REQUIRE(n > 0 && C && running_mean && running_var && mean_a && uvar_a);
  for (int base = 0; base < n; base += RU_LAYERS) {
    RunUpdArgs a{};
    const int cnt = n - base < RU_LAYERS ? n - base : RU_LAYERS;
    for (int i = 0; i < cnt; ++i) {
      NSID_REQUIRE(C[base + i] > 0 && running_mean[base + i] && running_var[base + i] && mean_a[base + i] && uvar_a[base + i]);
      a.rm[i] = running_mean[base + i]; a.rv[i] = running_var[base + i];
      a.nbt[i] = num_batches_tracked ? num_batches_tracked[base + i] : nullptr;
      a.mean_a[i] = mean_a[base + i]; a.uvar_a[i] = uvar_a[base + i];
      a.mean_b[i] = mean_b ? mean_b[base + i] : nullptr; a.uvar_b[i] = uvar_b ? uvar_b[base + i] : nullptr;
      NSID_REQUIRE((a.mean_b[i] == nullptr) == (a.uvar_b[i] == nullptr));
      a.C[i] = C[base + i];
    }
    a.momentum = momentum;
    NSID_LAUNCH(bn_running_update_kernel, dim3(cnt), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    const int rc = nsid_launch_status();
    if (rc != NSID_OK) return rc;
  }
  return NSID_OK;
}

extern "C" int nsid_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, int C, float* scale, float* shift,
                                   void* stream) {
  NSID_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0);
  NSID_LAUNCH(bn_eval_affine_kernel, dim3((C + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), gamma, beta,
              running_mean, running_var, eps, C, scale, shift);
  return nsid_launch_status();
}

extern "C" int nsid_bn_apply(const void* r, const float* scale, const float* shift, int act, const void* residual,
                             void* out, int M, int C, int dtype, void* stream) {
  NSID_REQUIRE(r && scale && shift && out && M > 0 && C > 0 && NSID_DTYPE_OK(dtype));
  NSID_REQUIRE(C % (dtype == NSID_BF16 ? 8 : 4) == 0 && nsid_aligned16(r) && nsid_aligned16(out));
  NSID_DISPATCH_DTYPE(dtype, T, {
    const long nchunks = (long)M * C / Chunk<T>::N;
    NSID_LAUNCH((bn_apply_kernel<T>), dim3(stream_grid(nchunks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                static_cast<const T*>(r), scale, shift, act, static_cast<const T*>(residual), static_cast<T*>(out),
                nchunks, C / Chunk<T>::N);
  });
  return nsid_launch_status();
}

extern "C" int nsid_bn_bwd_reduce(const void* dout, const void* r, int M, int C, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, int act, float* partial,
                                  int dtype, void* stream) {
  NSID_REQUIRE(dout && r && scale && shift && mean && invstd && partial && M > 0 && C > 0 && NSID_DTYPE_OK(dtype));
  NSID_REQUIRE(C % (dtype == NSID_BF16 ? 8 : 4) == 0);
  NSID_REQUIRE(act == NSID_ACT_NONE || act == NSID_ACT_RELU || act == NSID_ACT_LEAKY);
  const int tiles = nsid_row_tiles(M);
  NSID_DISPATCH_DTYPE(dtype, T, {
    NSID_LAUNCH((col_reduce_kernel<1, T>), dim3(tiles, (C + CR_CH - 1) / CR_CH), dim3(256), 0,
                static_cast<hipStream_t>(stream), static_cast<const T*>(dout), static_cast<const T*>(r), (long)C, M, C,
                scale, shift, mean, invstd, bn_slope(act), partial, tiles);
  });
  return nsid_launch_status();
}

extern "C" int nsid_bn_bwd_finalize(const float* partial, int tiles, int C, int M, float* dgamma, float* dbeta,
                                    float* coef, void* stream) {
  NSID_REQUIRE(partial && coef && C > 0 && M > 0 && tiles >= 1);      // rows of partial sums: nsid_row_tiles(M) from a GEMM epilogue / the reduce pass, or one per clip (nsid_mr_aggregate_bwd_bn)
  NSID_FIN_LAUNCH(bn_bwd_finalize_kernel, tiles, C, stream, partial, tiles, C, M, dgamma, dbeta, coef, static_cast<const float*>(nullptr),
              static_cast<const float*>(nullptr), static_cast<const float*>(nullptr), static_cast<const float*>(nullptr),
              static_cast<float*>(nullptr));
  return nsid_launch_status();
}

extern "C" int nsid_bn_bwd_finalize_fused(const float* partial, int tiles, int C, int M, float* dgamma, float* dbeta,
                                          float* coef, const float* scale, const float* shift, const float* mean,
                                          const float* invstd, float* coef4, void* stream) {
  NSID_REQUIRE(partial && coef && coef4 && scale && shift && mean && invstd && C > 0 && M > 0 && tiles >= 1);
  NSID_FIN_LAUNCH(bn_bwd_finalize_kernel, tiles, C, stream, partial, tiles, C, M, dgamma, dbeta, coef, scale, shift, mean, invstd, coef4);
  return nsid_launch_status();
}

extern "C" int nsid_bn_bwd_apply(const void* dout, const void* r, int M, int C, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, int act,
                                 const float* coef, void* dr, int dtype, void* stream) {
  NSID_REQUIRE(dout && r && scale && shift && mean && invstd && coef && dr && M > 0 && C > 0 && NSID_DTYPE_OK(dtype));
  NSID_REQUIRE(C % (dtype == NSID_BF16 ? 8 : 4) == 0);
  NSID_REQUIRE(act == NSID_ACT_NONE || act == NSID_ACT_RELU || act == NSID_ACT_LEAKY);
  NSID_DISPATCH_DTYPE(dtype, T, {
    constexpr int U = 4;
    const int CV = C / Chunk<T>::N;
    const long nchunks = (long)M * CV;
    // thread count: a multiple of the chunks per row (a thread keeps its column chunk), about one thread per U chunks
    int g0 = CV, d256 = 256;
    while (d256 % 2 == 0 && g0 % 2 == 0) { d256 /= 2; g0 /= 2; }          // g0 = CV / gcd(CV, 256)
    long want = (nchunks + 256L * U - 1) / (256L * U);
    // At most 384 workgroups (1.5 per CU): with 2 048 the pass takes every wave slot and saturates HBM for its 9 us while the
    // OTHER view's branch of the step stalls (this pass was 88 % exposed in the two-stream step, tools/ablate.sh); fewer, longer
    // workgroups stream a little slower alone and leave room beside them. One-box A/B of the whole step, three repetitions each:
    // 2 048: 8.31 / 8.28 / 8.28 ms, 512: 8.21 / 8.23 / 8.23, 384: 8.22 / 8.18 / 8.17 (tuning key bn_bwd_apply_max_wg).
    const long max_wg = nsid_tune(NSID_T_bn_bwd_apply_max_wg);
    nsid_count(NSID_C_bn_bwd_apply);
    if (want > max_wg) { want = max_wg; nsid_count(NSID_C_bn_bwd_apply_capped); }
    const long grid = (want + g0 - 1) / g0 * g0;
    NSID_LAUNCH((bn_bwd_apply_kernel<T, U>), dim3((int)grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                static_cast<const T*>(dout), static_cast<const T*>(r), (long)M, CV, scale, shift, mean,
                invstd, bn_slope(act), coef, static_cast<T*>(dr));
  });
  return nsid_launch_status();
}

// out[c] += sum_m x[m,c] (bias gradients of layers that are NOT in front of a BatchNorm: proj, projector).
extern "C" int nsid_colsum_acc(const void* x, int ldx, int M, int C, float* out, int dtype, void* stream) {
  NSID_REQUIRE(x && out && M > 0 && C > 0 && NSID_DTYPE_OK(dtype) && nsid_aligned16(x));
  NSID_REQUIRE(C % (dtype == NSID_BF16 ? 8 : 4) == 0 && ldx % (dtype == NSID_BF16 ? 8 : 4) == 0);
  NSID_DISPATCH_DTYPE(dtype, T, {
    NSID_LAUNCH((colsum_atomic_kernel<T>), dim3(nsid_row_tiles(M), (C + CR_CH - 1) / CR_CH), dim3(256), 0,
                static_cast<hipStream_t>(stream), static_cast<const T*>(x), (long)ldx, M, C, out);
  });
  return nsid_launch_status();
}
