// Training-mode BatchNorm2d split around the GEMMs (all HBM-bound streaming kernels, 16 B per lane):
//   GEMM epilogue -> per-row-tile column sums  -> bn_finalize (fp64 combine, running stats)  -> scale/shift
//   consumer GEMM applies scale/shift(+act) on load; the residual stream is materialised by bn_apply.
// Backward: bn_bwd_reduce (two column reductions) -> bn_bwd_finalize -> bn_bwd_apply (dr, in place).
#include "nsid_common.h"

namespace {

// 1024 threads = 16 tile groups x 64 channels: lanes run along channels (coalesced 256-B rows of the partial
// matrix), tile groups stride over the row tiles; partial sums are combined in fp64 in a fixed order (deterministic).
constexpr int FIN_CH = 64, FIN_TG = 16;

__device__ __forceinline__ void tile_sums(const float* __restrict__ p0, const float* __restrict__ p1, int tiles,
                                          int C, int c, int tg, bool ok, double* red0, double* red1, double& sum,
                                          double& sq) {
  double a = 0.0, b = 0.0;
  if (ok)
    for (int t = tg; t < tiles; t += FIN_TG) {
      a += (double)p0[(long)t * C + c];
      b += (double)p1[(long)t * C + c];
    }
  red0[threadIdx.x] = a;
  red1[threadIdx.x] = b;
  __syncthreads();
  sum = 0.0; sq = 0.0;
  if (tg == 0)
    for (int g = 0; g < FIN_TG; ++g) { sum += red0[g * FIN_CH + (threadIdx.x & (FIN_CH - 1))]; sq += red1[g * FIN_CH + (threadIdx.x & (FIN_CH - 1))]; }
}

__global__ __launch_bounds__(FIN_CH * FIN_TG) void bn_finalize_kernel(
    const float* __restrict__ stat, int tiles, int C, int M, const float* gamma, const float* beta,
    float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, float* scale, float* shift,
    float* mean_out, float* invstd_out) {
  __shared__ double red0[FIN_CH * FIN_TG], red1[FIN_CH * FIN_TG];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), tg = threadIdx.x / FIN_CH;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
  double sum, sq;
  tile_sums(stat, stat + (long)tiles * C, tiles, C, c, tg, c < C, red0, red1, sum, sq);
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
  if (running_mean != nullptr) {
    const double unbiased = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
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

// out = act(scale*r+shift) + residual, float4 per thread, grid-stride
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ r, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int act,
                                                       const float* __restrict__ residual, float* __restrict__ out,
                                                       long n4, int C4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    f32x4 v = reinterpret_cast<const f32x4*>(r)[i];
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = nsid_act(sc[e] * v[e] + sh[e], act);
    if (residual != nullptr) {
      const f32x4 rs = reinterpret_cast<const f32x4*>(residual)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += rs[e];
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}

// Column reductions over a 128-row tile. Block = 256 threads laid out as (row groups) x (column quads);
// wide matrices are walked in column passes. MODE 0: sums of x (colsum). MODE 1: BN backward pair.
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ r,
                                                         long ld, int M, int C, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, int act,
                                                         float* __restrict__ partial, int tiles) {
  __shared__ f32x4 red[2][256];
  const int tile = blockIdx.x;
  const int row0 = tile * NSID_ROW_TILE;
  const int rows = min(NSID_ROW_TILE, M - row0);
  const int C4 = C >> 2;
  const int cpp = C4 < 256 ? C4 : 256;         // column quads per pass (power of two or C4 itself)
  // row groups: largest power of two with cpp*rg <= 256
  int rg = 1;
  while (cpp * rg * 2 <= 256) rg *= 2;
  const int t = threadIdx.x;
  const int my_c = t % cpp, my_g = t / cpp;
  const bool active = my_g < rg;
  for (int cbase = 0; cbase < C4; cbase += cpp) {
    const int c4 = cbase + my_c;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (active && c4 < C4) {
      f32x4 sc, sh, mu, is;
      if (MODE == 1) {
        sc = *reinterpret_cast<const f32x4*>(scale + 4 * c4);
        sh = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
        mu = *reinterpret_cast<const f32x4*>(mean + 4 * c4);
        is = *reinterpret_cast<const f32x4*>(invstd + 4 * c4);
      }
      for (int rr = my_g; rr < rows; rr += rg) {
        const long off = (long)(row0 + rr) * ld + 4 * c4;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dout + off);
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s0[e] += d[e];
        } else {
          const f32x4 x = *reinterpret_cast<const f32x4*>(r + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = d[e] * nsid_act_grad(sc[e] * x[e] + sh[e], act);
            s0[e] += g;
            s1[e] += g * ((x[e] - mu[e]) * is[e]);
          }
        }
      }
    }
    red[0][t] = s0;
    red[1][t] = s1;
    __syncthreads();
    if (t < cpp && cbase + t < C4) {     // fixed-order sum over the row groups: deterministic
      f32x4 a0 = red[0][t], a1 = red[1][t];
      for (int gI = 1; gI < rg; ++gI) {
        const f32x4 b0 = red[0][gI * cpp + t], b1 = red[1][gI * cpp + t];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] += b0[e]; a1[e] += b1[e]; }
      }
      const long o = (long)tile * C + 4 * (cbase + t);
      *reinterpret_cast<f32x4*>(partial + o) = a0;
      if (MODE == 1) *reinterpret_cast<f32x4*>(partial + (long)tiles * C + o) = a1;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(FIN_CH * FIN_TG) void bn_bwd_finalize_kernel(const float* __restrict__ partial,
                                                                           int tiles, int C, int M, float* dgamma,
                                                                           float* dbeta, float* coef) {
  __shared__ double red0[FIN_CH * FIN_TG], red1[FIN_CH * FIN_TG];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), tg = threadIdx.x / FIN_CH;
  double sg, sgx;
  tile_sums(partial, partial + (long)tiles * C, tiles, C, c, tg, c < C, red0, red1, sg, sgx);
  if (tg != 0 || c >= C) return;
  if (dbeta) dbeta[c] += (float)sg;
  if (dgamma) dgamma[c] += (float)sgx;
  coef[c] = (float)(sg / M);
  coef[C + c] = (float)(sgx / M);
}

__global__ void colsum_finalize_kernel(const float* __restrict__ partial, int tiles, int C, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int t = 0; t < tiles; ++t) s += (double)partial[(long)t * C + c];
  out[c] += (float)s;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ r,
                                                           long n4, int C4, const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, int act,
                                                           const float* __restrict__ coef, float* __restrict__ dr) {
  const int C = C4 * 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const f32x4 d = reinterpret_cast<const f32x4*>(dout)[i];
    const f32x4 x = reinterpret_cast<const f32x4*>(r)[i];
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(coef + c);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(coef + C + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = d[e] * nsid_act_grad(sc[e] * x[e] + sh[e], act);
      const float xh = (x[e] - mu[e]) * is[e];
      o[e] = sc[e] * (g - c0[e] - xh * c1[e]);
    }
    reinterpret_cast<f32x4*>(dr)[i] = o;
  }
}

inline int stream_grid(long n4) {
  long b = (n4 + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));   // cap and grid-stride (guide §6 G11)
}

}  // namespace

extern "C" int nsid_bn_finalize(const float* stat, int tiles, int C, int M, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* invstd, void* stream) {
  NSID_REQUIRE(stat && gamma && beta && scale && shift && mean && invstd && C > 0 && M > 0);
  NSID_REQUIRE(tiles == nsid_row_tiles(M));
  NSID_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  NSID_LAUNCH(bn_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_TG), 0, static_cast<hipStream_t>(stream), stat,
                     tiles, C, M, gamma, beta, running_mean, running_var, nbt, momentum, eps, scale, shift, mean,
                     invstd);
  return nsid_launch_status();
}

extern "C" int nsid_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, int C, float* scale, float* shift,
                                   void* stream) {
  NSID_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0);
  NSID_LAUNCH(bn_eval_affine_kernel, dim3((C + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream),
                     gamma, beta, running_mean, running_var, eps, C, scale, shift);
  return nsid_launch_status();
}

extern "C" int nsid_bn_apply(const float* r, const float* scale, const float* shift, int act, const float* residual,
                             float* out, int M, int C, void* stream) {
  NSID_REQUIRE(r && scale && shift && out && M > 0 && C > 0 && C % 4 == 0);
  NSID_REQUIRE(nsid_aligned16(r) && nsid_aligned16(out) && nsid_aligned16(scale) && nsid_aligned16(shift));
  const long n4 = (long)M * C / 4;
  NSID_LAUNCH(bn_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, static_cast<hipStream_t>(stream), r, scale,
                     shift, act, residual, out, n4, C / 4);
  return nsid_launch_status();
}

extern "C" int nsid_bn_bwd_reduce(const float* dout, const float* r, int M, int C, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, int act, float* partial,
                                  void* stream) {
  NSID_REQUIRE(dout && r && scale && shift && mean && invstd && partial && M > 0 && C > 0 && C % 4 == 0);
  const int tiles = nsid_row_tiles(M);
  NSID_LAUNCH((col_reduce_kernel<1>), dim3(tiles), dim3(256), 0, static_cast<hipStream_t>(stream), dout, r,
                     (long)C, M, C, scale, shift, mean, invstd, act, partial, tiles);
  return nsid_launch_status();
}

extern "C" int nsid_bn_bwd_finalize(const float* partial, int tiles, int C, int M, float* dgamma, float* dbeta,
                                    float* coef, void* stream) {
  NSID_REQUIRE(partial && coef && C > 0 && M > 0 && tiles == nsid_row_tiles(M));
  NSID_LAUNCH(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH), dim3(FIN_CH * FIN_TG), 0, static_cast<hipStream_t>(stream),
                     partial, tiles, C, M, dgamma, dbeta, coef);
  return nsid_launch_status();
}

extern "C" int nsid_bn_bwd_apply(const float* dout, const float* r, int M, int C, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, int act,
                                 const float* coef, float* dr, void* stream) {
  NSID_REQUIRE(dout && r && scale && shift && mean && invstd && coef && dr && M > 0 && C > 0 && C % 4 == 0);
  const long n4 = (long)M * C / 4;
  NSID_LAUNCH(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, static_cast<hipStream_t>(stream), dout,
                     r, n4, C / 4, scale, shift, mean, invstd, act, coef, dr);
  return nsid_launch_status();
}

// out[c] += sum_m x[m,c]. Two launches (tile partials in a caller-invisible static scratch would need allocation,
// so the partial buffer is carved from `out`'s caller: see the Python host) — here: direct atomics per tile.
namespace {
__global__ __launch_bounds__(256) void colsum_atomic_kernel(const float* __restrict__ x, long ld, int M, int C,
                                                            float* __restrict__ out) {
  // each block reduces a 128-row tile per column quad, then one atomic per column
  __shared__ f32x4 red[256];
  const int row0 = blockIdx.x * NSID_ROW_TILE;
  const int rows = min(NSID_ROW_TILE, M - row0);
  const int C4 = C >> 2;
  const int cpp = C4 < 256 ? C4 : 256;
  int rg = 1;
  while (cpp * rg * 2 <= 256) rg *= 2;
  const int t = threadIdx.x, my_c = t % cpp, my_g = t / cpp;
  for (int cbase = 0; cbase < C4; cbase += cpp) {
    const int c4 = cbase + my_c;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (my_g < rg && c4 < C4)
      for (int rr = my_g; rr < rows; rr += rg) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(x + (long)(row0 + rr) * ld + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += d[e];
      }
    red[t] = s;
    __syncthreads();
    if (t < cpp && cbase + t < C4) {
      f32x4 a = red[t];
      for (int gI = 1; gI < rg; ++gI) {
        const f32x4 b = red[gI * cpp + t];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += b[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(out + 4 * (cbase + t) + e, a[e]);
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int nsid_colsum_acc(const float* x, int ldx, int M, int C, float* out, void* stream) {
  NSID_REQUIRE(x && out && M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && nsid_aligned16(x));
  NSID_LAUNCH(colsum_atomic_kernel, dim3(nsid_row_tiles(M)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     (long)ldx, M, C, out);
  return nsid_launch_status();
}
