// Fused NT-Xent (simclr/ntxent.py:5-30): similarity GEMM on fp32 MFMA + masked online log-sum-exp + positive
// pick (pass 1), and dZ = (G + G^T) Z / tau (pass 2). The (2B x 2B) logits never leave the accumulators.
//
// Orientation trick: each wave computes the TRANSPOSED tile T[j][i] = z_j . z_i (A operand = column block rows j,
// B operand = its own 16 rows i). In the C/D layout a lane then holds, for its own row i = lane&15, the four
// logits j = 4*(lane>>4)+reg — which is exactly the A-operand layout (row i, reduction index (lane>>4, reg)) of the
// second product dZ[i][:] += Q[i][j] z_j[:], so Q is consumed from registers with no LDS round trip or shuffle.
#include "nsid_common.h"

namespace {

constexpr int RB = 16;   // rows i per workgroup (one MFMA row tile, shared by the 4 waves)
constexpr int CBK = 128; // rows j per staged column block (64 KB of LDS: one staging latency + two barriers per 128 columns)

__device__ __forceinline__ const float* zrow(const float* z_i, const float* z_j, int row, int d) {
  return ((row & 1) ? z_j : z_i) + (long)(row >> 1) * d;
}

__device__ __forceinline__ void stage_rows(float* dst, int ld, const float* z_i, const float* z_j, int row0, int nrows,
                                           int M, int d) {
  const int d4 = d >> 2;
  constexpr int UB = 8;          // all loads of a batch are issued before the first LDS store (the plain loop was one
  //                                global round trip per iteration: 16 serial trips per 128-row block)
  for (int q0 = threadIdx.x; q0 < nrows * d4; q0 += UB * blockDim.x) {
    f32x4 v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = q0 + u * blockDim.x;
      const int rr = q / d4, c = (q % d4) * 4;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < nrows * d4 && row0 + rr < M) v[u] = *reinterpret_cast<const f32x4*>(zrow(z_i, z_j, row0 + rr, d) + c);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = q0 + u * blockDim.x;
      const int rr = q / d4, c = (q % d4) * 4;
      if (q < nrows * d4) *reinterpret_cast<f32x4*>(dst + rr * ld + c) = v[u];
    }
  }
}

// Column blocks are double-buffered through registers: the loads of block j0+CBK are issued before the MFMA / exp work on
// block j0 and written to LDS after it (one staging latency per block was ~2/3 of these kernels at the 8-GPU global batch:
// 32 blocks of 128 rows per workgroup). PF chunks of 16 bytes per thread: CBK*d/4/256 <= 16 (d <= 128).
constexpr int PF_MAX = 16;
struct Prefetch { f32x4 v[PF_MAX]; };

__device__ __forceinline__ void prefetch_rows(Prefetch& pf, const float* z_i, const float* z_j, int row0, int nrows, int M,
                                              int d) {
  const int d4 = d >> 2;
#pragma unroll
  for (int u = 0; u < PF_MAX; ++u) {
    const int q = threadIdx.x + u * 256;
    const int rr = q / d4, c = (q % d4) * 4;
    pf.v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < nrows * d4 && row0 + rr < M) pf.v[u] = *reinterpret_cast<const f32x4*>(zrow(z_i, z_j, row0 + rr, d) + c);
  }
}

__device__ __forceinline__ void commit_rows(float* dst, int ld, const Prefetch& pf, int nrows, int d) {
  const int d4 = d >> 2;
#pragma unroll
  for (int u = 0; u < PF_MAX; ++u) {
    const int q = threadIdx.x + u * 256;
    const int rr = q / d4, c = (q % d4) * 4;
    if (q < nrows * d4) *reinterpret_cast<f32x4*>(dst + rr * ld + c) = pf.v[u];
  }
}

// NT T-tiles at once (column rows j .. j+16*NT-1): independent accumulator chains keep the matrix pipe issuing
template <int NT>
__device__ __forceinline__ void sim_tiles(const float* zi_w, const float* zj_t, int ld, int d, int lr, int rq,
                                          f32x4 (&acc)[NT]) {
#pragma unroll
  for (int u = 0; u < NT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* pa = zj_t + lr * ld + 4 * rq;     // A: row j = lr of tile u
  const float* pb = zi_w + lr * ld + 4 * rq;     // B: column i = lr
  for (int ch = 0; ch < d; ch += 16) {
    const f32x4 fb = *reinterpret_cast<const f32x4*>(pb + ch);
    f32x4 fa[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) fa[u] = *reinterpret_cast<const f32x4*>(pa + u * 16 * ld + ch);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[e], acc[u], 0, 0, 0);
  }
}

// Work decomposition of both passes: a workgroup owns ONE 16-row tile (RB rows i); its 4 waves split every staged block of
// CBK column rows j into quarters (CBK/64 tiles each), so the exp/log work of a row tile — the long pole of these kernels
// (dependent VALU chains, one wave per SIMD) — is spread over 4x the waves and 4x the CUs of the 64-row form, and the
// global batch of the 8-GPU run (M = 4096 rows on every rank) gets 256 workgroups. The partial results of the 4 waves
// meet in LDS at the end.

// pass 1: lse[i] = logsumexp_{j != i} a_ij and rowloss[i] = lse[i] - a_{i, i^1}, for every row i < M
// grid.y > 1: the columns are split over workgroups and every workgroup writes its partial (max, sum, positive) of a row to
// part_out[group][3][M]; ntxent_lse_merge_kernel combines them (online-softmax merge).
__global__ __launch_bounds__(256) void ntxent_lse_kernel(const float* __restrict__ z_i, const float* __restrict__ z_j,
                                                         int M, int d, float tau, int cols_per_group,
                                                         float* __restrict__ lse, float* __restrict__ rowloss,
                                                         float* __restrict__ part_out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = d + 4;
  float* zi_s = sm;                 // [RB][ld]
  float* zj_s = sm + RB * ld;       // [CBK][ld]
  float* part = zj_s + CBK * ld;    // [4 waves][3][16]
  const int i0 = blockIdx.x * RB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, rq = lane >> 4;
  stage_rows(zi_s, ld, z_i, z_j, i0, RB, M, d);
  const int i = i0 + lr;
  float mrun = -__builtin_inff(), srun = 0.f, pos = 0.f;
  constexpr int TPW = CBK / 16 / 4;                   // column tiles per wave per block
  const bool pipelined = CBK * (d >> 2) <= PF_MAX * 256;     // uniform
  Prefetch pf;
  const int jbeg = blockIdx.y * cols_per_group, jend = min(M, jbeg + cols_per_group);
  if (pipelined) {
    prefetch_rows(pf, z_i, z_j, jbeg, CBK, M, d);
    __syncthreads();                                          // zi_s staged, nobody reads zj_s yet
    commit_rows(zj_s, ld, pf, CBK, d);
  }
  for (int j0 = jbeg; j0 < jend; j0 += CBK) {
    if (pipelined) {
      __syncthreads();                                        // block j0 is in LDS
      prefetch_rows(pf, z_i, z_j, j0 + CBK, CBK, M, d);       // next block: in flight under the work below (zeros past M)
    } else {
      __syncthreads();
      stage_rows(zj_s, ld, z_i, z_j, j0, CBK, M, d);
      __syncthreads();
    }
    f32x4 acc[TPW];
    sim_tiles<TPW>(zi_s, zj_s + 16 * (wave * TPW) * ld, ld, d, lr, rq, acc);
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + 16 * (wave * TPW + u) + 4 * rq + e;
        const float a = acc[u][e] / tau;
        if (j < M && j != i) {
          if (j == (i ^ 1)) pos = a;
          const float mn = fmaxf(mrun, a);
          srun = srun * expf(mrun - mn) + expf(a - mn);    // mrun = -inf first time: exp(-inf) = 0
          mrun = mn;
        }
      }
    }
    if (pipelined) {
      __syncthreads();                                        // every wave is done with block j0
      commit_rows(zj_s, ld, pf, CBK, d);
    }
  }
  // merge the four lanes (rq = 0..3) that share row i, then the four waves
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    const float m2 = __shfl_xor(mrun, o, 64), s2 = __shfl_xor(srun, o, 64), p2 = __shfl_xor(pos, o, 64);
    const float mn = fmaxf(mrun, m2);
    const float e1 = mrun == -__builtin_inff() ? 0.f : expf(mrun - mn);
    const float e2 = m2 == -__builtin_inff() ? 0.f : expf(m2 - mn);
    srun = srun * e1 + s2 * e2;
    mrun = mn;
    pos += p2;
  }
  if (rq == 0) { part[(wave * 3 + 0) * 16 + lr] = mrun; part[(wave * 3 + 1) * 16 + lr] = srun; part[(wave * 3 + 2) * 16 + lr] = pos; }
  __syncthreads();
  if (threadIdx.x < 16 && i0 + (int)threadIdx.x < M) {
    float m = -__builtin_inff(), sacc = 0.f, p = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float m2 = part[(w * 3 + 0) * 16 + threadIdx.x], s2 = part[(w * 3 + 1) * 16 + threadIdx.x];
      const float mn = fmaxf(m, m2);
      const float e1 = m == -__builtin_inff() ? 0.f : expf(m - mn);
      const float e2 = m2 == -__builtin_inff() ? 0.f : expf(m2 - mn);
      sacc = sacc * e1 + s2 * e2;
      m = mn;
      p += part[(w * 3 + 2) * 16 + threadIdx.x];
    }
    if (gridDim.y > 1) {
      float* po = part_out + (long)blockIdx.y * 3 * M + i0 + threadIdx.x;
      po[0] = m; po[M] = sacc; po[2 * M] = p;
    } else {
      const float l = m + logf(sacc);
      lse[i0 + threadIdx.x] = l;
      rowloss[i0 + threadIdx.x] = l - p;
    }
  }
}

__global__ void ntxent_lse_merge_kernel(const float* __restrict__ part, int groups, int M, float* __restrict__ lse,
                                        float* __restrict__ rowloss) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  float m = -__builtin_inff(), sacc = 0.f, p = 0.f;
  for (int g = 0; g < groups; ++g) {
    const float m2 = part[(long)g * 3 * M + i], s2 = part[(long)g * 3 * M + M + i];
    const float mn = fmaxf(m, m2);
    const float e1 = m == -__builtin_inff() ? 0.f : expf(m - mn);
    const float e2 = m2 == -__builtin_inff() ? 0.f : expf(m2 - mn);
    sacc = sacc * e1 + s2 * e2;
    m = mn;
    p += part[(long)g * 3 * M + 2 * M + i];
  }
  const float l = m + logf(sacc);
  lse[i] = l;
  rowloss[i] = l - p;
}

// pass 2: dz[i] = (1/(M tau)) * sum_j Q_ij z_j,  Q_ij = exp(a_ij - lse_i) + exp(a_ij - lse_j) - 2[j == i^1], Q_ii = 0
template <int DT>   // d / 16 column tiles of the output
__global__ __launch_bounds__(256) void ntxent_grad_kernel(const float* __restrict__ z_i, const float* __restrict__ z_j,
                                                          int M, int d, float tau, const float* __restrict__ lse,
                                                          int row0, int nrows, int cols_per_group,
                                                          float* __restrict__ dz_i, float* __restrict__ dz_j) {
  // grid.y > 1: the column blocks are split over workgroups (few own rows against a large global batch: 32 row tiles at the
  // 8-GPU batch would leave 7/8 of the chip idle); the partial dZ rows are then added atomically into zeroed dz_i / dz_j
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ld = d + 4;
  float* zi_s = sm;
  float* zj_s = sm + RB * ld;       // also the [4 waves][16][ld] reduction scratch of the epilogue (CBK >= 64)
  const int i0 = row0 + blockIdx.x * RB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, rq = lane >> 4;
  stage_rows(zi_s, ld, z_i, z_j, i0, RB, M, d);
  const int i = i0 + lr;
  const bool iok = i < row0 + nrows && i < M;
  const float lse_i = iok ? lse[i] : 0.f;
  f32x4 out[DT];
#pragma unroll
  for (int c = 0; c < DT; ++c) out[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int TPW = CBK / 16 / 4;
  const bool pipelined = CBK * (d >> 2) <= PF_MAX * 256;     // uniform
  Prefetch pf;
  const int jbeg = blockIdx.y * cols_per_group, jend = min(M, jbeg + cols_per_group);
  if (pipelined) {
    prefetch_rows(pf, z_i, z_j, jbeg, CBK, M, d);
    __syncthreads();                                          // zi_s staged, nobody reads zj_s yet
    commit_rows(zj_s, ld, pf, CBK, d);
  }
  for (int j0 = jbeg; j0 < jend; j0 += CBK) {
    if (pipelined) {
      __syncthreads();                                        // block j0 is in LDS
      prefetch_rows(pf, z_i, z_j, j0 + CBK, CBK, M, d);       // next block: in flight under the work below (zeros past M)
    } else {
      __syncthreads();
      stage_rows(zj_s, ld, z_i, z_j, j0, CBK, M, d);
      __syncthreads();
    }
    f32x4 acc[TPW];
    sim_tiles<TPW>(zi_s, zj_s + 16 * (wave * TPW) * ld, ld, d, lr, rq, acc);
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int jt = wave * TPW + u;
      f32x4 q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = j0 + 16 * jt + 4 * rq + e;
        float v = 0.f;
        if (iok && j < M && j != i) {
          const float a = acc[u][e] / tau;
          v = expf(a - lse_i) + expf(a - lse[j]);
          if (j == (i ^ 1)) v -= 2.f;
        }
        q[e] = v;
      }
      // dZ tile: A = Q (row i = lr, reduction (rq, e) <-> j = 4*rq+e), B = z_j[j][16*c + lr]
      const float* zb = zj_s + (16 * jt + 4 * rq) * ld + lr;
#pragma unroll
      for (int c = 0; c < DT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          out[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], zb[e * ld + 16 * c], out[c], 0, 0, 0);
    }
    if (pipelined) {
      __syncthreads();
      commit_rows(zj_s, ld, pf, CBK, d);
    }
  }
  // the 4 waves hold partial sums over disjoint column sets: add them through LDS.
  // C/D layout of out[c]: column = feature 16*c + lr, row = local row 4*rq + reg
  __syncthreads();
  float* red = zj_s;                                       // [4][16][ld]
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < DT; ++c) red[(wave * 16 + 4 * rq + e) * ld + 16 * c + lr] = out[c][e];
  __syncthreads();
  const float sc = 1.f / ((float)M * tau);
  for (int q = threadIdx.x; q < RB * d; q += blockDim.x) {
    const int rr = q / d, c = q % d;
    const int row = i0 + rr;
    if (row < row0 + nrows && row < M) {
      const float v = (red[rr * ld + c] + red[(16 + rr) * ld + c]) + (red[(32 + rr) * ld + c] + red[(48 + rr) * ld + c]);
      float* dst = ((row & 1) ? dz_j : dz_i) + (long)((row >> 1) - (row0 >> 1)) * d;
      if (gridDim.y > 1) atomicAdd(dst + c, v * sc);
      else dst[c] = v * sc;
    }
  }
}

__global__ __launch_bounds__(256) void ntxent_loss_kernel(const float* __restrict__ rowloss, int row0, int nrows, int M,
                                                          float* __restrict__ loss_out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nrows; i += blockDim.x) s += (double)rowloss[row0 + i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss_out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / (double)M);
}

template <typename K>
int raise_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return NSID_OK;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess ? NSID_OK : NSID_ELAUNCH;
}

}  // namespace

constexpr int LSE_MAX_GROUPS = 16;
extern "C" size_t nsid_ntxent_ws_floats(int Bg) { return (size_t)4 * Bg + 8 + (size_t)LSE_MAX_GROUPS * 3 * 2 * Bg; }

extern "C" int nsid_ntxent_fwd_bwd(const float* z_i, const float* z_j, int Bg, int d, float tau, int p0, int np,
                                   float* ws, float* loss_out, float* dz_i, float* dz_j, void* stream) {
  NSID_REQUIRE(z_i && z_j && ws && loss_out && Bg > 0 && np > 0 && p0 >= 0 && p0 + np <= Bg && tau > 0.f);
  NSID_REQUIRE(d % 16 == 0 && d <= 256 && nsid_aligned16(z_i) && nsid_aligned16(z_j));
  NSID_REQUIRE((dz_i == nullptr) == (dz_j == nullptr));
  const int M = 2 * Bg;
  float* lse = ws;
  float* rowloss = ws + M;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t bytes = ((size_t)(RB + CBK) * (d + 4) + 4 * 3 * 16) * sizeof(float);
  if (raise_lds(ntxent_lse_kernel, bytes) != NSID_OK) return NSID_ELAUNCH;
  {
    const int row_tiles = (M + RB - 1) / RB, nblocks = (M + CBK - 1) / CBK;
    int cg = row_tiles >= 512 ? 1 : (512 + row_tiles - 1) / row_tiles;        // ~512 workgroups
    if (cg > nblocks) cg = nblocks;
    if (cg > LSE_MAX_GROUPS) cg = LSE_MAX_GROUPS;
    const int cols_per_group = (nblocks + cg - 1) / cg * CBK;
    cg = (M + cols_per_group - 1) / cols_per_group;
    float* part = ws + 2 * M + 8;
    NSID_LAUNCH(ntxent_lse_kernel, dim3(row_tiles, cg), dim3(256), bytes, s, z_i, z_j, M, d, tau, cols_per_group, lse,
                rowloss, part);
    if (cg > 1) NSID_LAUNCH(ntxent_lse_merge_kernel, dim3((M + 255) / 256), dim3(256), 0, s, part, cg, M, lse, rowloss);
  }
  NSID_LAUNCH(ntxent_loss_kernel, dim3(1), dim3(256), 0, s, rowloss, 2 * p0, 2 * np, M, loss_out);
  if (dz_i != nullptr) {
    const int row0 = 2 * p0, nrows = 2 * np;
    const int row_tiles = (nrows + RB - 1) / RB, nblocks = (M + CBK - 1) / CBK;
    int cg = row_tiles >= 128 ? 1 : (256 + row_tiles - 1) / row_tiles;       // ~256 workgroups
    if (cg > nblocks) cg = nblocks;
    const int cols_per_group = (nblocks + cg - 1) / cg * CBK;
    cg = (M + cols_per_group - 1) / cols_per_group;
    if (cg > 1) {          // partial rows are accumulated atomically: zero the outputs first (a memset node when captured)
      if (hipMemsetAsync(dz_i, 0, (size_t)np * d * sizeof(float), s) != hipSuccess ||
          hipMemsetAsync(dz_j, 0, (size_t)np * d * sizeof(float), s) != hipSuccess)
        return NSID_ELAUNCH;
    }
    dim3 grid(row_tiles, cg);
#define NSID_NTX_CASE(DTV)                                                                                   \
  case DTV:                                                                                                  \
    if (raise_lds(ntxent_grad_kernel<DTV>, bytes) != NSID_OK) return NSID_ELAUNCH;                           \
    NSID_LAUNCH((ntxent_grad_kernel<DTV>), grid, dim3(256), bytes, s, z_i, z_j, M, d, tau, lse, row0, \
                       nrows, cols_per_group, dz_i, dz_j);                                                                   \
    break;
    switch (d / 16) {
      NSID_NTX_CASE(1) NSID_NTX_CASE(2) NSID_NTX_CASE(4) NSID_NTX_CASE(8) NSID_NTX_CASE(16)
      default: return NSID_EINVAL;
    }
#undef NSID_NTX_CASE
  }
  return nsid_launch_status();
}
