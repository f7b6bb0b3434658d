// Dilated kNN graph of one clip per workgroup: the whole clip (N nodes x C channels, 64 KB at every stage of
// the 't' encoder) is staged once into LDS, BatchNorm-applied and L2-normalised there, the N x N distance matrix
// is produced 16 rows at a time by fp32 MFMA straight from LDS and consumed by the top-k selection without ever
// reaching HBM.  Algorithmic HBM traffic = read N*C*4 B, write N*k*4 B per clip.
#include <cstdlib>
#include "nsid_common.h"
#include <type_traits>

namespace {

// F.normalize divides every channel of a row by the same norm: 16 384 IEEE divisions per clip (~10 instructions each, three
// of them quarter rate) were 6.5 us of a 20 us kernel. One IEEE reciprocal per row, then per element the product and ONE
// residual correction: q = v*r, q += (v - q*d)*r with fused multiply-adds. With r the correctly rounded 1/d this is the
// correctly rounded quotient v/d (Markstein's theorem; the lone exception, a divisor whose significand is all ones, is off
// by at most one ulp). |v| <= d here, so nothing overflows.
__device__ __forceinline__ float div_shared(float v, float d, float r) {
  const float q = v * r;
  return fmaf(fmaf(-q, d, v), r, q);
}


constexpr int KNN_WAVES = 4;       // waves of the strip kernel (one 16-row distance strip each per pass) when the clip leaves room for them

__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// wave-wide unsigned minimum with DPP only (no LDS permutes): xor-1, xor-2 inside quads, mirror inside 8 and 16 lanes,
// then row broadcasts 15 / 31; lane 63 ends with the minimum of all 64 lanes and is read back as a scalar.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  unsigned t;
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false); v = v < t ? v : t;    // quad_perm [1,0,3,2]
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false); v = v < t ? v : t;    // quad_perm [2,3,0,1]
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); v = v < t ? v : t;   // row_half_mirror
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); v = v < t ? v : t;   // row_mirror
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false); v = v < t ? v : t;   // row_bcast:15
  t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false); v = v < t ? v : t;   // row_bcast:31
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <typename T>
__global__ __launch_bounds__(256) void knn_kernel(const T* __restrict__ r, long ldr,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int N, int C, int k,
                                                          int dilation, int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LD = C + 4;                 // feature row stride (keeps 16-B alignment, staggers banks)
  const int SLD = N + 4;                // distance strip row stride
  float* yn = smem;                     // [N][LD]
  float* sq = yn + (long)N * LD;        // [N]
  float* strips = sq + N;               // [KNN_WAVES][16][SLD]

  const int b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // the number of waves (= distance strips in flight) comes from the launch: 4 by default, fewer when the fp32 clip + 4 strips exceed the
  // 160 KB of LDS (N * C = 24 576 at encoder size 'm': 102 KB of features + 4 x 16.6 KB of strips; two strips fit)
  const int KNN_WAVES = blockDim.x >> 6, KNN_THREADS = blockDim.x;
  const T* src = r + (long)b * N * ldr;
  constexpr int NV = Chunk<T>::N;
  const int CV = C / NV;

  // ---- phase 1a: stage y = scale*r + shift into LDS as fp32 (coalesced 16-byte chunks of either storage type)
  for (int q = t; q < N * CV; q += KNN_THREADS) {
    const int n = q / CV, c = (q % CV) * NV;
    float v[NV];
    Chunk<T>::load(src + (long)n * ldr + c, v);
    if (scale != nullptr) {
      float sc[NV], sh[NV];
      load_channels<NV>(scale, c, sc);
      load_channels<NV>(shift, c, sh);
#pragma unroll
      for (int e = 0; e < NV; ++e) v[e] = sc[e] * v[e] + sh[e];
    }
#pragma unroll
    for (int e = 0; e < NV; e += 4) *reinterpret_cast<f32x4*>(yn + n * LD + c + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
  }
  __syncthreads();
  // ---- phase 1b: F.normalize(p=2, dim=channels, eps=1e-12), then |y^|^2 as the reference recomputes it
  for (int n = wave; n < N; n += KNN_WAVES) {
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = yn[n * LD + c]; ss += v * v; }
    ss = wave_sum(ss);
    const float denom = fmaxf(sqrtf(ss), 1e-12f), rden = 1.f / denom;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float v = div_shared(yn[n * LD + c], denom, rden);
      yn[n * LD + c] = v;
      s2 += v * v;
    }
    s2 = wave_sum(s2);
    if (lane == 0) sq[n] = s2;
  }
  __syncthreads();

  const int lr = lane & 15, rq = lane >> 4;
  const int NS = N >> 4;                 // 16-row strips
  const int kd = k * dilation;
  float* strip = strips + wave * 16 * SLD;
  const int NE = (N + 63) >> 6;          // distance values per lane in the selection (<= 4)

  for (int sb = 0; sb < NS; sb += KNN_WAVES) {
    const int s = sb + wave;
    const bool valid = s < NS;           // wave-uniform
    if (valid) {
      // ---- phase 2: D[16 rows of strip s][all N] = |a|^2 - 2 a.b + |b|^2, two column tiles at a time
      const float* arow = yn + (16 * s + lr) * LD + 4 * rq;
      for (int tn = 0; tn < NS; tn += 2) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* b0 = yn + (16 * tn + lr) * LD + 4 * rq;
        const float* b1 = b0 + 16 * LD;
        for (int ch = 0; ch < C; ch += 16) {
          const f32x4 fa = *reinterpret_cast<const f32x4*>(arow + ch);
          const f32x4 f0 = *reinterpret_cast<const f32x4*>(b0 + ch);
          const f32x4 f1 = *reinterpret_cast<const f32x4*>(b1 + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], f0[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], f1[e], acc1, 0, 0, 0);
          }
        }
        // C/D layout: column (node j) = lane&15, row (node i) = 4*(lane>>4) + reg
        const float sj0 = sq[16 * tn + lr], sj1 = sq[16 * tn + 16 + lr];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float si = sq[16 * s + 4 * rq + e];
          strip[(4 * rq + e) * SLD + 16 * tn + lr] = (si + (-2.f * acc0[e])) + sj0;
          strip[(4 * rq + e) * SLD + 16 * tn + 16 + lr] = (si + (-2.f * acc1[e])) + sj1;
        }
      }
    }
    __syncthreads();
    if (valid) {
      // ---- phase 3: k*dilation rounds of arg-min per row. Each lane owns up to 4 distances (node lane + 64e) as
      // order-preserving 32-bit keys plus a live bit; a round is: lane-local min, DPP wave min, then ballots pick the
      // lowest node id among the lanes that hold that minimum (ties -> lower index), and that slot is retired.
      // Emitted ids are always < N (padding slots are never live), also when the distances are NaN.
      for (int i = 0; i < 16; ++i) {
        unsigned key[4];
        bool live[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = lane + 64 * e;
          live[e] = e < NE && j < N;
          key[e] = live[e] ? orderable(strip[i * SLD + j]) : 0xFFFFFFFFu;
        }
        int32_t* out = idx + ((long)b * N + 16 * s + i) * k;
        for (int round = 0; round < kd; ++round) {
          unsigned loc = 0xFFFFFFFFu;
#pragma unroll
          for (int e = 0; e < 4; ++e) loc = (live[e] && key[e] < loc) ? key[e] : loc;
          const unsigned m = wave_min_u32(loc);
          int j = 0;
          bool found = false;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned long long hit = __ballot(live[e] && key[e] == m);
            if (!found && hit != 0ull) {          // wave-uniform
              found = true;
              j = 64 * e + (__ffsll((long long)hit) - 1);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (lane + 64 * e == j) live[e] = false;
          if (lane == 0 && (round % dilation) == 0) out[round / dilation] = j < N ? j : N - 1;
        }
      }
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Graphs beyond 256 nodes (a cfg with more patches than grafp.yaml's 64 x 128 / (4 x 8): e.g. the literal 256-mel input, 1 024 / 512
// nodes at the first two stages — encoder/graph_encoder.py:144 takes any N). A clip's fp32 features no longer fit the LDS beside a
// distance strip, so a workgroup takes ONE 16-row strip of one clip and reads the column nodes' features from L2 as it goes:
//   phase 0  every node's norm, reciprocal and |y^|^2 (the same expressions and summation order as knn_kernel: lane = channel mod 64,
//            wave_sum) — each strip's workgroup repeats this for the whole clip (N / 16 times redundant, one clip read each: a
//            correctness path for configurations the timed ones never reach, not a tuned one);
//   phase 1  the strip's own 16 rows normalised into LDS;
//   phase 2  D[16][N] by the exact-fp32 MFMA of knn_kernel, the column operand normalised on its way from global memory
//            (div_shared with the node's norm from phase 0), the four waves taking column-tile pairs in turn;
//   phase 3  k * dilation rounds of arg-min per row straight on the LDS strip (ties -> the lower node id, as knn_kernel), four rows
//            per wave. Ids and distances are those of knn_kernel on the same input (tests/test_ops_gpu.py compares them where both run).
// LDS: 16 (C + 4) + 3 N + 16 (N + 4) floats (82 KB at N = 1 024, C = 64).
template <typename T>
__global__ __launch_bounds__(256) void knn_big_kernel(const T* __restrict__ r, long ldr, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int N, int C, int k, int dilation,
                                                      int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LD = C + 4, SLD = N + 4;
  float* arows = smem;                 // [16][LD]
  float* den = arows + 16 * LD;        // [N] max(|y|, 1e-12)
  float* rdn = den + N;                // [N] 1 / den
  float* sq = rdn + N;                 // [N] |y^|^2
  float* strip = sq + N;               // [16][SLD]
  const int s = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 15, rq = lane >> 4;
  const T* src = r + (long)b * N * ldr;
  auto feat = [&](int n, int c) -> float {
    float v = (float)src[(long)n * ldr + c];
    if (scale != nullptr) v = scale[c] * v + shift[c];
    return v;
  };

  for (int n = wave; n < N; n += 4) {
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = feat(n, c); ss += v * v; }
    ss = wave_sum(ss);
    const float denom = fmaxf(sqrtf(ss), 1e-12f), rden = 1.f / denom;
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = div_shared(feat(n, c), denom, rden); s2 += v * v; }
    s2 = wave_sum(s2);
    if (lane == 0) { den[n] = denom; rdn[n] = rden; sq[n] = s2; }
  }
  __syncthreads();
  for (int q = t; q < 16 * C; q += 256) {
    const int i = q / C, c = q - i * C, n = 16 * s + i;
    arows[i * LD + c] = div_shared(feat(n, c), den[n], rdn[n]);
  }
  __syncthreads();

  const int NS = N >> 4;
  const float* arow = arows + lr * LD + 4 * rq;
  for (int tn = 2 * wave; tn < NS; tn += 8) {
    const int n0 = 16 * tn + lr, n1 = n0 + 16;
    const float d0 = den[n0], r0 = rdn[n0], d1 = den[n1], r1 = rdn[n1];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < C; ch += 16) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(arow + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = ch + 4 * rq + e;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], div_shared(feat(n0, c), d0, r0), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], div_shared(feat(n1, c), d1, r1), acc1, 0, 0, 0);
      }
    }
    const float sj0 = sq[n0], sj1 = sq[n1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float si = sq[16 * s + 4 * rq + e];
      strip[(4 * rq + e) * SLD + 16 * tn + lr] = (si + (-2.f * acc0[e])) + sj0;
      strip[(4 * rq + e) * SLD + 16 * tn + 16 + lr] = (si + (-2.f * acc1[e])) + sj1;
    }
  }
  __syncthreads();

  const int kd = k * dilation;
  for (int i = wave; i < 16; i += 4) {
    float* row = strip + i * SLD;
    int32_t* out = idx + ((long)b * N + 16 * s + i) * k;
    for (int round = 0; round < kd; ++round) {
      unsigned loc = 0xFFFFFFFFu;
      int at = 0x7FFFFFFF;
      for (int j = lane; j < N; j += 64) {                 // the lane's columns in increasing order: strict < keeps the lower id
        const unsigned key = orderable(row[j]);
        if (key < loc) { loc = key; at = j; }
      }
      const unsigned m = wave_min_u32(loc);
      const int j = (int)wave_min_u32(loc == m ? (unsigned)at : 0x7FFFFFFFu);
      if (j < N && (j & 63) == lane) row[j] = __builtin_nanf("");          // retired: a NaN's key is above every finite distance and +inf
      if (lane == 0 && (round % dilation) == 0) out[round / dilation] = j < N ? j : N - 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fast path (k*dilation <= 8, C a multiple of 64): 8 waves per clip, no distance strips in LDS, no wave-wide reductions.
// The MFMA is issued with the operands swapped (A = the COLUMN nodes' rows, B = the ROW nodes' rows), so the C/D layout
// hands lane (lr, rq) the distances D[i = i0+lr][j = j0 + 4*rq + e]: every row of the 16-row tile is spread over the 4
// lanes lr, lr+16, lr+32, lr+48, each of which sees a quarter of the columns in increasing order and keeps its own
// sorted top-KD list in registers (branch-free insertion, ties keep the lower index). The four lists of a row are merged
// with two butterfly exchanges (ds_bpermute), lexicographic on (distance, index). Since round 3 the distance products of this
// path, of knn_rank_kernel and of knn_sel_kernel are the split-fp16 three-term products described below (22 significant bits per
// factor, fp32 accumulation), in BOTH precision modes; only the general strip kernel above still runs the exact-fp32 MFMA. The two
// agree to ~3e-7 absolute on unit-norm rows (tests/test_ops_gpu.py::test_knn_split_fp16_distance_error_bound holds the split
// product to an fp64 evaluation), so the neighbour SETS agree outside near-ties; the distances are NOT bit-identical.
// Small graphs (N < 128) have fewer than 8 row tiles: the column tiles of a row tile are then split over 2 waves and the
// second wave's lists reach the first through a small LDS buffer.
constexpr int KNN2_THREADS = 512;
constexpr int KNN2_WAVES = 8;

template <int KD>
struct TopList {
  float key[KD];
  int id[KD];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int p = 0; p < KD; ++p) { key[p] = __builtin_inff(); id[p] = 0; }
  }
  // candidates arrive in increasing index order: strict < keeps the earlier (lower) index on ties; NaN never enters
  __device__ __forceinline__ void push_ordered(float d, int j) {
    bool lt[KD];
#pragma unroll
    for (int p = 0; p < KD; ++p) lt[p] = d < key[p];
#pragma unroll
    for (int p = KD - 1; p > 0; --p) {
      key[p] = lt[p] ? (lt[p - 1] ? key[p - 1] : d) : key[p];
      id[p] = lt[p] ? (lt[p - 1] ? id[p - 1] : j) : id[p];
    }
    key[0] = lt[0] ? d : key[0];
    id[0] = lt[0] ? j : id[0];
  }
  // arbitrary arrival order: (distance, index) lexicographic
  __device__ __forceinline__ void push_any(float d, int j) {
    bool lt[KD];
#pragma unroll
    for (int p = 0; p < KD; ++p) lt[p] = d < key[p] || (d == key[p] && j < id[p]);
#pragma unroll
    for (int p = KD - 1; p > 0; --p) {
      key[p] = lt[p] ? (lt[p - 1] ? key[p - 1] : d) : key[p];
      id[p] = lt[p] ? (lt[p - 1] ? id[p - 1] : j) : id[p];
    }
    key[0] = lt[0] ? d : key[0];
    id[0] = lt[0] ? j : id[0];
  }
};

// sum over the 16 lanes of a DPP row (quad xor-1, xor-2, half mirror, mirror): every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

// debug timeline, as in gemm.hip: {start, features staged, normalised, end} on the 100 MHz clock per workgroup
__device__ unsigned long long* g_knn_trace = nullptr;

// ---- split-fp16 distance products (round 3) -----------------------------------------------------------------------------------
// The f32-input MFMA runs at the f32 VECTOR rate (64 flop/clk/SIMD, 1/16 of the 16-bit rate), and in knn2_kernel it competes with
// the top-list insertions for the same issue cycles: 13.7 us of a 25.6 us clip at N = 256. A normalised feature y (|y| <= 1) is split
// into two fp16 values, a = fp16(y) and b = fp16(y - a) (the difference is exact in fp32): a + b carries 22 significant bits of y,
// and a product keeps three terms:   y_i y_j = a_i a_j + (a_i b_j + b_i a_j) + O(2^-22 |y_i y_j|).
// Measured against an fp64 evaluation on unit-norm rows (C = 64 ... 512): rms error 2.6e-8, max 1.5e-7 — the fp32 GEMM the reference
// runs on the CPU sits at 1.9e-8 / 2.8e-7 (its own summation order), i.e. the split product is as close to the true distances as the
// reference is; a bf16 split (8 bits per part) would need three parts and six terms for that, a two-part bf16 product is 1e-5 off.
// Three v_mfma_f32_16x16x32_f16 per 32 channels (48 cycles) replace eight v_mfma_f32_16x16x4_f32 (256 cycles); the leading term and
// the two small ones go to separate fp32 accumulators and are added once per tile. (fp16's narrow exponent costs nothing here: below
// 6e-5 a value keeps an ABSOLUTE precision of 3e-8, which is what a distance between unit vectors needs.)
// LDS image: img[s][kc][n][8] fp16 (s = a, b; kc = 8-channel chunk; 16 bytes per (kc, n)): a fragment read (lane (lr, rq) takes
// node row lr, chunk 4*ks + rq) is conflict-free without padding — ds_read_b128 serves lanes {0-3,12-15,20-27} together, i.e. two
// COMPLEMENTARY halves of the 16 rows at chunks rq and rq + 1, 16 different 16-byte slots of a 256-byte bank row (N % 16 == 0).
// Two images of a clip are 64 KB at every stage (N * C = 16 384): two workgroups per CU, as with the fp32 image before.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ char* knn_img(char* img, int s, int kc, int n, int KC, int N) {
  return img + ((((long)s * KC + kc) * N + n) << 4);
}

template <typename T> __device__ __forceinline__ void knn_load8(const T* p, float* v);
template <> __device__ __forceinline__ void knn_load8<float>(const float* p, float* v) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void knn_load8<__bf16>(const __bf16* p, float* v) {
  const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
}

// Stage + BatchNorm-apply + F.normalize + two-way fp16 split of one clip, straight from global memory (no fp32 LDS image).
// A wave takes 8 rows per pass: lane (nl = lane & 7, kl = lane >> 3) owns the chunks kc = kl + 8 j (j < JN = C / 64) of row nl, so
// that the 8 lanes the LDS serves together write 8 consecutive rows of one chunk (128 contiguous bytes: conflict-free), and a row's
// sum of squares is three xor-shuffles (8, 16, 32) away. The quotient is the correctly rounded y / max(|y|, 1e-12) of F.normalize.
template <typename T, int JN>
__device__ __forceinline__ void knn_stage_split(const T* __restrict__ src, long ldr, const float* __restrict__ scale,
                                                const float* __restrict__ shift, int N, int nwaves, char* img, float* sq) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nl = lane & 7, kl = lane >> 3;
  constexpr int KC = 8 * JN;
  constexpr int RU = JN <= 2 ? 2 : 1;                      // row passes in flight (independent chains of ~2 000 cycles each)
  for (int n0 = wave * 8; n0 < N; n0 += RU * nwaves * 8) {
    float v[RU][JN][8];
    int nn[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int n = n0 + u * nwaves * 8 + nl;
      nn[u] = n < N ? n : -1;
      const T* rowp = src + (long)(n < N ? n : N - 1) * ldr;
#pragma unroll
      for (int j = 0; j < JN; ++j) knn_load8<T>(rowp + (kl + 8 * j) * 8, v[u][j]);
    }
    if (scale != nullptr) {
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        float sc[8], sh[8];
        load_channels<8>(scale, (kl + 8 * j) * 8, sc);
        load_channels<8>(shift, (kl + 8 * j) * 8, sh);
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[u][j][e] = sc[e] * v[u][j][e] + sh[e];
      }
    }
    float ss[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      ss[u] = 0.f;
#pragma unroll
      for (int j = 0; j < JN; ++j)
        ss[u] += ((v[u][j][0] * v[u][j][0] + v[u][j][1] * v[u][j][1]) + (v[u][j][2] * v[u][j][2] + v[u][j][3] * v[u][j][3])) +
                 ((v[u][j][4] * v[u][j][4] + v[u][j][5] * v[u][j][5]) + (v[u][j][6] * v[u][j][6] + v[u][j][7] * v[u][j][7]));
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int u = 0; u < RU; ++u) ss[u] += __shfl_xor(ss[u], o, 64);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const float denom = fmaxf(sqrtf(ss[u]), 1e-12f), rden = 1.f / denom;
      ss[u] = 0.f;
#pragma unroll
      for (int j = 0; j < JN; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[u][j][e] = div_shared(v[u][j][e], denom, rden);
        ss[u] += ((v[u][j][0] * v[u][j][0] + v[u][j][1] * v[u][j][1]) + (v[u][j][2] * v[u][j][2] + v[u][j][3] * v[u][j][3])) +
                 ((v[u][j][4] * v[u][j][4] + v[u][j][5] * v[u][j][5]) + (v[u][j][6] * v[u][j][6] + v[u][j][7] * v[u][j][7]));
      }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int u = 0; u < RU; ++u) ss[u] += __shfl_xor(ss[u], o, 64);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (nn[u] < 0) continue;
      if (kl == 0) sq[nn[u]] = ss[u];
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        f16x8 a, bb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float y = v[u][j][e];
          a[e] = (_Float16)y;                            // round to nearest even (v_cvt_f16_f32)
          bb[e] = (_Float16)(y - (float)a[e]);           // the difference is exact in fp32
        }
        *reinterpret_cast<f16x8*>(knn_img(img, 0, kl + 8 * j, nn[u], KC, N)) = a;
        *reinterpret_cast<f16x8*>(knn_img(img, 1, kl + 8 * j, nn[u], KC, N)) = bb;
      }
    }
  }
}

// Wide rows (C = 256 / 512: 4 / 8 chunks of 8 channels per lane). Holding a row's 32 / 64 affine-applied fp32 values across the two
// reductions spilled under the 128-register budget of the two-workgroups-per-CU kernels (round 3: 16-28 VGPRs to scratch in every
// instantiation — the C switch below inlines all four widths — reloaded once per 8-row pass). Here the row is walked TWICE from
// memory, one chunk live at a time: pass A accumulates the sum of squares, pass B re-reads the chunk (an L1 / L2 hit: the same 512 B /
// 1 KB the lane group fetched a moment ago), normalises, splits and stores. Same expressions in the same order as the narrow form,
// so the images are bit-identical. The empty asm between the passes keeps the compiler from merging the two reads back into
// registers.
template <typename T, int JN>
__device__ __forceinline__ void knn_stage_split_wide(const T* __restrict__ src, long ldr, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int N, int nwaves, char* img, float* sq) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nl = lane & 7, kl = lane >> 3;
  constexpr int KC = 8 * JN;
  for (int n0 = wave * 8; n0 < N; n0 += nwaves * 8) {
    const int n = n0 + nl;
    const int nn = n < N ? n : -1;
    const T* rowp = src + (long)(n < N ? n : N - 1) * ldr;
    float ss = 0.f;
#pragma unroll 2
    for (int j = 0; j < JN; ++j) {
      float v[8];
      knn_load8<T>(rowp + (kl + 8 * j) * 8, v);
      if (scale != nullptr) {
        float sc[8], sh[8];
        load_channels<8>(scale, (kl + 8 * j) * 8, sc);
        load_channels<8>(shift, (kl + 8 * j) * 8, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sc[e] * v[e] + sh[e];
      }
      ss += ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) ss += __shfl_xor(ss, o, 64);
    const float denom = fmaxf(sqrtf(ss), 1e-12f), rden = 1.f / denom;
    asm volatile("" ::: "memory");
    ss = 0.f;
#pragma unroll 2
    for (int j = 0; j < JN; ++j) {
      float v[8];
      knn_load8<T>(rowp + (kl + 8 * j) * 8, v);
      if (scale != nullptr) {
        float sc[8], sh[8];
        load_channels<8>(scale, (kl + 8 * j) * 8, sc);
        load_channels<8>(shift, (kl + 8 * j) * 8, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sc[e] * v[e] + sh[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = div_shared(v[e], denom, rden);
      ss += ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
      if (nn >= 0) {
        f16x8 a, bb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a[e] = (_Float16)v[e];
          bb[e] = (_Float16)(v[e] - (float)a[e]);
        }
        *reinterpret_cast<f16x8*>(knn_img(img, 0, kl + 8 * j, nn, KC, N)) = a;
        *reinterpret_cast<f16x8*>(knn_img(img, 1, kl + 8 * j, nn, KC, N)) = bb;
      }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) ss += __shfl_xor(ss, o, 64);
    if (nn >= 0 && kl == 0) sq[nn] = ss;
  }
}

// RAW (round 5, forward-only extraction): the features are the STORED bf16 values themselves (eval mode: the BatchNorm in front is folded
// into its conv, no affine on the load). bf16 x bf16 products are exact in fp32, so ONE bf16 MFMA pass on the raw features gives
// y_i . y_j to fp32 accumulation accuracy, and the normalisation becomes two fp32 factors per distance:
//     y^_i . y^_j = (y_i . y_j) / (|y_i| |y_j|),        D_ij = (|y^_i|^2 - 2 y^_i . y^_j) + |y^_j|^2      (torch_edge.py:281-284, 16-18)
// -- a third of the matrix work of the two-part fp16 split, half its LDS image, and no per-element divide / split in the staging pass
// (5-10 us of a clip's 13-27 us). Error against fp64 on unit-norm rows: that of one fp32 accumulation chain (<= 1.5e-7 |y_i||y_j|),
// the same order as the split product's and as the reference's own fp32 GEMM.
template <int JN>
__device__ __forceinline__ void knn_stage_raw(const __bf16* __restrict__ src, long ldr, int N, int nwaves, char* img, float* sq, float* inv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nl = lane & 7, kl = lane >> 3;
  constexpr int KC = 8 * JN;
  constexpr int RU = JN <= 2 ? 2 : 1;
  for (int n0 = wave * 8; n0 < N; n0 += RU * nwaves * 8) {
    bf16x8 raw[RU][JN];
    int nn[RU];
    float ss[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int n = n0 + u * nwaves * 8 + nl;
      nn[u] = n < N ? n : -1;
      const __bf16* rowp = src + (long)(n < N ? n : N - 1) * ldr;
#pragma unroll
      for (int j = 0; j < JN; ++j) raw[u][j] = *reinterpret_cast<const bf16x8*>(rowp + (kl + 8 * j) * 8);
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      ss[u] = 0.f;
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)raw[u][j][e];
        ss[u] += ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
      }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
      for (int u = 0; u < RU; ++u) ss[u] += __shfl_xor(ss[u], o, 64);
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (nn[u] < 0) continue;
      if (kl == 0) {
        const float rn = 1.f / fmaxf(sqrtf(ss[u]), 1e-12f);            // F.normalize's max(|y|, eps)
        inv[nn[u]] = rn;
        sq[nn[u]] = (ss[u] * rn) * rn;                                 // |y^|^2: 1 up to rounding, 0 for an all-zero row
      }
#pragma unroll
      for (int j = 0; j < JN; ++j) *reinterpret_cast<bf16x8*>(knn_img(img, 0, kl + 8 * j, nn[u], KC, N)) = raw[u][j];
    }
  }
}

template <typename T>
__device__ __forceinline__ void knn_stage_split_any(const T* src, long ldr, const float* scale, const float* shift, int N, int C,
                                                    int nwaves, char* img, float* sq) {
  switch (C) {            // uniform; C is 64, 128, 256 or 512 on these paths (host-checked)
    case 64: knn_stage_split<T, 1>(src, ldr, scale, shift, N, nwaves, img, sq); break;
    case 128: knn_stage_split<T, 2>(src, ldr, scale, shift, N, nwaves, img, sq); break;
    case 256: knn_stage_split_wide<T, 4>(src, ldr, scale, shift, N, nwaves, img, sq); break;
    default: knn_stage_split_wide<T, 8>(src, ldr, scale, shift, N, nwaves, img, sq); break;
  }
}

// one 32-channel step of a 16x16 distance tile: the two fragments of one node set (x) against those of the other (y); the MFMA's A
// operand indexes the tile's ROWS (C/D layout: row = 4*(lane>>4) + reg), its B operand the COLUMNS (lane & 15)
struct KnnFrag { f16x8 a, b; };
__device__ __forceinline__ KnnFrag knn_frag(const char* img, int kc, int n, int KC, int N) {
  KnnFrag f;
  f.a = *reinterpret_cast<const f16x8*>(knn_img(const_cast<char*>(img), 0, kc, n, KC, N));
  f.b = *reinterpret_cast<const f16x8*>(knn_img(const_cast<char*>(img), 1, kc, n, KC, N));
  return f;
}
__device__ __forceinline__ void knn_mfma3(const KnnFrag& x, const KnnFrag& y, f32x4& lead, f32x4& corr) {
  lead = __builtin_amdgcn_mfma_f32_16x16x32_f16(x.a, y.a, lead, 0, 0, 0);
  corr = __builtin_amdgcn_mfma_f32_16x16x32_f16(x.a, y.b, corr, 0, 0, 0);
  corr = __builtin_amdgcn_mfma_f32_16x16x32_f16(x.b, y.a, corr, 0, 0, 0);
}

// NT = column tiles per MFMA pass (4 for N >= 128, 2 for N = 64, 1 for N = 32) is a template parameter: as a run-time
// value every MFMA sat behind its own scalar branch (tools/asm_profile.py: 187 branches, one MFMA per basic block).
// PF = read the fragments of the next 32 channels while the MFMAs of the current ones run (ten fragments live: 146 VGPRs, one
// workgroup per CU). Without it the kernel fits 128 VGPRs and TWO workgroups share a CU, which is what hides latency when a launch
// has more clips than CUs (fingerprint extraction: 2 048 clips per micro-batch); with one clip per CU (a training step) the
// prefetch wins. The launcher picks by the clip count.
template <typename T, int KD, int NT, bool PF, bool RAW = false>
__device__ __forceinline__ void knn2_body(const T* __restrict__ r, long ldr, const float* __restrict__ scale,
                                          const float* __restrict__ shift, int N, int C, int k, int dilation,
                                          int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(!RAW || (std::is_same<T, __bf16>::value && !PF), "the raw-feature form reads stored bf16 values, two workgroups per CU");
  char* img = reinterpret_cast<char*>(smem);          // [2][C/8][N][8] fp16: the two split images of the normalised features (RAW: one bf16 image)
  float* sq = smem + (RAW ? 1 : 2) * (N * C / 2);     // [N]
  float* inv = sq + N;                                // RAW: [N] 1 / max(|y|, 1e-12)
  float* xkey = sq + (RAW ? 2 : 1) * N;               // [4 row tiles][16 rows][KD] lists handed over by the second column group
  int* xid = reinterpret_cast<int*>(xkey + 4 * 16 * KD);
  unsigned long long* const trace = g_knn_trace;
  unsigned long long tt[3] = {0, 0, 0};
  if (trace) tt[0] = __builtin_amdgcn_s_memrealtime();

  const int b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const T* src = r + (long)b * N * ldr;
  const int KC = C >> 3;

  // ---- phase 1: y = scale*r + shift, F.normalize, |y^|^2, two-way fp16 split — one pass from global memory into the LDS images
  if constexpr (RAW) {
    switch (C) {            // uniform (host-checked: 64, 128, 256 or 512)
      case 64: knn_stage_raw<1>(src, ldr, N, KNN2_WAVES, img, sq, inv); break;
      case 128: knn_stage_raw<2>(src, ldr, N, KNN2_WAVES, img, sq, inv); break;
      case 256: knn_stage_raw<4>(src, ldr, N, KNN2_WAVES, img, sq, inv); break;
      default: knn_stage_raw<8>(src, ldr, N, KNN2_WAVES, img, sq, inv); break;
    }
  } else {
    knn_stage_split_any<T>(src, ldr, scale, shift, N, C, KNN2_WAVES, img, sq);
  }
  if (trace) tt[1] = __builtin_amdgcn_s_memrealtime();
  __syncthreads();

  if (trace) tt[2] = __builtin_amdgcn_s_memrealtime();
  const int lr = lane & 15, rq = lane >> 4;
  const int RT = N >> 4;                                  // row tiles == column tiles (2, 4, 8 or 16)
  const int G = RT >= KNN2_WAVES ? 1 : 2;                 // column groups per row tile
  const int CTG = RT / G;                                 // column tiles per group: 16, 8, 2 or 1 (a multiple of NT)
  const int cg = RT >= KNN2_WAVES ? 0 : wave / RT;
  const bool active = cg < G;                             // wave-uniform
  TopList<KD> top;

  for (int rt = (RT >= KNN2_WAVES ? wave : wave % RT); rt < RT && active; rt += KNN2_WAVES) {
    top.init();
    const int i = 16 * rt + lr;                           // this lane's row node
    const float si = sq[i];
    float m2ri = 0.f;
    if constexpr (RAW) m2ri = -2.f * inv[i];
    for (int ct0 = cg * CTG; ct0 < (cg + 1) * CTG; ct0 += NT) {
      f32x4 lead[NT], corr[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) lead[u] = corr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (RAW) {
        for (int kc = 0; kc < KC; kc += 4) {
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(knn_img(img, 0, kc + rq, i, KC, N));
#pragma unroll
          for (int u = 0; u < NT; ++u)
            lead[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                *reinterpret_cast<const bf16x8*>(knn_img(img, 0, kc + rq, 16 * (ct0 + u) + lr, KC, N)), fa, lead[u], 0, 0, 0);
        }
      } else if constexpr (PF) {
        KnnFrag fa = knn_frag(img, rq, i, KC, N), fb[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) fb[u] = knn_frag(img, rq, 16 * (ct0 + u) + lr, KC, N);
        for (int kc = 0; kc < KC; kc += 4) {
          const int kn = (kc + 4 < KC ? kc + 4 : kc) + rq;   // the last prefetch re-reads the current chunk (unused)
          const KnnFrag na = knn_frag(img, kn, i, KC, N);
          KnnFrag nb[NT];
#pragma unroll
          for (int u = 0; u < NT; ++u) nb[u] = knn_frag(img, kn, 16 * (ct0 + u) + lr, KC, N);
#pragma unroll
          for (int u = 0; u < NT; ++u) knn_mfma3(fb[u], fa, lead[u], corr[u]);     // A = column nodes, B = this lane's row node
          fa = na;
#pragma unroll
          for (int u = 0; u < NT; ++u) fb[u] = nb[u];
        }
      } else {
        for (int kc = 0; kc < KC; kc += 4) {
          const KnnFrag fa = knn_frag(img, kc + rq, i, KC, N);
#pragma unroll
          for (int u = 0; u < NT; ++u) knn_mfma3(knn_frag(img, kc + rq, 16 * (ct0 + u) + lr, KC, N), fa, lead[u], corr[u]);
        }
      }
      // lead + corr = y_j . y_i with j = 16*(ct0+u) + 4*rq + e, i = this lane's row: D = (|i|^2 - 2 i.j) + |j|^2
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int j0 = 16 * (ct0 + u) + 4 * rq;
        const f32x4 sj = *reinterpret_cast<const f32x4*>(sq + j0);
        if constexpr (RAW) {
          const f32x4 rj = *reinterpret_cast<const f32x4*>(inv + j0);
#pragma unroll
          for (int e = 0; e < 4; ++e) top.push_ordered(fmaf(m2ri, lead[u][e] * rj[e], si) + sj[e], j0 + e);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) top.push_ordered((si + (-2.f * (lead[u][e] + corr[u][e]))) + sj[e], j0 + e);
        }
      }
    }
    // ---- merge the four quarter lists of every row (lanes lr + 16q): two butterfly exchanges, after which all four
    // lanes hold the same top-KD of the union
#pragma unroll
    for (int step = 16; step <= 32; step <<= 1) {
      float ok[KD];
      int oi[KD];
#pragma unroll
      for (int p = 0; p < KD; ++p) { ok[p] = __shfl_xor(top.key[p], step, 64); oi[p] = __shfl_xor(top.id[p], step, 64); }
#pragma unroll
      for (int p = 0; p < KD; ++p) top.push_any(ok[p], oi[p]);
    }
    if (G > 1 && cg > 0) {                                // hand the list over to the wave that owns the row tile
      if (rq == 0) {
#pragma unroll
        for (int p = 0; p < KD; ++p) { xkey[(rt * 16 + lr) * KD + p] = top.key[p]; xid[(rt * 16 + lr) * KD + p] = top.id[p]; }
      }
    } else if (G == 1 && rq == 0) {
      int32_t* out = idx + ((long)b * N + i) * k;
#pragma unroll
      for (int p = 0; p < KD; ++p)
        if (p % dilation == 0 && p / dilation < k) out[p / dilation] = top.id[p];
    }
  }
  if (G > 1) {                                            // uniform over the workgroup
    __syncthreads();
    if (active && cg == 0) {
      const int rt = wave % RT;
#pragma unroll
      for (int p = 0; p < KD; ++p) top.push_any(xkey[(rt * 16 + lr) * KD + p], xid[(rt * 16 + lr) * KD + p]);
      if (rq == 0) {
        int32_t* out = idx + ((long)b * N + 16 * rt + lr) * k;
#pragma unroll
        for (int p = 0; p < KD; ++p)
          if (p % dilation == 0 && p / dilation < k) out[p / dilation] = top.id[p];
      }
    }
  }
  if (trace && threadIdx.x == 0) {
    trace[4 * blockIdx.x + 0] = tt[0];
    trace[4 * blockIdx.x + 1] = tt[1];
    trace[4 * blockIdx.x + 2] = tt[2];
    trace[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

template <typename T, int KD, int NT>
__global__ __launch_bounds__(KNN2_THREADS) void knn2_kernel(const T* __restrict__ r, long ldr,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int N, int C, int k,
                                                            int dilation, int32_t* __restrict__ idx) {
  knn2_body<T, KD, NT, true>(r, ldr, scale, shift, N, C, k, dilation, idx);
}
// the two-workgroups-per-CU form: four waves per SIMD = at most 128 VGPRs
template <typename T, int KD, int NT>
__global__ __launch_bounds__(KNN2_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void knn2_pair_kernel(const T* __restrict__ r, long ldr, const float* __restrict__ scale, const float* __restrict__ shift,
                      int N, int C, int k, int dilation, int32_t* __restrict__ idx) {
  knn2_body<T, KD, NT, false>(r, ldr, scale, shift, N, C, k, dilation, idx);
}

// the raw-feature form (stored bf16 features, no affine: forward-only extraction), two workgroups per CU
// WPE waves per SIMD = WPE / 2 workgroups per CU (35 KB of LDS each): the register budget follows (4: 128, 6: 80 VGPRs)
template <int KD, int NT, int WPE>
__global__ __launch_bounds__(KNN2_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void knn2_raw_kernel(const __bf16* __restrict__ r, long ldr, int N, int C, int k, int dilation, int32_t* __restrict__ idx) {
  knn2_body<__bf16, KD, NT, false, true>(r, ldr, nullptr, nullptr, N, C, k, dilation, idx);
}

// ------------------------------------------------------------------------------------------------------------------
// Large k*dilation on small graphs (N <= 128; by default the 64- and 32-node graphs, where k*d is most of the row — the deep
// configuration asks for the 54 nearest of 64 nodes — while 128 nodes take knn_sel_kernel, tuning key knn_sel_min_n): selection by
// RANK COUNTING. A wave computes a 16-row distance strip
// into LDS (4 MFMA accumulator chains), then every lane takes (row, j) pairs and counts the entries of that row that
// precede D[row][j] in (distance, index) order — N compares on LDS broadcasts, no dependent reductions, no rounds. The
// element of rank p*dilation is neighbour p. The strip kernel's k*d rounds of wave-wide arg-min took 330 us per call here.
// NT = column tiles per MFMA pass (4, or 2 for N = 32): a template parameter for the same reason as in knn2_kernel.
template <typename T, int NT>
__global__ __launch_bounds__(KNN2_THREADS) void knn_rank_kernel(const T* __restrict__ r, long ldr,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int N, int C, int k,
                                                                int dilation, int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int SLD = N + 4;
  char* img = reinterpret_cast<char*>(smem);   // [2][C/8][N][8] fp16 split images (see knn2_kernel)
  float* sq = smem + 2 * (N * C / 2);          // [N]
  float* strips = sq + N;                      // [8 waves][16][SLD]
  const int b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const T* src = r + (long)b * N * ldr;
  const int KC = C >> 3;
  knn_stage_split_any<T>(src, ldr, scale, shift, N, C, KNN2_WAVES, img, sq);
  __syncthreads();

  const int lr = lane & 15, rq = lane >> 4;
  const int NS = N >> 4;
  const int kd = k * dilation;
  if (2 * NS <= KNN2_WAVES) {
    // Few strips (N <= 64: the 64- and 32-node graphs of the deep plan, 4 / 2 strips for 8 waves): TWO waves per strip instead of idle
    // ones -- each takes half of the strip's column tiles in the distance pass and half of its (row, j) pairs in the rank-counting pass,
    // one workgroup barrier between the passes (round 6: the kernel is one clip per CU, i.e. pure latency; 33.5 us per launch at
    // B = 256, N = 64 with four of eight waves working).
    const int s = wave % NS, h = wave / NS;            // waves >= 2 NS stay idle (N = 32)
    const bool act = wave < 2 * NS;                    // wave-uniform
    float* strip2 = strips + s * 16 * SLD;
    if (act) {
      const int t0 = h * (NS / 2), t1 = t0 + NS / 2;   // NT divides NS / 2 (host: NT = NS / 2)
      for (int tn = t0; tn < t1; tn += NT) {
        f32x4 lead[NT], corr[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) lead[u] = corr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        KnnFrag fa = knn_frag(img, rq, 16 * s + lr, KC, N), fb[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) fb[u] = knn_frag(img, rq, 16 * (tn + u) + lr, KC, N);
        for (int kc = 0; kc < KC; kc += 4) {
          const int kn = (kc + 4 < KC ? kc + 4 : kc) + rq;
          const KnnFrag na = knn_frag(img, kn, 16 * s + lr, KC, N);
          KnnFrag nb[NT];
#pragma unroll
          for (int u = 0; u < NT; ++u) nb[u] = knn_frag(img, kn, 16 * (tn + u) + lr, KC, N);
#pragma unroll
          for (int u = 0; u < NT; ++u) knn_mfma3(fa, fb[u], lead[u], corr[u]);
          fa = na;
#pragma unroll
          for (int u = 0; u < NT; ++u) fb[u] = nb[u];
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const float sj = sq[16 * (tn + u) + lr];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float si = sq[16 * s + 4 * rq + e];
            strip2[(4 * rq + e) * SLD + 16 * (tn + u) + lr] = (si + (-2.f * (lead[u][e] + corr[u][e]))) + sj;      // the same expression as below
          }
        }
      }
    }
    __syncthreads();
    if (act) {
      for (int p = lane + 64 * h; p < 16 * N; p += 128) {        // the two waves of a strip interleave its (row, j) pairs
        const int row = p / N, j = p % N;
        const float* drow = strip2 + row * SLD;
        const float dj = drow[j];
        int rank = 0, same = 0;
        for (int m = 0; m < N; m += 4) {
          const f32x4 dm = *reinterpret_cast<const f32x4*>(drow + m);
#pragma unroll
          for (int e = 0; e < 4; ++e) { rank += dm[e] < dj ? 1 : 0; same += dm[e] == dj ? 1 : 0; }
        }
        if (__any(same > 1)) {                           // wave-uniform
          if (same > 1)
            for (int m = 0; m < j; ++m) rank += drow[m] == dj ? 1 : 0;
        }
        int32_t* out = idx + ((long)b * N + 16 * s + row) * k;
        if (dj != dj) {                                  // NaN distances: every entry ranks 0 — emit valid ids anyway
          if (j < k) out[j] = j;
        } else if (rank < kd && rank % dilation == 0) {
          out[rank / dilation] = j;
        }
      }
    }
    return;
  }
  float* strip = strips + wave * 16 * SLD;
  for (int s = wave; s < NS; s += KNN2_WAVES) {      // no workgroup barrier below: a wave owns its strip buffer
    for (int tn = 0; tn < NS; tn += NT) {
      f32x4 lead[NT], corr[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) lead[u] = corr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      // the fragments of the next 32 channels are read while the MFMAs of the current ones run
      KnnFrag fa = knn_frag(img, rq, 16 * s + lr, KC, N), fb[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) fb[u] = knn_frag(img, rq, 16 * (tn + u) + lr, KC, N);
      for (int kc = 0; kc < KC; kc += 4) {
        const int kn = (kc + 4 < KC ? kc + 4 : kc) + rq;
        const KnnFrag na = knn_frag(img, kn, 16 * s + lr, KC, N);
        KnnFrag nb[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) nb[u] = knn_frag(img, kn, 16 * (tn + u) + lr, KC, N);
#pragma unroll
        for (int u = 0; u < NT; ++u) knn_mfma3(fa, fb[u], lead[u], corr[u]);      // A = the strip's row nodes, B = column nodes
        fa = na;
#pragma unroll
        for (int u = 0; u < NT; ++u) fb[u] = nb[u];
      }
      // C/D layout: column (node j) = lane&15, row (node i) = 4*(lane>>4) + reg
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const float sj = sq[16 * (tn + u) + lr];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float si = sq[16 * s + 4 * rq + e];
          strip[(4 * rq + e) * SLD + 16 * (tn + u) + lr] = (si + (-2.f * (lead[u][e] + corr[u][e]))) + sj;
        }
      }
    }
    // rank counting over the strip (LDS operations of one wave are ordered: no barrier needed).
    // rank(j) = #{m : D[m] < D[j]} + #{m < j : D[m] == D[j]}. The tie term matters only for values that occur more than once in a
    // row (duplicated nodes): rare, and counting `==` beside `<` was half of the loop's instructions. Without ties the N ranks of
    // a row are a permutation of 0 .. N-1; tied entries all get the LOWER rank, so the row's rank sum falls short of N (N-1) / 2.
    // A row is visited by min(64, N) lanes at once, each holding N / 64 (or 1) of its entries: the lanes add their ranks up, and
    // only on a shortfall does the row run the equality pass.
    if (N < 128) {
      // small graphs: both counters in one pass (measured: the row-sum form below costs 4 % more here, its per-row reduction is not
      // amortised over 64 entries)
      for (int p = lane; p < 16 * N; p += 64) {
        const int row = p / N, j = p % N;
        const float* drow = strip + row * SLD;
        const float dj = drow[j];
        int rank = 0, same = 0;
        for (int m = 0; m < N; m += 4) {
          const f32x4 dm = *reinterpret_cast<const f32x4*>(drow + m);
#pragma unroll
          for (int e = 0; e < 4; ++e) { rank += dm[e] < dj ? 1 : 0; same += dm[e] == dj ? 1 : 0; }
        }
        if (__any(same > 1)) {                           // wave-uniform
          if (same > 1)
            for (int m = 0; m < j; ++m) rank += drow[m] == dj ? 1 : 0;
        }
        int32_t* out = idx + ((long)b * N + 16 * s + row) * k;
        if (dj != dj) {                                  // NaN distances: every entry ranks 0 — emit valid ids anyway
          if (j < k) out[j] = j;
        } else if (rank < kd && rank % dilation == 0) {
          out[rank / dilation] = j;
        }
      }
    } else
    {
      const int lprw = N < 64 ? N : 64;                  // lanes per row
      const int parts = N > 64 ? N >> 6 : 1;             // entries per lane and row (1 or 2: N <= 128)
      const int rpt = 64 / lprw;                         // rows per trip
      const int jb = lane % lprw;
      for (int row0 = 0; row0 < 16; row0 += rpt) {
        const int row = row0 + lane / lprw;
        const float* drow = strip + row * SLD;
        float dj[2];
        int rank[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < 2; ++q) dj[q] = drow[q < parts ? jb + 64 * q : jb];
        if (parts > 1) {                                 // uniform: one LDS broadcast feeds both of the lane's entries
          for (int m = 0; m < N; m += 4) {
            const f32x4 dm = *reinterpret_cast<const f32x4*>(drow + m);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              rank[0] += dm[e] < dj[0] ? 1 : 0;
              rank[1] += dm[e] < dj[1] ? 1 : 0;
            }
          }
        } else {
          for (int m = 0; m < N; m += 4) {
            const f32x4 dm = *reinterpret_cast<const f32x4*>(drow + m);
#pragma unroll
            for (int e = 0; e < 4; ++e) rank[0] += dm[e] < dj[0] ? 1 : 0;
          }
        }
        int sum = rank[0] + (parts > 1 ? rank[1] : 0);
        if (lprw == 64) {
          sum = (int)wave_sum((float)sum);               // DPP only, exact: the total is < 2^24
        } else {
          for (int o = 1; o < lprw; o <<= 1) sum += __shfl_xor(sum, o, 64);
        }
        const bool suspect = sum != N * (N - 1) / 2;     // also true for rows that hold NaN (harmless: NaN equals nothing)
        if (__any(suspect)) {                            // wave-uniform
          if (suspect) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              if (q >= parts) continue;
              const int j = jb + 64 * q;
              int same = 0;
              for (int m = 0; m < N; m += 4) {
                const f32x4 dm = *reinterpret_cast<const f32x4*>(drow + m);
#pragma unroll
                for (int e = 0; e < 4; ++e) same += dm[e] == dj[q] ? 1 : 0;
              }
              if (same > 1)
                for (int m = 0; m < j; ++m) rank[q] += drow[m] == dj[q] ? 1 : 0;
            }
          }
        }
        int32_t* out = idx + ((long)b * N + 16 * s + row) * k;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (q >= parts) continue;
          const int j = jb + 64 * q;
          if (dj[q] != dj[q]) {                          // NaN distances: every entry ranks 0 — emit valid ids anyway
            if (j < k) out[j] = j;
          } else if (rank[q] < kd && rank[q] % dilation == 0) {
            out[rank[q] / dilation] = j;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Large k*dilation on the 256-node graphs (deep configuration, stage 0: the 18 nearest of 256): THRESHOLD SELECT IN REGISTERS.
// A wave owns a 16-row distance strip and never writes it anywhere: the MFMA C/D layout leaves row 4*(lane>>4)+e of the strip in
// register e of the 16 accumulator tiles of ONE 16-lane group (lane&15 = column within a tile), i.e. a row's 256 distances are 16
// registers x 16 lanes, and the four groups of a wave work on four rows at once (e = 0..3 in turn):
//   1. the distances become order-preserving u32 keys (NaN -> largest);
//   2. T = a key with k*d <= #{key <= T} <= k*d + KSEL_SLACK, by bisection of the key space: one step is 16 compares per lane and a
//      4-step DPP sum over the group — no LDS, no scalar round trips (about a dozen steps for distances of unit vectors; the search
//      ends at the exact (k*d)-th smallest key when ties keep the count above the slack);
//   3. the entries <= T are compacted into the group's LDS slot as packed (key << 32 | column) words (DPP prefix sum);
//   4. each candidate's exact rank among the candidates is one 64-bit compare per other candidate on 16-byte LDS reads, eight
//      candidates per trip -> neighbour rank / dilation.
// No workgroup barrier after staging. History: k*d rounds of a dependent 64-lane arg-min (the strip kernel) took 480 us per call
// here; a strip in LDS + the 64 lane minima ranked against each other for a threshold 157 us (tools/knn_sel_trace.py: 5 500 of a
// row's 9 100 cycles went to those 64 compares on LDS broadcasts); the same with a wave-wide bisection on ballots 91 us.
constexpr int KSEL_SLACK = 3;             // surplus candidates accepted instead of another bisection step
constexpr int KSEL_CH = 8;                // candidates per ranking trip (four 16-byte reads)

__device__ __forceinline__ unsigned knn_key(float x) {
  const unsigned b = __float_as_uint(x + 0.f);                       // -0 -> +0: equal distances must get equal keys
  const unsigned key = b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
  return x != x ? 0xffffffffu : key;
}
__device__ __forceinline__ float knn_unkey(unsigned key) {          // the value a key stands for (inverse of knn_key on non-NaN keys)
  return __uint_as_float((key & 0x80000000u) ? key ^ 0x80000000u : ~key);
}
__device__ __forceinline__ int row16_sum(int v) {                    // every lane of a 16-lane row gets the row's total
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
  return v;
}
__device__ __forceinline__ unsigned row16_min(unsigned v) {          // minimum / maximum over a 16-lane row, in every lane
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = o < v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); v = o < v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = o < v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); v = o < v ? o : v;
  return v;
}
__device__ __forceinline__ unsigned row16_max(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = o > v ? o : v;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); v = o > v ? o : v;
  return v;
}
__device__ __forceinline__ int row16_scan(int v) {                   // inclusive prefix sum within a 16-lane row
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1, zeros shifted in
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  return v;
}

// TILES = column tiles of a strip = N / 16 (16: the 256-node graphs; 8: the 128-node graphs, where k*d = 36 of 128 still leaves
// rank counting 12x more compares than ranking the candidates)
template <typename T, int TILES>
__global__ __launch_bounds__(KNN2_THREADS) void knn_sel_kernel(const T* __restrict__ r, long ldr,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int N, int C, int k,
                                                               int dilation, int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int SLOT = 16 * TILES + 2 * KSEL_CH;            // candidates of a row + padding + the dump word
  char* img = reinterpret_cast<char*>(smem);         // [2][C/8][N][8] fp16 split images (see knn2_kernel)
  float* sq = smem + 2 * (N * C / 2);                // [N]
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(sq + N);      // [8 waves][4 groups][SLOT]
  const int b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const T* src = r + (long)b * N * ldr;
  const int KC = C >> 3;
  knn_stage_split_any<T>(src, ldr, scale, shift, N, C, KNN2_WAVES, img, sq);
  __syncthreads();

  const int lr = lane & 15, rq = lane >> 4;
  const int kd = k * dilation;
  const unsigned dinv = (65536u + dilation - 1) / dilation;          // rank / dilation == rank * dinv >> 16 for rank < 256
  unsigned long long* slot = slots + (wave * 4 + rq) * SLOT;
#ifdef NSID_KSEL_TRACE          // diagnosis build: cycles of wave 0 per section of the loop (tools/knn_sel_trace.py)
  unsigned long long ksel_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ksel_last = __builtin_amdgcn_s_memtime();
#define KSEL_MARK(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ksel_acc[i] += now_ - ksel_last; ksel_last = now_; }
#else
#define KSEL_MARK(i)
#endif
  for (int s = wave; s < TILES; s += KNN2_WAVES) {
    KSEL_MARK(0);
    unsigned key[TILES][4];
    // ---- phase A: the strip's 16 x N distances, up to four column tiles per pass, straight into keys
    constexpr int TP = TILES < 4 ? TILES : 4;
#pragma unroll
    for (int t0 = 0; t0 < TILES; t0 += TP) {
      f32x4 lead[TP], corr[TP];
#pragma unroll
      for (int u = 0; u < TP; ++u) lead[u] = corr[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < KC; kc += 4) {
        const KnnFrag fa = knn_frag(img, kc + rq, 16 * s + lr, KC, N);
        KnnFrag fb[TP];
#pragma unroll
        for (int u = 0; u < TP; ++u) fb[u] = knn_frag(img, kc + rq, 16 * (t0 + u) + lr, KC, N);
#pragma unroll
        for (int u = 0; u < TP; ++u) knn_mfma3(fa, fb[u], lead[u], corr[u]);      // A = the strip's row nodes, B = column nodes
      }
#pragma unroll
      for (int u = 0; u < TP; ++u) {
        const float sj = sq[16 * (t0 + u) + lr];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float si = sq[16 * s + 4 * rq + e];
          // 2 x as x + x, not -2.f * x: hipcc contracts the product form into v_pk_fma_f32 (pairs of tiles) with the literal 2.0 and
          // op_sel on the |y_i|^2 pair, and on MI355X / ROCm 7.2 that instruction intermittently (0.1-1 % of launches, same rows each
          // time, lanes 48-63 of the low half) dropped the |y_i|^2 term: distances exactly 1.0 too small in one tile of one row
          // (tools/knn_sel_repro.py; docs/experiments.md; guarded by tests/test_cabi.py). Same value either way: doubling is exact.
          const float x_ = lead[u][e] + corr[u][e];
          key[t0 + u][e] = knn_key((si - (x_ + x_)) + sj);
        }
      }
    }
#ifdef NSID_KSEL_DUMP       // diagnosis build: dump every key (tools/knn_sel_repro.py): [clip][row][column]
    if (g_knn_trace) {
      unsigned* dump = reinterpret_cast<unsigned*>(g_knn_trace);
#pragma unroll
      for (int u = 0; u < TILES; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) dump[((long)b * 16 * TILES + 16 * s + 4 * rq + e) * 16 * TILES + 16 * u + lr] = key[u][e];
    }
#endif
    KSEL_MARK(1);
    // ---- phase B: row 16 s + 4 rq + e in this lane's group
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int32_t* out = idx + ((long)b * N + 16 * s + 4 * rq + e) * k;
      // threshold search: clo = #{key <= lo} < kd <= #{key <= hi} = chi throughout. The bracket starts at the row's own smallest and
      // largest comparable key; steps alternate between INTERPOLATION (the key at which a locally uniform density would put rank
      // k*d + 1.5) and plain bisection (keeps the worst case at twice the bisection count whatever the density): 7-8 steps where
      // bisection of the whole key space took 13-14. Any threshold in the slack gives the same ids: the ranks below are exact.
      unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
      for (int u = 0; u < TILES; ++u) {
        kmin = key[u][e] < kmin ? key[u][e] : kmin;
        const unsigned kk = key[u][e] == 0xffffffffu ? 0u : key[u][e];        // NaN is not a threshold
        kmax = kk > kmax ? kk : kmax;
      }
      kmin = row16_min(kmin);
      kmax = row16_max(kmax);
      unsigned lo = kmin - 1u, hi = kmax;                      // kmin >= key(-inf) = 0x007fffff: no wrap
      int clo = 0, chi;
      {
        int c = 0;
#pragma unroll
        for (int u = 0; u < TILES; ++u) c += key[u][e] <= hi ? 1 : 0;
        chi = row16_sum(c);
      }
      if (chi < kd)                                            // NaN-poisoned row: fewer comparable entries than wanted
        for (int j = lr; j < k; j += 16) out[j] = j;
      bool act = chi > kd + KSEL_SLACK && hi > lo + 1u;
      for (int it = 0; __any(act); ++it) {
        const unsigned span = hi - lo;
        unsigned mid = lo + (span >> 1);
        if ((it & 1) == 0) {                                   // interpolate between the DISTANCES the two keys stand for
          const float want = ((float)(kd - clo) + 1.5f) * __builtin_amdgcn_rcpf((float)(chi - clo));
          const float vlo = knn_unkey(lo), vhi = knn_unkey(hi);
          unsigned guess = knn_key(vlo + (vhi - vlo) * want);
          guess = guess <= lo ? lo + 1u : guess;
          guess = guess >= hi ? hi - 1u : guess;
          mid = act ? guess : mid;                             // (span >= 2 while act)
        }
        int c = 0;
#pragma unroll
        for (int u = 0; u < TILES; ++u) c += key[u][e] <= mid ? 1 : 0;
        c = row16_sum(c);
        const bool down = act && c >= kd, up = act && c < kd;
        hi = down ? mid : hi;
        chi = down ? c : chi;
        lo = up ? mid : lo;
        clo = up ? c : clo;
        act = act && chi > kd + KSEL_SLACK && hi - lo > 1u;
      }
      KSEL_MARK(2);
      int n = 0;
#pragma unroll
      for (int u = 0; u < TILES; ++u) n += key[u][e] <= hi ? 1 : 0;
      int pos = row16_scan(n) - n;
#pragma unroll
      for (int u = 0; u < TILES; ++u) {
        const bool cand = key[u][e] <= hi;
        slot[cand ? pos : 16 * TILES + KSEL_CH] = ((unsigned long long)key[u][e] << 32) | (unsigned)(16 * u + lr);
        pos += cand ? 1 : 0;
      }
      if (lr < KSEL_CH) slot[chi + lr] = ~0ull;                // pad to whole trips: the largest word precedes nothing
      KSEL_MARK(3);
      for (int c0 = lr; c0 < chi; c0 += 16) {
        const unsigned long long mine = slot[c0];
        int rank = 0;
        for (int m = 0; m < chi; m += KSEL_CH) {
          typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
          u64x2 o[KSEL_CH / 2];
#pragma unroll
          for (int q = 0; q < KSEL_CH / 2; ++q) o[q] = *reinterpret_cast<const u64x2*>(slot + m + 2 * q);
#pragma unroll
          for (int q = 0; q < KSEL_CH / 2; ++q) rank += (o[q][0] < mine ? 1 : 0) + (o[q][1] < mine ? 1 : 0);
        }
        if (rank < kd) {
          const int q = (int)(((unsigned)rank * dinv) >> 16);
          if (q * dilation == rank) out[q] = (int)(unsigned)mine;
        }
      }
      KSEL_MARK(4);
    }
  }
#ifdef NSID_KSEL_TRACE
  if (g_knn_trace && t == 0)
    for (int i = 0; i < 8; ++i) g_knn_trace[8 * blockIdx.x + i] = ksel_acc[i];
#endif
}

template <typename T, int TILES>
int launch_knn_sel(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C, int k,
                   int dilation, int32_t* idx, hipStream_t s) {
  const size_t bytes = (size_t)2 * N * C * 2 + (size_t)N * sizeof(float) +
                       (size_t)KNN2_WAVES * 4 * (16 * TILES + 2 * KSEL_CH) * sizeof(unsigned long long);
  if (bytes > 160 * 1024 || N != 16 * TILES) return 1;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn_sel_kernel<T, TILES>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  NSID_LAUNCH((knn_sel_kernel<T, TILES>), dim3(B), dim3(KNN2_THREADS), bytes, s, static_cast<const T*>(r), (long)ldr, scale,
              shift, N, C, k, dilation, idx);
  return nsid_launch_status();
}

template <typename T, int NT>
int launch_knn_rank_nt(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C, int k,
                       int dilation, int32_t* idx, hipStream_t s) {
  const size_t bytes = (size_t)2 * N * C * 2 + ((size_t)N + (size_t)KNN2_WAVES * 16 * (N + 4)) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn_rank_kernel<T, NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  NSID_LAUNCH((knn_rank_kernel<T, NT>), dim3(B), dim3(KNN2_THREADS), bytes, s, static_cast<const T*>(r), (long)ldr, scale,
              shift, N, C, k, dilation, idx);
  return nsid_launch_status();
}
template <typename T>
int launch_knn_rank(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C, int k,
                    int dilation, int32_t* idx, hipStream_t s) {
  // column tiles per MFMA pass: 4 where a wave owns a whole strip (N = 128); N <= 64 runs two waves per strip, each with half of its
  // column tiles: 2 (N = 64) or 1 (N = 32)
  const int NS = N >> 4;
  if (2 * NS <= KNN2_WAVES)
    return NS >= 4 ? launch_knn_rank_nt<T, 2>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s)
                   : launch_knn_rank_nt<T, 1>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
  return launch_knn_rank_nt<T, 4>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
}

template <typename T, int KD, int NT>
int launch_knn2_nt(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C, int k,
                   int dilation, int32_t* idx, hipStream_t s) {
  const size_t bytes = (size_t)2 * N * C * 2 + ((size_t)N + 2 * 4 * 16 * KD) * sizeof(float);      // two fp16 images + |y|^2 + hand-over lists
  static bool configured = false;      // raise the dynamic-LDS cap once per instantiation (not a per-call sync)
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn2_kernel<T, KD, NT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  const int pair_min = nsid_tune(NSID_T_knn_pair_min);      // clips from which two workgroups share a CU (0 = never)
  if constexpr (std::is_same<T, __bf16>::value) {
    // stored bf16 features without an affine (eval mode, BatchNorm folded) and a launch of many clips: one bf16 MFMA pass on the raw
    // features (knn2_raw_kernel); a training step (affine on the load, <= 256 clips) never takes it, so its arithmetic does not change
    if (scale == nullptr && nsid_tune(NSID_T_knn_raw16) != 0 && pair_min > 0 && B >= pair_min) {
      const size_t rbytes = (size_t)N * C * 2 + ((size_t)2 * N + 2 * 4 * 16 * KD) * sizeof(float);
      nsid_count(NSID_C_knn2_raw);
      const long wpe = nsid_tune(NSID_T_knn_raw_wpe);
#define NSID_KNN_RAW_GO(W_)                                                                                                     \
      do {                                                                                                                      \
        static bool configured3 = false;                                                                                        \
        if (!configured3) {                                                                                                     \
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn2_raw_kernel<KD, NT, W_>),                                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                        \
            return NSID_ELAUNCH;                                                                                                \
          configured3 = true;                                                                                                   \
        }                                                                                                                       \
        NSID_LAUNCH((knn2_raw_kernel<KD, NT, W_>), dim3(B), dim3(KNN2_THREADS), rbytes, s, static_cast<const __bf16*>(r), (long)ldr, N, \
                    C, k, dilation, idx);                                                                                       \
      } while (0)
      if (wpe >= 6) NSID_KNN_RAW_GO(6);       // (8 waves per SIMD = 64 registers spills 20-48 bytes per lane at N >= 128 and measured the same)
      else NSID_KNN_RAW_GO(4);
#undef NSID_KNN_RAW_GO
      return nsid_launch_status();
    }
  }
  if (pair_min > 0 && B >= pair_min && 2 * bytes <= 160 * 1024) {
    static bool configured2 = false;
    if (!configured2) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn2_pair_kernel<T, KD, NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return NSID_ELAUNCH;
      configured2 = true;
    }
    nsid_count(NSID_C_knn2_pair);
    NSID_LAUNCH((knn2_pair_kernel<T, KD, NT>), dim3(B), dim3(KNN2_THREADS), bytes, s, static_cast<const T*>(r), (long)ldr,
                scale, shift, N, C, k, dilation, idx);
    return nsid_launch_status();
  }
  NSID_LAUNCH((knn2_kernel<T, KD, NT>), dim3(B), dim3(KNN2_THREADS), bytes, s, static_cast<const T*>(r), (long)ldr, scale,
              shift, N, C, k, dilation, idx);
  return nsid_launch_status();
}
template <typename T, int KD>
int launch_knn2(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C, int k, int dilation,
                int32_t* idx, hipStream_t s) {
  const int RT = N >> 4, CTG = RT / (RT >= KNN2_WAVES ? 1 : 2);      // column tiles per wave and row tile (kernel: G, CTG)
  if (CTG >= 4) return launch_knn2_nt<T, KD, 4>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
  if (CTG == 2) return launch_knn2_nt<T, KD, 2>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
  return launch_knn2_nt<T, KD, 1>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
}

}  // namespace

extern "C" int nsid_debug_knn_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_knn_trace), &buf, sizeof(buf)) == hipSuccess ? NSID_OK : NSID_EINVAL;
}

static int knn_graph_impl(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C,
                          int k, int dilation, int32_t* idx, int dtype, void* stream) {
  NSID_REQUIRE(r && idx && B > 0 && k > 0 && dilation > 0 && NSID_DTYPE_OK(dtype));
  NSID_REQUIRE(ldr % (dtype == NSID_BF16 ? 8 : 4) == 0);
  NSID_REQUIRE(N % 32 == 0 && C % 16 == 0 && ldr % 4 == 0 && ldr >= C && nsid_aligned16(r));
  NSID_REQUIRE(k * dilation <= N);
  NSID_REQUIRE((scale == nullptr) == (shift == nullptr));
  // beyond grafp.yaml's graphs (more than 256 nodes, or a clip of more than 32 768 features — the kernels below keep a clip in LDS;
  // 16 384 in every timed configuration, up to 32 768 at the encoder sizes 'm' / default): one workgroup per 16-row strip
  if (N > 256 || (long)N * C > 32768) {
    const size_t bytes = ((size_t)16 * (C + 4) + 3 * (size_t)N + (size_t)16 * (N + 4)) * sizeof(float);
    NSID_REQUIRE(bytes <= 160 * 1024 && B <= 65535);
    nsid_count(NSID_C_knn_big);
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn_big_kernel<float>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(knn_big_kernel<__bf16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return NSID_ELAUNCH;
      configured = true;
    }
    NSID_DISPATCH_DTYPE(dtype, T, {
      NSID_LAUNCH((knn_big_kernel<T>), dim3(N / 16, B), dim3(256), bytes, static_cast<hipStream_t>(stream),
                  static_cast<const T*>(r), (long)ldr, scale, shift, N, C, k, dilation, idx);
    });
    return nsid_launch_status();
  }
  const int kd = k * dilation;
  const bool use_fast = nsid_tune(NSID_T_knn_strips) == 0;
  const bool pow2 = C >= 64 && C <= 512 && (C & (C - 1)) == 0 && N >= 32 && (N & (N - 1)) == 0;
  // deep configuration: threshold select in registers where a row is much longer than the wanted list (256- and 128-node graphs),
  // rank counting where k*d is most of the row (64 and 32 nodes)
  if (use_fast && pow2 && kd > 8 && kd <= 64 && N >= nsid_tune(NSID_T_knn_sel_min_n)) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = 1;
#define NSID_KSEL_CASE(TL)                                                                                            \
    case 16 * TL:                                                                                                    \
      rc = dtype == NSID_BF16 ? launch_knn_sel<__bf16, TL>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s)       \
                              : launch_knn_sel<float, TL>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);       \
      break;
    switch (N) { NSID_KSEL_CASE(16) NSID_KSEL_CASE(8) NSID_KSEL_CASE(4) NSID_KSEL_CASE(2) default: break; }
#undef NSID_KSEL_CASE
    if (rc != 1) { nsid_count(NSID_C_knn_sel); return rc; }
  }
  if (use_fast && pow2 && kd > 8 && N <= 128) {              // small graphs: rank counting
    nsid_count(NSID_C_knn_rank);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return dtype == NSID_BF16 ? launch_knn_rank<__bf16>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s)
                              : launch_knn_rank<float>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
  }
  if (use_fast && kd <= 8 && pow2) {
    nsid_count(NSID_C_knn2);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == NSID_BF16) {
      if (kd <= 3) return launch_knn2<__bf16, 3>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
      if (kd <= 5) return launch_knn2<__bf16, 5>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
      return launch_knn2<__bf16, 8>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
    }
    if (kd <= 3) return launch_knn2<float, 3>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
    if (kd <= 5) return launch_knn2<float, 5>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
    return launch_knn2<float, 8>(r, ldr, scale, shift, B, N, C, k, dilation, idx, s);
  }
  int waves = KNN_WAVES;
  auto lds_bytes = [&](int w) { return ((size_t)N * (C + 4) + N + (size_t)w * 16 * (N + 4)) * sizeof(float); };
  while (waves > 1 && lds_bytes(waves) > 160 * 1024) waves >>= 1;
  const size_t bytes = lds_bytes(waves);
  NSID_REQUIRE(bytes <= 160 * 1024);
  nsid_count(NSID_C_knn_strips);
  static size_t configured = 0;       // raise the dynamic-LDS cap once per size step (not a per-call sync)
  if (bytes > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel<float>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel<__bf16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = 160 * 1024;
  }
  NSID_DISPATCH_DTYPE(dtype, T, {
    NSID_LAUNCH((knn_kernel<T>), dim3(B), dim3(64 * waves), bytes, static_cast<hipStream_t>(stream),
                static_cast<const T*>(r), (long)ldr, scale, shift, N, C, k, dilation, idx);
  });
  return nsid_launch_status();
}

extern "C" int nsid_knn_graph(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C,
                              int k, int dilation, int32_t* idx, int dtype, void* stream) {
  return knn_graph_impl(r, ldr, scale, shift, B, N, C, k, dilation, idx, dtype, stream);
}

