// Dilated kNN graph of one clip per workgroup: the whole clip (N nodes x C channels, 64 KB at every stage of
// the 't' encoder) is staged once into LDS, BatchNorm-applied and L2-normalised there, the N x N distance matrix
// is produced 16 rows at a time by fp32 MFMA straight from LDS and consumed by the top-k selection without ever
// reaching HBM.  Algorithmic HBM traffic = read N*C*4 B, write N*k*4 B per clip.
#include "nsid_common.h"

namespace {

constexpr int KNN_THREADS = 256;   // 4 waves, one 16-row strip each per pass
constexpr int KNN_WAVES = 4;

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
__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const T* __restrict__ r, long ldr,
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
    const float denom = fmaxf(sqrtf(ss), 1e-12f);
    float s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float v = yn[n * LD + c] / denom;
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

}  // namespace

extern "C" int nsid_knn_graph(const void* r, int ldr, const float* scale, const float* shift, int B, int N, int C,
                              int k, int dilation, int32_t* idx, int dtype, void* stream) {
  NSID_REQUIRE(r && idx && B > 0 && k > 0 && dilation > 0 && NSID_DTYPE_OK(dtype));
  NSID_REQUIRE(ldr % (dtype == NSID_BF16 ? 8 : 4) == 0);
  NSID_REQUIRE(N % 32 == 0 && N <= 256 && C % 16 == 0 && ldr % 4 == 0 && ldr >= C && nsid_aligned16(r));
  NSID_REQUIRE(k * dilation <= N);
  NSID_REQUIRE((scale == nullptr) == (shift == nullptr));
  const size_t bytes = ((size_t)N * (C + 4) + N + (size_t)KNN_WAVES * 16 * (N + 4)) * sizeof(float);
  NSID_REQUIRE(bytes <= 160 * 1024);
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
    NSID_LAUNCH((knn_kernel<T>), dim3(B), dim3(KNN_THREADS), bytes, static_cast<hipStream_t>(stream),
                static_cast<const T*>(r), (long)ldr, scale, shift, N, C, k, dilation, idx);
  });
  return nsid_launch_status();
}
