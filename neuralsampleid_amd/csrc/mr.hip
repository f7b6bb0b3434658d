// Max-relative neighbour aggregation (MRConv2d without its grouped conv) — HBM-bound gather kernels.
// forward : one thread per (row, channel quad); the k neighbour rows of a clip are re-read from L2 (a clip is 64 KB),
//           so HBM sees x once in, u once out (2*N*C*4 + N*k*4 bytes per clip).
// backward: a GATHER over the reversed graph, one workgroup per clip: du_odd and the arg-max bytes are staged in LDS,
//           the neighbour lists are reversed into a CSR with integer LDS atomics, every thread sums its incoming edges.
#include <algorithm>
#include <cstdlib>

#include "nsid_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void mr_fwd_kernel(const T* __restrict__ r, long ldr,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const int32_t* __restrict__ idx, long rows_total, int N, int C,
                                                     int k, T* __restrict__ u, uint8_t* __restrict__ argmax) {
  constexpr int NV = Chunk<T>::N;           // channels per thread: 4 (fp32) or 8 (bf16)
  const int CV = C / NV;
  const long total = rows_total * CV;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long row = q / CV;
    const int c = (int)(q % CV) * NV;
    const long clip0 = (row / N) * N;
    float sc[NV], sh[NV], y[NV], best[NV];
    int arg[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      sc[e] = 1.f;
      sh[e] = 0.f;
      best[e] = -__builtin_inff();
      arg[e] = 0;
    }
    if (scale != nullptr) {
      load_channels<NV>(scale, c, sc);
      load_channels<NV>(shift, c, sh);
    }
    Chunk<T>::load(r + row * ldr + c, y);
#pragma unroll
    for (int e = 0; e < NV; ++e) y[e] = sc[e] * y[e] + sh[e];
    const int32_t* nb = idx + row * k;
    for (int j = 0; j < k; ++j) {
      const long nrow = clip0 + min(max(nb[j], 0), N - 1);     // ids come from the caller: never fault on them
      float v[NV];
      Chunk<T>::load(r + nrow * ldr + c, v);
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const float d = (sc[e] * v[e] + sh[e]) - y[e];
        if (d > best[e]) { best[e] = d; arg[e] = j; }       // strict: first maximum wins, as torch.max
      }
    }
    // interleave: channel 2c = y, 2c+1 = max-relative; 2*NV outputs = two chunks
    float o[2 * NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) { o[2 * e] = y[e]; o[2 * e + 1] = best[e]; }
    T* dst = u + row * (2L * C) + 2 * c;
    Chunk<T>::store(dst, o);
    Chunk<T>::store(dst + NV, o + NV);
    if (argmax != nullptr) {
#pragma unroll
      for (int e = 0; e < NV; e += 4)
        *reinterpret_cast<uchar4*>(argmax + row * C + c + e) =
            make_uchar4((unsigned char)arg[e], (unsigned char)arg[e + 1], (unsigned char)arg[e + 2],
                        (unsigned char)arg[e + 3]);
    }
  }
}

// LDS-staged forward: ONE workgroup per clip. The clip's raw features (N*C elements: 32 KB as bf16, 64 KB as fp32) are
// streamed into LDS once with 16-byte coalesced loads; every (node, 8-channel chunk) then reads its own row and its k
// neighbour rows from LDS, so HBM sees r once in and u / arg-max once out, and the gather never leaves the CU (the
// grid-stride form above re-reads neighbour rows through L2: 2.8 TB/s of algorithmic bytes at batch 256).
constexpr int MRF_THREADS = 512;

template <typename T>
__global__ __launch_bounds__(MRF_THREADS) void mr_fwd_lds_kernel(const T* __restrict__ r, long ldr,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const int32_t* __restrict__ idx, int N, int C, int k,
                                                                 T* __restrict__ u, uint8_t* __restrict__ argmax,
                                                                 int split) {
  extern __shared__ __attribute__((aligned(16))) char mrf_smem[];
  constexpr int NV = Chunk<T>::N;
  T* clip = reinterpret_cast<T*>(mrf_smem);                       // [N][Cp] raw values (the affine is applied at use)
  // split > 1: `split` workgroups share a clip BY CHANNELS — channels are independent in the aggregation, so each workgroup
  // stages and aggregates only its own C/split columns (no redundant bytes) and two or four workgroups per CU overlap
  // their load / compute / store phases: at batch 256 one workgroup per clip is one per CU and latency-bound
  const int b = blockIdx.x / split, part = blockIdx.x % split, t = threadIdx.x;
  const long row0 = (long)b * N;
  const int CV = C / NV, CVp = CV / split, Cp = CVp * NV, c_lo = part * Cp, total = N * CVp;
  // the clip's neighbour lists go to LDS too (coalesced, clamped once): read from global inside the neighbour loop they were k
  // dependent L1 round trips per chunk (k = 18 in the deep configuration)
  int* nbl = reinterpret_cast<int*>(mrf_smem + (size_t)N * Cp * sizeof(T));
  for (int q = t; q < N * k; q += MRF_THREADS) nbl[q] = min(max(idx[row0 * k + q], 0), N - 1);
  for (int q = t; q < total; q += MRF_THREADS) {
    const int n = q / CVp, cl = (q % CVp) * NV;
    *reinterpret_cast<f32x4*>(clip + (long)n * Cp + cl) = *reinterpret_cast<const f32x4*>(r + (row0 + n) * ldr + c_lo + cl);
  }
  __syncthreads();
  for (int q = t; q < total; q += MRF_THREADS) {
    const int n = q / CVp, cl = (q % CVp) * NV, c = c_lo + cl;
    float sc[NV], sh[NV], y[NV], best[NV];
    int arg[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) { sc[e] = 1.f; sh[e] = 0.f; best[e] = -__builtin_inff(); arg[e] = 0; }
    if (scale != nullptr) {
      load_channels<NV>(scale, c, sc);
      load_channels<NV>(shift, c, sh);
    }
    Chunk<T>::load(clip + (long)n * Cp + cl, y);
#pragma unroll
    for (int e = 0; e < NV; ++e) y[e] = sc[e] * y[e] + sh[e];
    const int* nb = nbl + n * k;
    for (int j = 0; j < k; ++j) {
      const int m = nb[j];                                        // (clamped when staged: never read outside the clip)
      float v[NV];
      Chunk<T>::load(clip + (long)m * Cp + cl, v);
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const float d = (sc[e] * v[e] + sh[e]) - y[e];
        if (d > best[e]) { best[e] = d; arg[e] = j; }             // strict: first maximum wins, as torch.max
      }
    }
    float o[2 * NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) { o[2 * e] = y[e]; o[2 * e + 1] = best[e]; }
    T* dst = u + (row0 + n) * (2L * C) + 2 * c;
    Chunk<T>::store(dst, o);
    Chunk<T>::store(dst + NV, o + NV);
    if (argmax != nullptr) {
#pragma unroll
      for (int e = 0; e < NV; e += 4)
        *reinterpret_cast<uchar4*>(argmax + (row0 + n) * C + c + e) =
            make_uchar4((unsigned char)arg[e], (unsigned char)arg[e + 1], (unsigned char)arg[e + 2],
                        (unsigned char)arg[e + 3]);
    }
  }
}

// The same aggregation for MANY neighbours (the deep configuration: k = 18), bf16 storage. mr_fwd_lds_kernel spends six VALU operations
// per neighbour and element (convert, affine, subtract, compare, two selects): 13 us of an 18.8 us launch at k = 18. Here the search is
// done on INTEGERS:
//   * The clip is staged as order-preserving 16-bit keys (sign-magnitude -> unsigned; NaN -> 0, below everything), flipped per channel
//     where the BatchNorm scale is negative. d_j = (sc v_j + sh) - y is a monotone non-decreasing function of the key (every rounding
//     involved is monotone), so the neighbour with the largest key has the largest d.
//   * Key and position travel together: (key << 16) | (255 - j). ONE v_lshl_or / v_and_or and ONE v_max_u32 per neighbour and element;
//     among equal keys the smallest j wins (torch.max: the first maximum).
//   * d is evaluated ONCE per element from the winning key, with the expressions of mr_fwd_lds_kernel -- the same value bit for bit.
//   * What integers cannot see: two DIFFERENT keys whose d rounds to the same fp32 (tiny values beside a large shift; a zero scale;
//     +-inf) -- torch.max then takes the first of them. Those sit directly below the winner, so one extra evaluation of d at key - 1 says
//     whether that can have happened; if any lane of the wave says so, the wave redoes its chunk with the scalar search (rare: the data
//     must put a neighbour within 2^-16 of the maximum relative to the shift).
//   * d that is not above -inf (NaN, -inf): best = -inf, arg-max 0, as the strict '>' from -inf leaves them.
__device__ __forceinline__ uint32_t mrk_sortable2(uint32_t x) {           // two bf16 -> two keys
  const uint32_t neg = ((x >> 15) & 0x00010001u) * 0xFFFFu;
  uint32_t key = x ^ (neg | 0x80008000u);
  const uint32_t nan = ((((x & 0x7FFF7FFFu) + 0x007F007Fu) >> 15) & 0x00010001u) * 0xFFFFu;
  return key & ~nan;
}
__device__ __forceinline__ float mrk_value(uint32_t key16) {                // key -> the bf16 value as fp32 (key 0 -> NaN)
  const uint32_t raw = (key16 & 0x8000u) ? (key16 ^ 0x8000u) : (~key16 & 0xFFFFu);
  return __builtin_bit_cast(float, raw << 16);
}

__global__ __launch_bounds__(MRF_THREADS) void mr_fwd_key_kernel(const __bf16* __restrict__ r, long ldr,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const int32_t* __restrict__ idx, int N, int C, int k,
                                                                 __bf16* __restrict__ u, uint8_t* __restrict__ argmax, int split) {
  extern __shared__ __attribute__((aligned(16))) char mrf_smem[];
  constexpr int NV = 8;
  uint16_t* keys = reinterpret_cast<uint16_t*>(mrf_smem);           // [N][Cp]
  const int b = blockIdx.x / split, part = blockIdx.x % split, t = threadIdx.x;
  const long row0 = (long)b * N;
  const int CV = C / NV, CVp = CV / split, Cp = CVp * NV, c_lo = part * Cp, total = N * CVp;
  int* nbl = reinterpret_cast<int*>(mrf_smem + (size_t)N * Cp * 2);
  for (int q = t; q < N * k; q += MRF_THREADS) nbl[q] = min(max(idx[row0 * k + q], 0), N - 1);
  for (int q = t; q < total; q += MRF_THREADS) {
    const int n = q / CVp, cl = (q % CVp) * NV;
    const u32x4 x = *reinterpret_cast<const u32x4*>(r + (row0 + n) * ldr + c_lo + cl);
    float sc[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) sc[e] = 1.f;
    if (scale != nullptr) load_channels<NV>(scale, c_lo + cl, sc);
    u32x4 y;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      y[p] = mrk_sortable2(x[p]) ^ ((sc[2 * p] < 0.f ? 0x0000FFFFu : 0u) | (sc[2 * p + 1] < 0.f ? 0xFFFF0000u : 0u));
    *reinterpret_cast<u32x4*>(keys + (long)n * Cp + cl) = y;
  }
  __syncthreads();
  for (int q = t; q < total; q += MRF_THREADS) {
    const int n = q / CVp, cl = (q % CVp) * NV, c = c_lo + cl;
    float sc[NV], sh[NV], y[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (scale != nullptr) {
      load_channels<NV>(scale, c, sc);
      load_channels<NV>(shift, c, sh);
    }
    Chunk<__bf16>::load(r + (row0 + n) * ldr + c, y);             // the node's own values: raw, from memory (a NaN stays a NaN)
#pragma unroll
    for (int e = 0; e < NV; ++e) y[e] = sc[e] * y[e] + sh[e];
    const int* nb = nbl + n * k;
    uint32_t win[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) win[e] = 0u;
    for (int j = 0; j < k; ++j) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(keys + (long)nb[j] * Cp + cl);
      const uint32_t cj = 255u - (uint32_t)j;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        win[2 * p] = max(win[2 * p], (w[p] << 16) | cj);
        win[2 * p + 1] = max(win[2 * p + 1], (w[p] & 0xFFFF0000u) | (cj & 0x0000FFFFu));      // one v_bfi_b32
      }
    }
    float best[NV];
    int arg[NV];
    bool redo = false;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const uint32_t flip = sc[e] < 0.f ? 0xFFFFu : 0u;
      const uint32_t key = win[e] >> 16;
      const float d = (sc[e] * mrk_value(key ^ flip) + sh[e]) - y[e];
      const float dn = (sc[e] * mrk_value((key - 1u) ^ flip) + sh[e]) - y[e];       // (key 0: NaN either way)
      redo |= dn == d;
      const bool up = d > -__builtin_inff();
      best[e] = up ? d : -__builtin_inff();
      arg[e] = up ? 255 - (int)(win[e] & 255u) : 0;
    }
    if (__builtin_amdgcn_ballot_w64(redo) != 0) {             // the scalar search of mr_fwd_lds_kernel, on the decoded keys
#pragma unroll
      for (int e = 0; e < NV; ++e) { best[e] = -__builtin_inff(); arg[e] = 0; }
      for (int j = 0; j < k; ++j) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(keys + (long)nb[j] * Cp + cl);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          const uint32_t flip = sc[e] < 0.f ? 0xFFFFu : 0u;
          const uint32_t key = (e & 1) ? (w[e / 2] >> 16) : (w[e / 2] & 0xFFFFu);
          const float d = (sc[e] * mrk_value(key ^ flip) + sh[e]) - y[e];
          if (d > best[e]) { best[e] = d; arg[e] = j; }
        }
      }
    }
    float o[2 * NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) { o[2 * e] = y[e]; o[2 * e + 1] = best[e]; }
    __bf16* dst = u + (row0 + n) * (2L * C) + 2 * c;
    Chunk<__bf16>::store(dst, o);
    Chunk<__bf16>::store(dst + NV, o + NV);
    if (argmax != nullptr) {
#pragma unroll
      for (int e = 0; e < NV; e += 4)
        *reinterpret_cast<uchar4*>(argmax + (row0 + n) * C + c + e) =
            make_uchar4((unsigned char)arg[e], (unsigned char)arg[e + 1], (unsigned char)arg[e + 2], (unsigned char)arg[e + 3]);
    }
  }
}

// backward as a GATHER over the reversed graph (LDS float atomics cost ~3 cycles per lane: the scatter form spent 2/3 of
// its time in 16 K ds_add_f32 per clip). Per clip:
//   A. stage the odd half of du (the max-relative gradient, in its storage type) and the arg-max bytes in LDS;
//   B. reverse the neighbour lists with N*k INTEGER atomics: in-degree count, wave scan, fill -> CSR (start, src);
//   C. every thread owns NV channels of one node n: dy = du_even - du_odd (own row) + sum over the edges (m, j) that
//      point at n of du_odd[m, c] where argmax[m, c] == j.
// The order of a node's incoming edges depends on the fill race, so the fp32 sum order is not fixed (neither was the
// atomic scatter's); a node's incoming values are few (k on average).
// (Round 4 re-tried the scatter with ONE element per lane -- 64 consecutive channels per ds_add_f32, two lanes per bank, no conflicts,
// work independent of k: 31 us per launch at k = 3 and at k = 18 alike, against 10-12 / 22-29 us for this gather. The LDS float-add
// unit retires about one lane per 2.7 cycles whatever the banks: 16 384 adds per clip are 20 us. docs/experiments.md.)
constexpr int MRB_THREADS = 1024;

template <typename T>
__global__ __launch_bounds__(MRB_THREADS) void mr_bwd_kernel(const T* __restrict__ du, const int32_t* __restrict__ idx,
                                                             const uint8_t* __restrict__ argmax, int N, int C, int k,
                                                             T* __restrict__ dy) {
  extern __shared__ __attribute__((aligned(16))) char mrb_smem[];
  constexpr int NV = Chunk<T>::N;          // dy channels per thread; the matching du run is 2*NV elements (32 B)
  T* odd = reinterpret_cast<T*>(mrb_smem);                          // [N][C]
  uint8_t* am = reinterpret_cast<uint8_t*>(odd + (long)N * C);      // [N][C]
  int* cnt = reinterpret_cast<int*>(am + (long)N * C);              // [N]   in-degree, then fill cursor
  int* start = cnt + N;                                             // [N+1]
  int* src = start + N + 1;                                         // [N*k] (m << 8) | j
  const int b = blockIdx.x, t = threadIdx.x;
  const long row0 = (long)b * N;
  const int CV = C / NV;
  const int total = N * CV;
  const T* dub = du + row0 * (2L * C);
  const uint8_t* amb = argmax + row0 * C;
  const int32_t* nbb = idx + row0 * k;
  const int E = N * k;

  for (int n = t; n < N; n += MRB_THREADS) cnt[n] = 0;
  // ---- A: stage du_odd and argmax. A thread meets the same (node, chunk) items again in phase C: their own-row term
  // du_even - du_odd stays in registers across the graph reversal (KEEP items: every stage of the encoder has N*C = 16 384, i.e.
  // 2 items per thread in bf16 and 4 in fp32), so du is read from memory ONCE (the second read was 1.36x the algorithmic bytes)
  constexpr int KEEP = 16384 / (NV * MRB_THREADS);
  float own[KEEP][NV];
#pragma unroll
  for (int it = 0; it < KEEP; ++it) {
    const int q = t + it * MRB_THREADS;
    if (q >= total) break;
    float g0[NV], g1[NV], o[NV];
    Chunk<T>::load(dub + (long)q * 2 * NV, g0);
    Chunk<T>::load(dub + (long)q * 2 * NV + NV, g1);
#pragma unroll
    for (int e = 0; e < NV / 2; ++e) {
      o[e] = g0[2 * e + 1]; o[NV / 2 + e] = g1[2 * e + 1];
      own[it][e] = g0[2 * e] - g0[2 * e + 1];
      own[it][NV / 2 + e] = g1[2 * e] - g1[2 * e + 1];
    }
    Chunk<T>::store(odd + (long)q * NV, o);                  // exact: a copy in the storage type
#pragma unroll
    for (int e = 0; e < NV; e += 4)
      *reinterpret_cast<uint32_t*>(am + (long)q * NV + e) = *reinterpret_cast<const uint32_t*>(amb + (long)q * NV + e);
  }
  for (int q = t + KEEP * MRB_THREADS; q < total; q += MRB_THREADS) {     // larger clips (not in this encoder): LDS only
    float g0[NV], g1[NV], o[NV];
    Chunk<T>::load(dub + (long)q * 2 * NV, g0);
    Chunk<T>::load(dub + (long)q * 2 * NV + NV, g1);
#pragma unroll
    for (int e = 0; e < NV / 2; ++e) { o[e] = g0[2 * e + 1]; o[NV / 2 + e] = g1[2 * e + 1]; }
    Chunk<T>::store(odd + (long)q * NV, o);
#pragma unroll
    for (int e = 0; e < NV; e += 4)
      *reinterpret_cast<uint32_t*>(am + (long)q * NV + e) = *reinterpret_cast<const uint32_t*>(amb + (long)q * NV + e);
  }
  __syncthreads();
  // ---- B: reversed graph
  for (int e = t; e < E; e += MRB_THREADS) atomicAdd(&cnt[min(max(nbb[e], 0), N - 1)], 1);
  __syncthreads();
  if (t < 64) {                                              // exclusive scan of cnt[0..N) by one wave
    const int per = (N + 63) / 64;
    int loc = 0;
    for (int i2 = 0; i2 < per; ++i2) { const int n = t * per + i2; if (n < N) loc += cnt[n]; }
    int inc = loc;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(inc, o2, 64); if (t >= o2) inc += v; }
    int run = inc - loc;
    for (int i2 = 0; i2 < per; ++i2) {
      const int n = t * per + i2;
      if (n < N) { start[n] = run; const int c2 = cnt[n]; cnt[n] = run; run += c2; }
    }
    if (t == 63) start[N] = inc;
  }
  __syncthreads();
  for (int e = t; e < E; e += MRB_THREADS) {
    const int tg = min(max(nbb[e], 0), N - 1);
    const int p = atomicAdd(&cnt[tg], 1);
    src[p] = ((e / k) << 8) | (e % k);
  }
  __syncthreads();
  // ---- C: gather
  auto gather = [&](int q, float (&v)[NV]) {
    const int n = q / CV, c = (q % CV) * NV;
    const int p1 = start[n + 1];
    for (int p = start[n]; p < p1; ++p) {
      const int s2 = src[p];
      const int m = s2 >> 8, jj = s2 & 255;
      float o[NV];
      Chunk<T>::load(odd + (long)m * C + c, o);
      const uint8_t* a = am + (long)m * C + c;
#pragma unroll
      for (int e = 0; e < NV; e += 4) {
        const uint32_t a4 = *reinterpret_cast<const uint32_t*>(a + e);
#pragma unroll
        for (int x = 0; x < 4; ++x) v[e + x] += (int)((a4 >> (8 * x)) & 255u) == jj ? o[e + x] : 0.f;
      }
    }
    Chunk<T>::store(dy + row0 * C + (long)q * NV, v);
  };
#pragma unroll
  for (int it = 0; it < KEEP; ++it) {
    const int q = t + it * MRB_THREADS;
    if (q >= total) break;
    gather(q, own[it]);
  }
  for (int q = t + KEEP * MRB_THREADS; q < total; q += MRB_THREADS) {
    float g0[NV], g1[NV], v[NV];
    Chunk<T>::load(dub + (long)q * 2 * NV, g0);             // second read of du: L2
    Chunk<T>::load(dub + (long)q * 2 * NV + NV, g1);
#pragma unroll
    for (int e = 0; e < NV / 2; ++e) {
      v[e] = g0[2 * e] - g0[2 * e + 1];
      v[NV / 2 + e] = g1[2 * e] - g1[2 * e + 1];
    }
    gather(q, v);
  }
}

// ---- backward with many neighbours (deep plan: k = 18), bf16 storage, N*C = 16 384, C a power of two.
// The gather above costs a compare per incoming edge and element (~48 vector instructions per edge and 8 channels as the compiler
// writes them) and a wave waits for the largest in-degree among the 1-8 nodes its lanes hold. This form
//   * ranks the nodes of the clip by in-degree (N <= 256: every thread counts a slice of the others) and hands a wave nodes of
//     NEIGHBOURING rank, the heaviest group together with the lightest one (two items per thread), so that the lanes of a wave run
//     about the same number of iterations and the waves about the same total;
//   * does the select in 20 instructions per edge and 8 channels: v_cmp_eq_u32_sdwa compares one arg-max BYTE with the edge's
//     position, v_cndmask_b32_sdwa moves the selected bf16 HALF-WORD into the high half of a zero (= the fp32 value or 0), and the
//     sums are packed fp32 adds;
//   * keeps the one read of du and the LDS footprint of the form above (a clip's du_odd + arg-max bytes + the edge list, ~70 KB: a
//     first version that parked the sums in 64 KB more of LDS was 6 us faster alone and 0.2 ms SLOWER in the deep step — it shut the
//     other view's workgroups out of its CU): a thread requests the du chunks of ITS ranked node once the order is known and the
//     loads fly while the edge list is filled; the own-row term seeds the sum in the same thread.
// Sums are fp32: the own-row term, then the incoming edges in the order of the reversed edge list.
constexpr int MRS_THREADS = 1024;

// v[e] += (byte e of {a0, a1} == byte 0 of s2) ? fp32(bf16 e of o) : 0, e = 0 .. 7
__device__ __forceinline__ void mrs_edge(f32x2 (&v)[4], const u32x4 o, const uint32_t a0, const uint32_t a1, const uint32_t s2,
                                         const uint32_t zero) {
  float t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "v_cmp_eq_u32_sdwa vcc, %[a0], %[s] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t0], %[z], %[o0], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
      "v_cmp_eq_u32_sdwa vcc, %[a0], %[s] src0_sel:BYTE_1 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t1], %[z], %[o0], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
      "v_cmp_eq_u32_sdwa vcc, %[a0], %[s] src0_sel:BYTE_2 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t2], %[z], %[o1], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
      "v_cmp_eq_u32_sdwa vcc, %[a0], %[s] src0_sel:BYTE_3 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t3], %[z], %[o1], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
      "v_cmp_eq_u32_sdwa vcc, %[a1], %[s] src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t4], %[z], %[o2], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
      "v_cmp_eq_u32_sdwa vcc, %[a1], %[s] src0_sel:BYTE_1 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t5], %[z], %[o2], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
      "v_cmp_eq_u32_sdwa vcc, %[a1], %[s] src0_sel:BYTE_2 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t6], %[z], %[o3], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
      "v_cmp_eq_u32_sdwa vcc, %[a1], %[s] src0_sel:BYTE_3 src1_sel:BYTE_0\n\t"
      "v_cndmask_b32_sdwa %[t7], %[z], %[o3], vcc dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
      "s_nop 0"       // gfx950: a VALU that wrote with dst_sel != DWORD needs one wait state before its result is read
      : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
      : [a0] "v"(a0), [a1] "v"(a1), [s] "v"(s2), [z] "v"(zero), [o0] "v"(o[0]), [o1] "v"(o[1]), [o2] "v"(o[2]), [o3] "v"(o[3])
      : "vcc");
  v[0] += f32x2{t0, t1};
  v[1] += f32x2{t2, t3};
  v[2] += f32x2{t4, t5};
  v[3] += f32x2{t6, t7};
}

// BNS: the aggregation's input was y = act(BN(r)) of a conv+BN layer (Grapher fc1): the launch also returns that BatchNorm's backward
// column sums over the clip's rows, partial[0][b][c] = sum_n g, partial[1][b][c] = sum_n g * xhat with g = dy * act'(sc r + sh) taken from
// the ROUNDED dy (what a separate reduce pass would read back): nsid_bn_bwd_finalize[_fused] with tiles = B consumes them.
struct MrsBn {
  const __bf16* r; long ldr;
  const float* scale; const float* shift; const float* mean; const float* invstd; float slope;
  float* partial; long plane;      // plane = B * C
};

template <bool BNS>
__global__ __launch_bounds__(MRS_THREADS, 8) void mr_bwd_sorted_kernel(const __bf16* __restrict__ du, const int32_t* __restrict__ idx,
                                                                    const uint8_t* __restrict__ argmax, int N, int C, int cshift, int k,
                                                                    __bf16* __restrict__ dy, const MrsBn bn) {
  extern __shared__ __attribute__((aligned(16))) char mrs_smem[];
  constexpr int NC = 16384;                                          // elements of a clip (host-checked)
  char* const odd = mrs_smem;                                        // [N][C] bf16
  char* const am = mrs_smem + 2 * NC;                                // [N][C] arg-max bytes
  int* const cnt = reinterpret_cast<int*>(mrs_smem + 3 * NC);        // [N] in-degree
  int* const cur = cnt + N;                                          // [N] fill cursor
  int* const rank = cur + N;                                         // [N] position in the order by (in-degree descending, node)
  int* const perm = rank + N;                                        // [N] its inverse
  int* const start = perm + N;                                       // [N + 1]
  int* const src = start + N + 1;                                    // [N * k]: (source row * C) << 8 | neighbour position
  const int b = blockIdx.x, t = threadIdx.x;
  const long row0 = (long)b * N;
  const int E = N * k;
  const int32_t* nbb = idx + row0 * k;

  for (int n = t; n < N; n += MRS_THREADS) { cnt[n] = 0; rank[n] = 0; }
  __syncthreads();
  for (int e = t; e < E; e += MRS_THREADS) atomicAdd(&cnt[min(max(nbb[e], 0), N - 1)], 1);
  __syncthreads();
  if (t < 64) {                                                      // exclusive scan of cnt[0..N) by one wave
    const int per = (N + 63) / 64;
    int loc = 0;
    for (int i2 = 0; i2 < per; ++i2) { const int n = t * per + i2; if (n < N) loc += cnt[n]; }
    int inc = loc;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(inc, o2, 64); if (t >= o2) inc += v; }
    int run = inc - loc;
    for (int i2 = 0; i2 < per; ++i2) {
      const int n = t * per + i2;
      if (n < N) { start[n] = run; cur[n] = run; run += cnt[n]; }
    }
    if (t == 63) start[N] = inc;
  }
  {                                                                  // rank: MRS_THREADS / N threads per node, a slice of the others each
    const int parts = MRS_THREADS / N, n = t % N, part = t / N;
    const int per = (N + parts - 1) / parts;
    const int m0 = part * per, m1 = min(N, m0 + per);
    const int dn = cnt[n];
    int above = 0;
    for (int m = m0; m < m1; ++m) { const int dm = cnt[m]; above += (dm > dn || (dm == dn && m < n)) ? 1 : 0; }
    if (above) atomicAdd(&rank[n], above);
  }
  __syncthreads();
  if (t < N) perm[rank[t]] = t;
  __syncthreads();
  // ---- items by rank: wave w takes the groups w and 31 - w of 64 / CV consecutive ranks; 8 dy channels = 16 du elements of one node.
  // The loads fly while the reversed edge list is filled.
  const int CV = C >> 3, w = t >> 6, l = t & 63;
  const int spw = 64 / CV;                                           // nodes per wave item (CV <= 64: C <= 512)
  int node[2];
  u32x4 g[2][2];
  uint32_t ab[2][2];
  const int c = (l % CV) << 3;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int gidx = it == 0 ? w : 31 - w;
    const int n = perm[gidx * spw + l / CV];
    node[it] = n;
    const u32x4* gp = reinterpret_cast<const u32x4*>(du + (row0 + n) * 2 * C + 2 * c);
    g[it][0] = gp[0];
    g[it][1] = gp[1];
    const uint2 a2 = *reinterpret_cast<const uint2*>(argmax + (row0 + n) * C + c);
    ab[it][0] = a2.x; ab[it][1] = a2.y;
  }
  for (int e = t; e < E; e += MRS_THREADS) {
    const int tg = min(max(nbb[e], 0), N - 1);
    const int p = atomicAdd(&cur[tg], 1);
    const int m = e / k;
    src[p] = ((m << cshift) << 8) | (e - m * k);
  }
  // own-row term du_even - du_odd (registers), du_odd and the arg-max bytes (LDS)
  float own[2][8];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int mo = (node[it] << cshift) + c;
    u32x4 od;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int d = 0; d < 4; d += 2) {
        const uint32_t w0 = g[it][h2][d], w1 = g[it][h2][d + 1];       // (even, odd) pairs of channels 4 h2 + d, + 1
        own[it][4 * h2 + d] = __uint_as_float(w0 << 16) - __uint_as_float(w0 & 0xffff0000u);
        own[it][4 * h2 + d + 1] = __uint_as_float(w1 << 16) - __uint_as_float(w1 & 0xffff0000u);
        od[2 * h2 + d / 2] = (w0 >> 16) | (w1 & 0xffff0000u);
      }
    *reinterpret_cast<u32x4*>(odd + 2 * mo) = od;
    *reinterpret_cast<uint2*>(am + mo) = uint2{ab[it][0], ab[it][1]};
  }
  __syncthreads();
  const uint32_t zero = 0u;
  bf16x8 o8s[BNS ? 2 : 1];                      // the rounded dy of both items: the column sums are taken after the gather, so that
                                                // their per-channel constants are not live across it (60 -> 107 VGPRs = half the occupancy otherwise)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int n = node[it];
    f32x2 v[4] = {f32x2{own[it][0], own[it][1]}, f32x2{own[it][2], own[it][3]}, f32x2{own[it][4], own[it][5]},
                  f32x2{own[it][6], own[it][7]}};
    const int p1 = start[n + 1];
    int p = start[n];
    for (; p + 1 < p1; p += 2) {                   // two edges per trip: both id -> row read chains in flight together
      const uint32_t sa = (uint32_t)src[p], sb = (uint32_t)src[p + 1];
      const uint32_t ma = (sa >> 8) + c, mb = (sb >> 8) + c;
      const u32x4 oa = *reinterpret_cast<const u32x4*>(odd + 2 * ma);
      const uint2 aa = *reinterpret_cast<const uint2*>(am + ma);
      const u32x4 ob = *reinterpret_cast<const u32x4*>(odd + 2 * mb);
      const uint2 a_b = *reinterpret_cast<const uint2*>(am + mb);
      mrs_edge(v, oa, aa.x, aa.y, sa, zero);
      mrs_edge(v, ob, a_b.x, a_b.y, sb, zero);
    }
    if (p < p1) {
      const uint32_t s2 = (uint32_t)src[p];
      const uint32_t mo = (s2 >> 8) + c;
      const u32x4 o = *reinterpret_cast<const u32x4*>(odd + 2 * mo);
      const uint2 a2 = *reinterpret_cast<const uint2*>(am + mo);
      mrs_edge(v, o, a2.x, a2.y, s2, zero);
    }
    const float r8[8] = {v[0][0], v[0][1], v[1][0], v[1][1], v[2][0], v[2][1], v[3][0], v[3][1]};
    bf16x8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (__bf16)r8[e];
    // (uniform 64-bit base + 32-bit lane offset: a per-lane 64-bit address lived across the gather loop and spilled at 64 VGPRs)
    *reinterpret_cast<bf16x8*>(dy + row0 * C + (unsigned)(n * C + c)) = o8;
    if constexpr (BNS) o8s[it] = o8;
  }
  if constexpr (BNS) {
    float sum0[8], sum1[8];
    const bool masked = bn.slope != 1.f;         // uniform: a BatchNorm without an activation behind it needs no mask
    u32x4 rq[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) rq[it] = *reinterpret_cast<const u32x4*>(bn.r + row0 * bn.ldr + (unsigned)(node[it] * (int)bn.ldr + c));
    // four channels at a time: the 16 per-channel constants of a half are dead before the next half loads its own (all eight at once
    // were 64 VGPRs + 2 spilled at the 64-register budget of a 1 024-thread workgroup that shares its CU)
#pragma unroll
    for (int h4 = 0; h4 < 8; h4 += 4) {
      float bsc[4], bsh[4], bmu[4], bis[4];
      load_channels<4>(bn.mean, c + h4, bmu);
      load_channels<4>(bn.invstd, c + h4, bis);
      if (masked) { load_channels<4>(bn.scale, c + h4, bsc); load_channels<4>(bn.shift, c + h4, bsh); }
#pragma unroll
      for (int e = 0; e < 4; ++e) { sum0[h4 + e] = 0.f; sum1[h4 + e] = 0.f; if (!masked) { bsc[e] = 1.f; bsh[e] = 0.f; } }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const bf16x8 x8 = __builtin_bit_cast(bf16x8, rq[it]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dyv = (float)o8s[it][h4 + e], xx = (float)x8[h4 + e];
          float gg = dyv;
          if (masked) gg = (bsc[e] * xx + bsh[e]) > 0.f ? dyv : dyv * bn.slope;
          sum0[h4 + e] += gg;
          sum1[h4 + e] += gg * ((xx - bmu[e]) * bis[e]);
        }
      }
    }
    // lanes l, l + CV, ... hold the same 8 channels: butterfly over the lane bits above log2(CV), then one row per wave in LDS
    // (aliased onto du_odd once every wave has left the gather), waves 8-15 add into the rows of waves 0-7, 2 C threads finish.
    for (int o = CV; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sum0[e] += __shfl_xor(sum0[e], o, 64);
        sum1[e] += __shfl_xor(sum1[e], o, 64);
      }
    }
    __syncthreads();
    float* const red = reinterpret_cast<float*>(odd);                  // [8 rows][2][C]
    float* const mine = red + ((w & 7) * 2) * C + c;
    if (w < 8 && l < CV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { mine[e] = sum0[e]; mine[C + e] = sum1[e]; }
    }
    __syncthreads();
    if (w >= 8 && l < CV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { mine[e] += sum0[e]; mine[C + e] += sum1[e]; }
    }
    __syncthreads();
    if (t < 2 * C) {
      const int which = t >= C ? 1 : 0, cc = t - which * C;
      float a = 0.f;
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) a += red[(r8 * 2 + which) * C + cc];
      bn.partial[which * bn.plane + (long)b * C + cc] = a;
    }
  }
}

// batched_index_select of the reference (torch_nn.py:79-98), in the reference's own layouts: x (B, C, N) fp32,
// idx (B, N, k) clip-local -> out (B, C, N, k) contiguous, out[b,c,n,j] = x[b,c,idx[b,n,j]]. The hot path never
// materialises this tensor (mr_fwd_kernel gathers while it aggregates); the symbol exists for drop-in callers.
__global__ __launch_bounds__(256) void bis_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx, int C,
                                                      int N, int Nq, int k, long total, float* __restrict__ out) {
  const long nk = (long)Nq * k;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long e = i % nk, bc = i / nk;
    const long b = bc / C;
    out[i] = x[bc * N + min(max(idx[b * nk + e], 0), N - 1)];
  }
}
// backward: dx[b,c,idx[b,n,j]] += dout[b,c,n,j]  (dx zeroed by the caller)
__global__ __launch_bounds__(256) void bis_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                      int C, int N, int Nq, int k, long total, float* __restrict__ dx) {
  const long nk = (long)Nq * k;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long e = i % nk, bc = i / nk;
    const long b = bc / C;
    atomicAdd(dx + bc * N + min(max(idx[b * nk + e], 0), N - 1), dout[i]);
  }
}

}  // namespace

extern "C" int nsid_batched_index_select_fwd(const float* x, const int32_t* idx, int B, int C, int N, int Nq, int k,
                                             float* out, void* stream) {
  NSID_REQUIRE(x && idx && out && B > 0 && C > 0 && N > 0 && Nq > 0 && k > 0);
  const long total = (long)B * C * Nq * k;
  NSID_LAUNCH(bis_fwd_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
              static_cast<hipStream_t>(stream), x, idx, C, N, Nq, k, total, out);
  return nsid_launch_status();
}
extern "C" int nsid_batched_index_select_bwd(const float* dout, const int32_t* idx, int B, int C, int N, int Nq, int k,
                                             float* dx, void* stream) {
  NSID_REQUIRE(dout && idx && dx && B > 0 && C > 0 && N > 0 && Nq > 0 && k > 0);
  const long total = (long)B * C * Nq * k;
  NSID_LAUNCH(bis_bwd_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
              static_cast<hipStream_t>(stream), dout, idx, C, N, Nq, k, total, dx);
  return nsid_launch_status();
}

extern "C" int nsid_mr_aggregate_fwd(const void* r, int ldr, const float* scale, const float* shift,
                                     const int32_t* idx, int B, int N, int C, int k, void* u, uint8_t* argmax,
                                     int dtype, void* stream) {
  NSID_REQUIRE(r && idx && u && B > 0 && N > 0 && C > 0 && k > 0 && k <= 255 && NSID_DTYPE_OK(dtype));
  const int nv = dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(C % nv == 0 && ldr % nv == 0 && ldr >= C && nsid_aligned16(r) && nsid_aligned16(u));
  NSID_REQUIRE((scale == nullptr) == (shift == nullptr));
  // one workgroup per clip with the clip in LDS, whenever a clip fits (every stage of the GraFP encoder: N*C = 16384)
  const bool use_lds = nsid_tune(NSID_T_mr_grid_stride) == 0;
  const size_t clip_bytes = (size_t)N * C * (dtype == NSID_BF16 ? 2 : 4);
  if (use_lds && clip_bytes <= 64 * 1024 && (size_t)N * k * 4 <= 64 * 1024) {
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(mr_fwd_key_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(mr_fwd_lds_kernel<float>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(mr_fwd_lds_kernel<__bf16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
        return NSID_ELAUNCH;
      configured = true;
    }
    const int force_split = (int)nsid_tune(NSID_T_mr_split);
    nsid_count(NSID_C_mr_fwd_lds);
    int split = force_split > 0 ? force_split : (B <= 256 ? 2 : 1);
    while (split > 1 && (C / nv) % split != 0) split >>= 1;
    // many neighbours, bf16 storage: the integer-key search (tuning key mr_key_min_k: smallest k that takes it, 0 = never)
    const long key_min = nsid_tune(NSID_T_mr_key_min_k);
    if (dtype == NSID_BF16 && key_min > 0 && k >= key_min) {
      nsid_count(NSID_C_mr_fwd_key);
      NSID_LAUNCH(mr_fwd_key_kernel, dim3(B * split), dim3(MRF_THREADS), clip_bytes / split + (size_t)N * k * 4, static_cast<hipStream_t>(stream),
                  static_cast<const __bf16*>(r), (long)ldr, scale, shift, idx, N, C, k, static_cast<__bf16*>(u), argmax, split);
      return nsid_launch_status();
    }
    NSID_DISPATCH_DTYPE(dtype, T, {
      NSID_LAUNCH((mr_fwd_lds_kernel<T>), dim3(B * split), dim3(MRF_THREADS), clip_bytes / split + (size_t)N * k * 4, static_cast<hipStream_t>(stream),
                  static_cast<const T*>(r), (long)ldr, scale, shift, idx, N, C, k, static_cast<T*>(u), argmax, split);
    });
    return nsid_launch_status();
  }
  const long total = (long)B * N * (C / nv);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  nsid_count(NSID_C_mr_fwd_grid);
  NSID_DISPATCH_DTYPE(dtype, T, {
    NSID_LAUNCH((mr_fwd_kernel<T>), dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                static_cast<const T*>(r), (long)ldr, scale, shift, idx, (long)B * N, N, C, k, static_cast<T*>(u), argmax);
  });
  return nsid_launch_status();
}

// the degree-ranked gather where it applies (bf16 storage, the encoder's clip size, k >= tuning key mr_bwd_sorted_min_k); 1 otherwise
static int mrs_launch(const void* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C, int k, void* dy, int dtype,
                      const MrsBn* bn, void* stream) {
  const long sorted_min = nsid_tune(NSID_T_mr_bwd_sorted_min_k);
  const size_t sbytes = (size_t)3 * 16384 + ((size_t)5 * N + 1 + (size_t)N * k) * sizeof(int) + 16;
  if (!(dtype == NSID_BF16 && sorted_min > 0 && k >= sorted_min && (long)N * C == 16384 && (C & (C - 1)) == 0 && C >= 64 && C <= 512 &&
        sbytes <= 160 * 1024))
    return 1;
  static bool sconfigured = false;
  if (!sconfigured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_sorted_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_sorted_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    sconfigured = true;
  }
  int cshift = 0;
  while ((1 << cshift) < C) ++cshift;
  nsid_count(NSID_C_mr_bwd_sorted);
  if (bn != nullptr)
    NSID_LAUNCH(mr_bwd_sorted_kernel<true>, dim3(B), dim3(MRS_THREADS), sbytes, static_cast<hipStream_t>(stream),
                static_cast<const __bf16*>(du), idx, argmax, N, C, cshift, k, static_cast<__bf16*>(dy), *bn);
  else
    NSID_LAUNCH(mr_bwd_sorted_kernel<false>, dim3(B), dim3(MRS_THREADS), sbytes, static_cast<hipStream_t>(stream),
                static_cast<const __bf16*>(du), idx, argmax, N, C, cshift, k, static_cast<__bf16*>(dy), MrsBn{});
  return nsid_launch_status();
}

namespace {
// Clips that do not fit LDS in fp32 (N * C = 32 768 at the default encoder size, strict-fp32 mode only): the same sum as a scatter with
// global fp32 atomics in two launches -- dy = du_even - du_odd, then every (m, c) adds du_odd[m, c] to its arg-max neighbour's row.
// ---- clips beyond the LDS in bf16 storage (N * C > ~50 000: a cfg with more than 256 nodes): the same reversed-graph gather as
// mr_bwd_kernel with only the edge list in LDS; du_odd and the arg-max bytes of the source rows come from L2. One read of du for the
// own-row term, k reads of (du_odd, argmax) chunks per node on average. A correctness path (no timed configuration reaches it).
template <typename T>
__global__ __launch_bounds__(MRB_THREADS) void mr_bwd_big_kernel(const T* __restrict__ du, const int32_t* __restrict__ idx,
                                                                 const uint8_t* __restrict__ argmax, int N, int C, int k,
                                                                 T* __restrict__ dy) {
  extern __shared__ __attribute__((aligned(16))) char mrg_smem[];
  constexpr int NV = Chunk<T>::N;
  int* cnt = reinterpret_cast<int*>(mrg_smem);                      // [N]   in-degree, then fill cursor
  int* start = cnt + N;                                             // [N+1]
  int* src = start + N + 1;                                         // [N*k] (m << 8) | j
  const int b = blockIdx.x, t = threadIdx.x;
  const long row0 = (long)b * N;
  const int CV = C / NV, total = N * CV, E = N * k;
  const T* dub = du + row0 * (2L * C);
  const uint8_t* amb = argmax + row0 * C;
  const int32_t* nbb = idx + row0 * k;
  for (int n = t; n < N; n += MRB_THREADS) cnt[n] = 0;
  __syncthreads();
  for (int e = t; e < E; e += MRB_THREADS) atomicAdd(&cnt[min(max(nbb[e], 0), N - 1)], 1);
  __syncthreads();
  if (t < 64) {                                              // exclusive scan of cnt[0..N) by one wave
    const int per = (N + 63) / 64;
    int loc = 0;
    for (int i2 = 0; i2 < per; ++i2) { const int n = t * per + i2; if (n < N) loc += cnt[n]; }
    int inc = loc;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(inc, o2, 64); if (t >= o2) inc += v; }
    int run = inc - loc;
    for (int i2 = 0; i2 < per; ++i2) {
      const int n = t * per + i2;
      if (n < N) { start[n] = run; const int c2 = cnt[n]; cnt[n] = run; run += c2; }
    }
    if (t == 63) start[N] = inc;
  }
  __syncthreads();
  for (int e = t; e < E; e += MRB_THREADS) {
    const int tg = min(max(nbb[e], 0), N - 1);
    const int p = atomicAdd(&cnt[tg], 1);
    src[p] = ((e / k) << 8) | (e % k);
  }
  __syncthreads();
  for (int q = t; q < total; q += MRB_THREADS) {
    const int n = q / CV, c = (q % CV) * NV;
    float g0[NV], g1[NV], v[NV];
    Chunk<T>::load(dub + (long)q * 2 * NV, g0);
    Chunk<T>::load(dub + (long)q * 2 * NV + NV, g1);
#pragma unroll
    for (int e = 0; e < NV / 2; ++e) {
      v[e] = g0[2 * e] - g0[2 * e + 1];
      v[NV / 2 + e] = g1[2 * e] - g1[2 * e + 1];
    }
    const int p1 = start[n + 1];
    for (int p = start[n]; p < p1; ++p) {
      const int s2 = src[p];
      const int m = s2 >> 8, jj = s2 & 255;
      Chunk<T>::load(dub + ((long)m * C + c) * 2, g0);
      Chunk<T>::load(dub + ((long)m * C + c) * 2 + NV, g1);
      const uint8_t* a = amb + (long)m * C + c;
#pragma unroll
      for (int e = 0; e < NV; e += 4) {
        const uint32_t a4 = *reinterpret_cast<const uint32_t*>(a + e);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int ch = e + x;
          const float o = ch < NV / 2 ? g0[2 * ch + 1] : g1[2 * (ch - NV / 2) + 1];
          v[ch] += (int)((a4 >> (8 * x)) & 255u) == jj ? o : 0.f;
        }
      }
    }
    Chunk<T>::store(dy + row0 * C + (long)q * NV, v);
  }
}

__global__ __launch_bounds__(256) void mr_bwd_own_kernel(const float* __restrict__ du, long total, float* __restrict__ dy) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) dy[i] = du[2 * i] - du[2 * i + 1];
}
__global__ __launch_bounds__(256) void mr_bwd_scatter_kernel(const float* __restrict__ du, const int32_t* __restrict__ idx,
                                                             const uint8_t* __restrict__ argmax, int N, int C, int k, long total,
                                                             float* __restrict__ dy) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C;                       // global row
    const int c = (int)(i - m * C);
    const long b = m / N;
    const int j = min((int)argmax[i], k - 1);
    const int tg = min(max(idx[m * k + j], 0), N - 1);
    atomicAdd(dy + (b * N + tg) * C + c, du[2 * i + 1]);
  }
}
}  // namespace

extern "C" int nsid_mr_aggregate_bwd(const void* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C,
                                     int k, void* dy, int dtype, void* stream) {
  NSID_REQUIRE(du && idx && argmax && dy && B > 0 && N > 0 && C > 0 && k > 0 && k <= 255 && NSID_DTYPE_OK(dtype));
  NSID_REQUIRE(C % (dtype == NSID_BF16 ? 8 : 4) == 0 && nsid_aligned16(du) && nsid_aligned16(dy));
  const size_t esz = dtype == NSID_BF16 ? 2 : 4;
  const size_t bytes = (size_t)N * C * (esz + 1) + ((size_t)2 * N + 1 + (size_t)N * k) * sizeof(int) + 16;
  if (bytes > 160 * 1024 && dtype == NSID_F32) {
    const long total = (long)B * N * C;
    const int grid = (int)std::min<long>((total + 255) / 256, 4096);
    NSID_LAUNCH(mr_bwd_own_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(du), total,
                static_cast<float*>(dy));
    NSID_LAUNCH(mr_bwd_scatter_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const float*>(du), idx,
                argmax, N, C, k, total, static_cast<float*>(dy));
    return nsid_launch_status();
  }
  if (bytes > 160 * 1024) {                  // bf16 storage, a clip beyond the LDS: the edge list alone stays there
    const size_t gbytes = ((size_t)2 * N + 1 + (size_t)N * k) * sizeof(int);
    NSID_REQUIRE(gbytes <= 160 * 1024 && C % 8 == 0 && N < (1 << 23));
    static bool big_configured = false;
    if (!big_configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_big_kernel<__bf16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return NSID_ELAUNCH;
      big_configured = true;
    }
    NSID_LAUNCH((mr_bwd_big_kernel<__bf16>), dim3(B), dim3(MRB_THREADS), gbytes, static_cast<hipStream_t>(stream),
                static_cast<const __bf16*>(du), idx, argmax, N, C, k, static_cast<__bf16*>(dy));
    return nsid_launch_status();
  }
  NSID_REQUIRE(C % 4 == 0 && ((size_t)N * C * (esz + 1)) % 4 == 0);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_kernel<float>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_kernel<__bf16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  {
    const int rc = mrs_launch(du, idx, argmax, B, N, C, k, dy, dtype, nullptr, stream);
    if (rc != 1) return rc;
  }
  NSID_DISPATCH_DTYPE(dtype, T, {
    NSID_LAUNCH((mr_bwd_kernel<T>), dim3(B), dim3(MRB_THREADS), bytes, static_cast<hipStream_t>(stream),
                static_cast<const T*>(du), idx, argmax, N, C, k, static_cast<T*>(dy));
  });
  return nsid_launch_status();
}

// nsid_mr_aggregate_bwd + the BatchNorm-backward column sums of the layer whose output the aggregation read (MrsBn above): one partial
// row per clip. NSID_EINVAL outside the degree-ranked form (the caller then runs nsid_mr_aggregate_bwd and nsid_bn_bwd_reduce).
extern "C" int nsid_mr_aggregate_bwd_bn(const void* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C, int k, void* dy,
                                        const void* bn_r, int bn_ldr, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                        const float* bn_invstd, int bn_act, float* partial, int dtype, void* stream) {
  NSID_REQUIRE(du && idx && argmax && dy && bn_r && bn_scale && bn_shift && bn_mean && bn_invstd && partial && B > 0 && N > 0 && C > 0 &&
               k > 0 && k <= 255 && dtype == NSID_BF16);
  NSID_REQUIRE(C % 8 == 0 && bn_ldr % 8 == 0 && bn_ldr >= C && nsid_aligned16(du) && nsid_aligned16(dy) && nsid_aligned16(bn_r) &&
               nsid_aligned16(bn_scale) && nsid_aligned16(bn_shift) && nsid_aligned16(bn_mean) && nsid_aligned16(bn_invstd));
  NSID_REQUIRE(bn_act == NSID_ACT_NONE || bn_act == NSID_ACT_RELU || bn_act == NSID_ACT_LEAKY);
  const float slope = bn_act == NSID_ACT_RELU ? 0.f : (bn_act == NSID_ACT_LEAKY ? 0.2f : 1.f);
  MrsBn bn{static_cast<const __bf16*>(bn_r), bn_ldr, bn_scale, bn_shift, bn_mean, bn_invstd, slope, partial, (long)B * C};
  const int rc = mrs_launch(du, idx, argmax, B, N, C, k, dy, dtype, &bn, stream);
  return rc == 1 ? NSID_EINVAL : rc;
}
