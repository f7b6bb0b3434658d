// Eval-mode max-relative aggregation + grouped conv (+ folded BatchNorm + ReLU) in ONE launch, one workgroup per clip
// (MRConv2d.forward, encoder/gcn_lib/torch_vertex.py:19-34 + BasicConv, torch_nn.py:52-76, in eval mode; forward-only extraction):
//     v[n, g*K + j] = relu( sum_k W[g*K + j][k] * u[n, g*K + k] + b ),   u[n, 2c] = y[n, c],  u[n, 2c+1] = max_j (y[idx[n,j], c] - y[n, c])
// The interleaved tensor u (M x 2C: 134 MB written and read back per launch pair at a 2 048-clip micro-batch) is never formed: a lane
// builds the MFMA operand fragment it needs — 4 channels of its node and of the node's neighbours, gathered from the clip's LDS image,
// max-relative in fp32, rounded to bf16 exactly where the two-launch form rounds u — and feeds it to the matrix core directly.
// HBM sees y once in and v once out.
//
// Tiling: N * C = 16 384 at every stage, so a clip is always 32 KB; 4 waves share the clip's 16-node tiles. The MFMA operands are
// swapped (A = weight rows, B = nodes) so that a lane ends up with 4 CONSECUTIVE output channels of one node: bias + ReLU + bf16 pack,
// one 8-byte store into a per-group staging image, then 16-byte coalesced stores. Per group: weights -> LDS (the next group's are in
// flight meanwhile), barrier, fragments + MFMA for all of the wave's node tiles, epilogue, barrier, copy-out.
#include <algorithm>
#include "nsid_common.h"

namespace {

struct MrcArgs {
  const __bf16* y; const int32_t* idx; const __bf16* w; const float* bias; __bf16* out;
  int k;
};

// NW waves: min(NW, N/16) groups of node tiles x (NW / that) splits of a group's output tiles. DIRECT: the epilogue stores 8 bytes per
// lane straight to global memory (32 contiguous bytes per node and instruction; the group's other tiles complete the lines) instead of
// staging the group's slab in LDS: one barrier and 17-20 KB of LDS less per workgroup (more workgroups per CU).
template <int C, int NW, bool DIRECT>
__global__ __launch_bounds__(64 * NW) void mrconv_fused_kernel(const MrcArgs p) {
  constexpr int MRC_THREADS = 64 * NW;
  constexpr int N = 16384 / C;               // nodes per clip
  constexpr int K = C / 2;                   // per group: input (interleaved) channels = output channels
  constexpr int JT = K / 16, KS = K / 32;    // 16-channel output tiles, 32-deep MFMA steps per group
  constexpr int WN = (N / 16 < NW) ? N / 16 : NW;      // wave groups across the node tiles
  constexpr int JS = NW / WN;                // splits of a group's output tiles
  constexpr int NTW = N / 16 / WN;           // node tiles per wave
  constexpr int JW = JT / JS;                // output tiles per wave
  constexpr bool PAIRED = DIRECT && JW % 2 == 0;   // direct stores of 16 bytes per lane (see the epilogue)
  static_assert(JT % JS == 0 && (N / 16) % WN == 0, "whole tiles per wave");
  constexpr int SY = C * 2 + 16;             // clip image rows: consecutive nodes land on disjoint banks for the 8-byte gathers
  constexpr int SW = K * 2 + 32;             // weight image rows (fragment reads by ds_read_b128: conflict-free)
  constexpr int SO = K * 2 + 16;             // staging rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* yimg = smem;                         // [N][SY]
  char* wimg = yimg + N * SY;                // [K][SW]
  char* oimg = wimg + K * SW;                // [N][SO]  (absent when DIRECT)
  int* idxl = reinterpret_cast<int*>(oimg + (DIRECT ? 0 : N * SO));     // [N][k]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, rq = lane >> 4;
  const int wn = wave % WN, wjs = wave / WN;
  const long row0 = (long)blockIdx.x * N;
  const int k = p.k;

  // ---- the clip, its neighbour lists, and the first group's weights
  constexpr int YV = N * C / 8 / MRC_THREADS;            // 16-byte chunks of the clip per thread (8)
  constexpr int WV = K * K / 8 / MRC_THREADS > 0 ? K * K / 8 / MRC_THREADS : 1;
  constexpr int WCH = K * K / 8;                         // 16-byte chunks of one group's weights
  f32x4 wreg[WV];
  auto wload = [&](int g) {
#pragma unroll
    for (int q = 0; q < WV; ++q) {
      const int i = t + MRC_THREADS * q;
      if (WCH % MRC_THREADS == 0 || i < WCH) wreg[q] = *reinterpret_cast<const f32x4*>(p.w + (long)g * K * K + (long)i * 8);
    }
  };
  auto wstore = [&]() {
#pragma unroll
    for (int q = 0; q < WV; ++q) {
      const int i = t + MRC_THREADS * q;
      if (WCH % MRC_THREADS == 0 || i < WCH) *reinterpret_cast<f32x4*>(wimg + (i / (K / 8)) * SW + (i % (K / 8)) * 16) = wreg[q];
    }
  };
  {
    f32x4 v[YV];
#pragma unroll
    for (int q = 0; q < YV; ++q) {
      const int i = t + MRC_THREADS * q;
      v[q] = *reinterpret_cast<const f32x4*>(p.y + (row0 + i / (C / 8)) * C + (i % (C / 8)) * 8);
    }
    wload(0);
    for (int i = t; i < N * k; i += MRC_THREADS) {
      const int m = p.idx[row0 * k + i];
      idxl[i] = m < 0 ? 0 : (m >= N ? N - 1 : m);        // ids come from the caller: never read outside the clip
    }
#pragma unroll
    for (int q = 0; q < YV; ++q) {
      const int i = t + MRC_THREADS * q;
      *reinterpret_cast<f32x4*>(yimg + (i / (C / 8)) * SY + (i % (C / 8)) * 16) = v[q];
    }
  }

  for (int g = 0; g < 4; ++g) {
    wstore();                                 // (every wave left the previous group's MFMAs behind the barrier below)
    if (g + 1 < 4) wload(g + 1);
    __syncthreads();
    f32x4 acc[NTW][JW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int a = 0; a < JW; ++a) acc[nt][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = 16 * (wn + WN * nt) + lr;
      const int* nb = idxl + n * k;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // this lane's B fragment: interleaved channels 32 ks + 8 rq .. + 7 of group g = original channels c0 .. c0 + 3
        const int c0 = g * (C / 4) + 16 * ks + 4 * rq;
        const bf16x4 own = *reinterpret_cast<const bf16x4*>(yimg + n * SY + c0 * 2);
        float ys[4], best[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { ys[e] = (float)own[e]; best[e] = -__builtin_inff(); }
        for (int j = 0; j < k; ++j) {
          const bf16x4 v = *reinterpret_cast<const bf16x4*>(yimg + nb[j] * SY + c0 * 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = (float)v[e] - ys[e];
            best[e] = d > best[e] ? d : best[e];          // strict: the first maximum wins, NaN never enters (as torch.max / mr.hip)
          }
        }
        bf16x8 fb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { fb[2 * e] = own[e]; fb[2 * e + 1] = (__bf16)best[e]; }
#pragma unroll
        for (int a = 0; a < JW; ++a) {
          // PAIRED (direct stores): the A rows of tiles 2p and 2p + 1 are a permutation of the pair's 32 weight rows chosen so that
          // D row 4 rq + e of tile 2p + i is output channel 32 p + 8 rq + 4 i + e: a lane then owns 8 CONSECUTIVE channels of its node
          const int wrow = PAIRED ? 16 * wjs * JW + 32 * (a / 2) + 8 * (lr >> 2) + 4 * (a & 1) + (lr & 3) : 16 * (wjs * JW + a) + lr;
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(wimg + wrow * SW + (4 * ks + rq) * 16);
          acc[nt][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[nt][a], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: D[j][node lr] + bias, ReLU, bf16
    if constexpr (PAIRED) {
      // 8 consecutive channels of one node per lane = one 16-byte store; the four lanes of a node write 64 contiguous bytes, i.e. whole
      // 64-byte segments (the 8-byte form wrote every segment in two halves: PMC WRITE_SIZE 1.66x the output at C = 64, round 3)
#pragma unroll
      for (int pr = 0; pr < JW / 2; ++pr) {
        const int j0 = 16 * wjs * JW + 32 * pr + 8 * rq;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + g * K + j0), b1 = *reinterpret_cast<const f32x4*>(p.bias + g * K + j0 + 4);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int n = 16 * (wn + WN * nt) + lr;
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (__bf16)fmaxf(acc[nt][2 * pr][e] + b0[e], 0.f);
            o[4 + e] = (__bf16)fmaxf(acc[nt][2 * pr + 1][e] + b1[e], 0.f);
          }
          *reinterpret_cast<bf16x8*>(p.out + (row0 + n) * (2 * C) + g * K + j0) = o;
        }
      }
    } else {
    // 4 consecutive channels of one node per lane
#pragma unroll
    for (int a = 0; a < JW; ++a) {
      const int j0 = 16 * (wjs * JW + a) + 4 * rq;
      const f32x4 bj = *reinterpret_cast<const f32x4*>(p.bias + g * K + j0);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int n = 16 * (wn + WN * nt) + lr;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)fmaxf(acc[nt][a][e] + bj[e], 0.f);
        if constexpr (DIRECT) *reinterpret_cast<bf16x4*>(p.out + (row0 + n) * (2 * C) + g * K + j0) = o;
        else *reinterpret_cast<bf16x4*>(oimg + n * SO + j0 * 2) = o;
      }
    }
    }
    __syncthreads();                          // (DIRECT: every wave is done with this group's weight image)
    if constexpr (!DIRECT) {
      constexpr int OCH = N * K / 8;          // 16-byte chunks of the group's output slab
#pragma unroll
      for (int q = 0; q < OCH / MRC_THREADS; ++q) {
        const int i = t + MRC_THREADS * q;
        const int n = i / (K / 8), ch = i % (K / 8);
        *reinterpret_cast<f32x4*>(p.out + (row0 + n) * (2 * C) + g * K + ch * 8) = *reinterpret_cast<const f32x4*>(oimg + n * SO + ch * 16);
      }
    }
  }
}

// Round 6 variant (tuning key mrconv_variant bit 2): a workgroup keeps ONE group's weights in LDS and walks a range of clips.
// The per-clip form above re-stages the four 32 KB weight groups of a layer for every clip (268 MB of L2 -> LDS traffic per 2 048-clip
// launch at C = 256, against 200 MB of HBM traffic) and pays two workgroup barriers per group around a ~1 us compute phase. Here a
// group's K x K weights are staged once per workgroup, a clip contributes only the slice the group reads -- y[:, g*C/4 .. (g+1)*C/4),
// N x C/4 x 2 B = 8 KB at every stage -- prefetched into registers while the previous clip computes, and the epilogue stores straight
// to global memory (8 consecutive channels per lane, as the PAIRED form above). Grid = 4 groups x clip ranges.
struct MrcPgArgs {
  const __bf16* y; const int32_t* idx; const __bf16* w; const float* bias; __bf16* out;
  int k, B, per;          // clips per workgroup
};

template <int C, int NW>
__global__ __launch_bounds__(64 * NW) void mrconv_pg_kernel(const MrcPgArgs p) {
  constexpr int MRC_THREADS = 64 * NW;
  constexpr int N = 16384 / C, K = C / 2, CS = C / 4;      // nodes, group channels (in = out), y channels of a group's slice
  constexpr int JT = K / 16, KS = K / 32;
  constexpr int WN = (N / 16 < NW) ? N / 16 : NW, JS = NW / WN, NTW = N / 16 / WN, JW = JT / JS;
  static_assert(JT % JS == 0 && (N / 16) % WN == 0 && JW % 2 == 0, "whole tile pairs per wave");
  constexpr int SYS = CS * 2 + 16;           // slice image rows
  constexpr int SW = K * 2 + 32;             // weight image rows (fragment reads by ds_read_b128: conflict-free)
  constexpr int YV = (N * CS / 8 + MRC_THREADS - 1) / MRC_THREADS;      // 16-byte chunks of a slice per thread
  constexpr int YCH = N * CS / 8;
  constexpr int MAXK = 8;
  constexpr int IV = (N * MAXK + MRC_THREADS - 1) / MRC_THREADS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wimg = smem;                         // [K][SW]
  char* yimg = wimg + K * SW;                // [N][SYS]
  int* idxl = reinterpret_cast<int*>(yimg + N * SYS);      // [N][k]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 15, rq = lane >> 4;
  const int wn = wave % WN, wjs = wave / WN;
  const int g = blockIdx.x & 3;
  const int k = p.k;
  const int clip0 = (blockIdx.x >> 2) * p.per, clip1 = min(p.B, clip0 + p.per);
  if (clip0 >= clip1) return;

  // ---- the group's weights, once per workgroup (a form that kept them in registers as MFMA fragments -- no weight image, two slice
  // buffers, one barrier per clip -- measured 91 us per launch against 60 for this one: docs/experiments.md, round 6)
  for (int i = t; i < K * K / 8; i += MRC_THREADS)
    *reinterpret_cast<f32x4*>(wimg + (i / (K / 8)) * SW + (i % (K / 8)) * 16) =
        *reinterpret_cast<const f32x4*>(p.w + (long)g * K * K + (long)i * 8);
  // ---- register prefetch of a clip's slice and neighbour lists
  f32x4 yv[YV];
  int iv[IV];
  auto fetch = [&](int clip) {
    const long row0 = (long)clip * N;
#pragma unroll
    for (int q = 0; q < YV; ++q) {
      const int i = t + MRC_THREADS * q;
      if (YCH % MRC_THREADS == 0 || i < YCH)
        yv[q] = *reinterpret_cast<const f32x4*>(p.y + (row0 + i / (CS / 8)) * C + g * CS + (i % (CS / 8)) * 8);
    }
#pragma unroll
    for (int q = 0; q < IV; ++q) {
      const int i = t + MRC_THREADS * q;
      iv[q] = i < N * k ? p.idx[row0 * k + i] : 0;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < YV; ++q) {
      const int i = t + MRC_THREADS * q;
      if (YCH % MRC_THREADS == 0 || i < YCH) *reinterpret_cast<f32x4*>(yimg + (i / (CS / 8)) * SYS + (i % (CS / 8)) * 16) = yv[q];
    }
#pragma unroll
    for (int q = 0; q < IV; ++q) {
      const int i = t + MRC_THREADS * q;
      if (i < N * k) { const int m = iv[q]; idxl[i] = m < 0 ? 0 : (m >= N ? N - 1 : m); }
    }
  };
  fetch(clip0);
  for (int clip = clip0; clip < clip1; ++clip) {
    commit();
    __syncthreads();                          // the slice (and, the first time, the weights) are in LDS
    if (clip + 1 < clip1) fetch(clip + 1);    // lands under the compute phase
    const long row0 = (long)clip * N;
    f32x4 acc[NTW][JW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int a = 0; a < JW; ++a) acc[nt][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int n = 16 * (wn + WN * nt) + lr;
      const int* nb = idxl + n * k;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c0 = 16 * ks + 4 * rq;      // slice-local channels c0 .. c0 + 3 = interleaved channels 32 ks + 8 rq .. + 7 of the group
        const bf16x4 own = *reinterpret_cast<const bf16x4*>(yimg + n * SYS + c0 * 2);
        float ys[4], best[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { ys[e] = (float)own[e]; best[e] = -__builtin_inff(); }
        for (int j = 0; j < k; ++j) {
          const bf16x4 v = *reinterpret_cast<const bf16x4*>(yimg + nb[j] * SYS + c0 * 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = (float)v[e] - ys[e];
            best[e] = d > best[e] ? d : best[e];          // strict: the first maximum wins, NaN never enters (as torch.max / mr.hip)
          }
        }
        bf16x8 fb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { fb[2 * e] = own[e]; fb[2 * e + 1] = (__bf16)best[e]; }
#pragma unroll
        for (int a = 0; a < JW; ++a) {
          const int wrow = 16 * wjs * JW + 32 * (a / 2) + 8 * (lr >> 2) + 4 * (a & 1) + (lr & 3);      // (the PAIRED permutation)
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(wimg + wrow * SW + (4 * ks + rq) * 16);
          acc[nt][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[nt][a], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int pr = 0; pr < JW / 2; ++pr) {
      const int j0 = 16 * wjs * JW + 32 * pr + 8 * rq;
      // (re-read per clip from L1: kept in registers across the loop the 16 values cost occupancy, 59.8 -> 69.6 us per launch)
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + g * K + j0), b1 = *reinterpret_cast<const f32x4*>(p.bias + g * K + j0 + 4);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int n = 16 * (wn + WN * nt) + lr;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (__bf16)fmaxf(acc[nt][2 * pr][e] + b0[e], 0.f);
          o[4 + e] = (__bf16)fmaxf(acc[nt][2 * pr + 1][e] + b1[e], 0.f);
        }
        *reinterpret_cast<bf16x8*>(p.out + (row0 + n) * (2 * C) + g * K + j0) = o;
      }
    }
    __syncthreads();                          // every wave has read this clip's slice: the next one may be committed
  }
}

template <int C, int NW>
int mrc_launch_pg(const MrcArgs& p, int B, hipStream_t s) {
  constexpr int N = 16384 / C, K = C / 2, CS = C / 4;
  if (p.k > 8) return 1;
  const size_t bytes = (size_t)K * (K * 2 + 32) + (size_t)N * (CS * 2 + 16) + (size_t)N * p.k * 4;
  if (bytes > 160 * 1024) return 1;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mrconv_pg_kernel<C, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  const long target = std::max<long>(1, nsid_tune(NSID_T_mrconv_pg_wgs) / 4);       // clip ranges
  const int per = (int)std::max<long>(1, (B + target - 1) / target);
  const int ranges = (B + per - 1) / per;
  MrcPgArgs q{p.y, p.idx, p.w, p.bias, p.out, p.k, B, per};
  NSID_LAUNCH((mrconv_pg_kernel<C, NW>), dim3(4 * ranges), dim3(64 * NW), bytes, s, q);
  return nsid_launch_status();
}

template <int C, int NW, bool DIRECT>
int mrc_launch_v(const MrcArgs& p, int B, hipStream_t s) {
  constexpr int N = 16384 / C, K = C / 2;
  const size_t bytes = (size_t)N * (C * 2 + 16) + (size_t)K * (K * 2 + 32) + (DIRECT ? 0 : (size_t)N * (K * 2 + 16)) + (size_t)N * p.k * 4;
  if (bytes > 160 * 1024) return 1;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mrconv_fused_kernel<C, NW, DIRECT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  NSID_LAUNCH((mrconv_fused_kernel<C, NW, DIRECT>), dim3(B), dim3(64 * NW), bytes, s, p);
  return nsid_launch_status();
}

template <int C>
int mrc_launch(const MrcArgs& p, int B, hipStream_t s) {
  const int v = (int)nsid_tune(NSID_T_mrconv_variant);         // bit 0: 8 waves, bit 1: direct stores, bit 2: one group per workgroup, many clips
  // measured per launch at 2 048 clips (rocprofv3, per-clip form | this form): C = 256: 76.7 | 59.8 us; C = 128: 47.0 | 49.1; C = 64: 48.5 | 65.8
  // (a slice row is only 32 bytes there): the wide stage only, unless bit 3 asks for every width
  if ((v & 4) && (C == 256 || (v & 8))) {
    const int rc = mrc_launch_pg<C, 8>(p, B, s);
    if (rc != 1) return rc;
  }
  switch (v & 3) {
    case 0: return mrc_launch_v<C, 4, false>(p, B, s);
    case 1: return mrc_launch_v<C, 8, false>(p, B, s);
    case 2: return mrc_launch_v<C, 4, true>(p, B, s);
    default: return mrc_launch_v<C, 8, true>(p, B, s);
  }
}

}  // namespace
// y: (B*N, C) bf16 contiguous node-major features (BatchNorm already folded: plain values); idx: (B, N, k) clip-local int32;
// w: (2C, C/2) bf16 = the grouped conv's weight with its BatchNorm folded in, bias: (2C) fp32; out: (B*N, 2C) bf16.
// Returns 1 (nothing launched) outside the fused form: C in {64, 128, 256} with N * C = 16 384, k <= 64.
extern "C" int nsid_mrconv_fused_fwd(const void* y, const int32_t* idx, int B, int N, int C, int k, const void* w, const float* bias,
                                     void* out, void* stream) {
  NSID_REQUIRE(y && idx && w && bias && out && B > 0 && k > 0);
  if (!(C == 64 || C == 128 || C == 256) || (long)N * C != 16384 || k > 64) return 1;
  NSID_REQUIRE(nsid_aligned16(y) && nsid_aligned16(w) && nsid_aligned16(bias) && nsid_aligned16(out));
  MrcArgs p{static_cast<const __bf16*>(y), idx, static_cast<const __bf16*>(w), bias, static_cast<__bf16*>(out), k};
  hipStream_t s = static_cast<hipStream_t>(stream);
  // (C = 512, 139 KB of weights per group = one workgroup per CU, was tried in the weights-resident form: 166 us per launch against
  //  119 us for the aggregation + grouped GEMM pair: not taken)
  const int rc = C == 64 ? mrc_launch<64>(p, B, s) : (C == 128 ? mrc_launch<128>(p, B, s) : mrc_launch<256>(p, B, s));
  if (rc == NSID_OK) nsid_count(NSID_C_mrconv_fused);
  return rc;
}
