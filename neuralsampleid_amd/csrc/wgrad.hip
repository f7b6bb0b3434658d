// Weight-gradient GEMM of the bf16 path for wide layers: dW[Nout, K] += dY[M, Nout]^T f(X[M, K])  (fp32 atomics).
//
// What bounds this op on MI355X (measured, tools/gemm_bench.py):
//   * L2->LDS bytes of re-reading the operand panels, 2*M*Nout*K*(1/BM + 1/BN): 268 MB for the 16384x1024x256 layer with
//     64x64 tiles, sustained at ~11 TB/s = 24 us — not HBM (42 MB), not MFMA;
//   * the fp32 atomics of the split reduction: workgroups x tile bytes, issued at one 256-B wave-instruction per ~50 ns per
//     CU (~1.3 TB/s chip-wide) = ~1 us per MB; they overlap the main loop only when other workgroups are resident.
// A variant that kept 64x64 tiles but removed every workgroup barrier and 3/4 of the atomics (row chunk split over the
// waves, per-wave LDS slices, LDS reduction of the four partial tiles) ran at the SAME speed as the first form
// (gemm_kernel<64,64,false,false>): the first bullet decides. This file is the 128x128-tile, 8-wave form, used where it wins
// (>= 64 tiles: the stage-3 FFN layers); it halves the L2 term, its main loop alone takes 15 us on the layer above, but with
// one 8-wave workgroup per CU the 16 MB of atomics are exposed (+11 us), so mid-size layers stay on the first form.
#include <cstdlib>
#include "nsid_common.h"

namespace {

struct WgArgs {
  const __bf16* A; long lda; long a_goff;      // dY [R][i], i contiguous
  const __bf16* B; long ldb; long b_goff;      // X  [R][j], j contiguous
  float* C; long ldc; long c_goff;             // dW [i][j] fp32
  int R;                                       // rows (M)
  int rchunk;                                  // rows per workgroup (multiple of 256)
  int tiles_j;                                 // K / tile
  const float* b_scale; const float* b_shift; float b_slope; long b_aff_goff;
};

// ------------------------------------------------------------------------------------------------------------------
// Third form, for Nout % 128 == 0 and K % 128 == 0: the weight-gradient GEMMs of the wide layers are bound by the
// L2->LDS traffic of re-reading the operand panels (2*M*Nout*K*(1/BM + 1/BN) bytes: 268 MB for the 16384x1024x256 layer
// with 64x64 tiles at ~11 TB/s), not by HBM, barriers or atomics (the second form above removed the latter two and ran
// at the same speed). 128x128 tiles halve that traffic; to keep the chip full with few, long row chunks (atomic bytes =
// splits x Nout x K x 4) the workgroup has 8 waves: quadrant = wave & 3 (64x64 each), and the two wave groups take the
// two 32-row halves of every 64-row stage; their partial tiles meet in LDS at the end.
constexpr int W3_T = 128, W3_BK = 64, W3_THREADS = 512;
// 288 B rows + 128 B more per block of 8 rows: ds_read_b64_tr_b16 serves 32 lanes together = rows {0-3} and {8-11} of the
// 16 a fragment spans; row starts r*288 + (r>>3)*128 put those 8 rows on 8 disjoint 32-byte bank ranges (gemm.hip TileGeom)
constexpr int W3_STRIDE = W3_T * 2 + 32;
constexpr int W3_SHIFT8 = 128;
constexpr int W3_OP_BYTES = W3_BK * W3_STRIDE + (W3_BK / 8) * W3_SHIFT8;   // 19456
constexpr int W3_STAGE = 2 * W3_OP_BYTES;                // A + B
constexpr int W3_LDS = 2 * W3_STAGE;                     // double buffer: 77824 B (>= the 64 KB epilogue scratch)
constexpr int W3_CHUNKS = W3_BK * W3_T * 2 / 16 / W3_THREADS;   // 2 chunks per thread per operand per stage

struct W3Stage { f32x4 a[W3_CHUNKS], b[W3_CHUNKS]; };

__device__ __forceinline__ void w3_issue(W3Stage& s, const char* A, const char* B, long lda2, long ldb2, int r0) {
#pragma unroll
  for (int q = 0; q < W3_CHUNKS; ++q) {
    const int idx = threadIdx.x + W3_THREADS * q, row = idx >> 4, ch = idx & 15;   // 16 chunks per 128-column row
    s.a[q] = *reinterpret_cast<const f32x4*>(A + (long)(r0 + row) * lda2 + ch * 16);
    s.b[q] = *reinterpret_cast<const f32x4*>(B + (long)(r0 + row) * ldb2 + ch * 16);
  }
}

template <bool BAFF>
__device__ __forceinline__ void w3_commit(char* lds, const W3Stage& s, float slope, const f32x4* csc, const f32x4* csh) {
#pragma unroll
  for (int q = 0; q < W3_CHUNKS; ++q) {
    const int idx = threadIdx.x + W3_THREADS * q, row = idx >> 4, ch = idx & 15;
    const int ro = row * W3_STRIDE + (row >> 3) * W3_SHIFT8;
    *reinterpret_cast<f32x4*>(lds + ro + ch * 16) = s.a[q];
    f32x4 raw = s.b[q];
    if (BAFF) {
      const bf16x8 h = __builtin_bit_cast(bf16x8, raw);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {           // packed fp32 math; slope in [0, 1]: max(v, v*slope) == (v < 0 ? v*slope : v)
        const f32x2 v = f32x2{csc[e >> 2][e & 3], csc[e >> 2][(e & 3) + 1]} * f32x2{(float)h[e], (float)h[e + 1]} +
                        f32x2{csh[e >> 2][e & 3], csh[e >> 2][(e & 3) + 1]};
        const f32x2 w = v * slope;
        o[e] = (__bf16)fmaxf(v[0], w[0]);
        o[e + 1] = (__bf16)fmaxf(v[1], w[1]);
      }
      raw = __builtin_bit_cast(f32x4, o);
    }
    *reinterpret_cast<f32x4*>(lds + W3_OP_BYTES + ro + ch * 16) = raw;
  }
}

__device__ __forceinline__ bf16x8 w3_frag(const char* lds, int row0, int col0, int lr, int rq) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
  const char* base = lds + (row0 + 8 * rq + (lr >> 2)) * W3_STRIDE + ((row0 >> 3) + rq) * W3_SHIFT8 + (col0 + 4 * (lr & 3)) * 2;
  const bf16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(base));
  const bf16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(base + 4 * W3_STRIDE));
  return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool BAFF>
__device__ __forceinline__ void wgrad3_body(const WgArgs& p, const int split, const int tile, const int g) {
  __shared__ __attribute__((aligned(16))) char lds_raw[W3_LDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, rq = lane >> 4;
  const int quad = wave & 3, half = wave >> 2;
  const int wm0 = (quad >> 1) * 64, wn0 = (quad & 1) * 64;
  const int ti = tile / p.tiles_j, tj = tile % p.tiles_j;
  const int i0 = ti * W3_T, j0 = tj * W3_T;
  const char* A = reinterpret_cast<const char*>(p.A + g * p.a_goff + i0);
  const char* B = reinterpret_cast<const char*>(p.B + g * p.b_goff + j0);
  const long lda2 = p.lda * 2, ldb2 = p.ldb * 2;
  const int rbeg = split * p.rchunk, rend = rbeg + p.rchunk;
  const int nst = p.rchunk / W3_BK;            // even

  f32x4 csc[2], csh[2];
  if (BAFF) {                                  // the thread's 8 columns of X are the same in every stage (512 % 16 == 0)
    const float* sc = p.b_scale + g * p.b_aff_goff + j0 + 8 * (threadIdx.x & 15);
    const float* sh = p.b_shift + g * p.b_aff_goff + j0 + 8 * (threadIdx.x & 15);
    csc[0] = *reinterpret_cast<const f32x4*>(sc); csc[1] = *reinterpret_cast<const f32x4*>(sc + 4);
    csh[0] = *reinterpret_cast<const f32x4*>(sh); csh[1] = *reinterpret_cast<const f32x4*>(sh + 4);
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const char* st) {
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = w3_frag(st, 32 * half, wm0 + 16 * a, lr, rq);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = w3_frag(st + W3_OP_BYTES, 32 * half, wn0 + 16 * b, lr, rq);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
  };

  // LDS[st & 1] holds stage st; register sets hold stages st+1 and st+2 (in flight). Branch-free body, clamped prefetch.
  W3Stage s0, s1;
  w3_issue(s0, A, B, lda2, ldb2, rbeg);
  w3_commit<BAFF>(lds_raw, s0, p.b_slope, csc, csh);
  w3_issue(s1, A, B, lda2, ldb2, min(rbeg + W3_BK, rend - W3_BK));
  __syncthreads();
  for (int st = 0; st < nst; st += 2) {
    w3_issue(s0, A, B, lda2, ldb2, min(rbeg + (st + 2) * W3_BK, rend - W3_BK));
    __builtin_amdgcn_sched_barrier(0);
    compute(lds_raw);
    __builtin_amdgcn_sched_barrier(0);
    w3_commit<BAFF>(lds_raw + W3_STAGE, s1, p.b_slope, csc, csh);
    __syncthreads();
    w3_issue(s1, A, B, lda2, ldb2, min(rbeg + (st + 3) * W3_BK, rend - W3_BK));
    __builtin_amdgcn_sched_barrier(0);
    compute(lds_raw + W3_STAGE);
    __builtin_amdgcn_sched_barrier(0);
    w3_commit<BAFF>(lds_raw, s0, p.b_slope, csc, csh);
    __syncthreads();
  }

  // ---- the second wave group hands its partial quadrants over through LDS; the first adds and issues the atomics
  float* scratch = reinterpret_cast<float*>(lds_raw);          // [4 quadrants][64 regs][64 lanes] = 64 KB
  if (half == 1) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[(quad * 64 + a * 16 + b * 4 + r) * 64 + lane] = acc[a][b][r];
  }
  __syncthreads();
  if (half == 0) {
    float* C = p.C + g * p.c_goff;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[a][b][r] + scratch[(quad * 64 + a * 16 + b * 4 + r) * 64 + lane];
          atomicAdd(C + (long)(i0 + wm0 + 16 * a + 4 * rq + r) * p.ldc + (j0 + wn0 + 16 * b + lr), v);
        }
  }
}

template <bool BAFF>
__global__ __launch_bounds__(W3_THREADS, 2) void wgrad3_kernel(const WgArgs p) {
  wgrad3_body<BAFF>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// many problems in one launch (nsid_linear_bwd_weight_grouped, see gemm.hip wgrad_grouped_kernel): the same body, its problem from the
// table in the kernel arguments. In the deferred phase of a step nothing else runs, so the 8-wave workgroup with 78 KB of LDS -- which
// loses inside the two-stream chains because it owns its CU -- is the form to use: a third fewer L2 -> LDS bytes per flop than 128x64.
template <bool BAFF>
__global__ __launch_bounds__(W3_THREADS, 2) void wgrad3_grouped_kernel(const WgGroupArgs ga) {
  const int total = ga.wg0[ga.n];
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    int pi, split, bid, g, seg;
    if (wgg_decode(ga, w, pi, split, bid, g, seg)) {
      const WgProb& q = ga.prob[pi];
      WgArgs p;
      p.A = static_cast<const __bf16*>(q.A[seg]); p.lda = q.lda; p.a_goff = q.I;
      p.B = static_cast<const __bf16*>(q.B[seg]); p.ldb = q.ldb; p.b_goff = q.J;
      p.C = q.C; p.ldc = q.J; p.c_goff = (long)q.I * q.J;
      p.R = q.R; p.rchunk = q.rchunk; p.tiles_j = q.J / W3_T;
      p.b_scale = q.bsc[seg]; p.b_shift = q.bsh[seg]; p.b_slope = q.slope; p.b_aff_goff = q.J;
      wgrad3_body<BAFF>(p, split, bid, g);
    }
    if (gridDim.x < total) __syncthreads();       // a capped launch: the epilogue's LDS scratch is read before the next item stages
  }
}

}  // namespace

int nsid_wgrad3_grouped_launch(const WgGroupArgs& ga, int grid, bool affine, hipStream_t stream) {
  if (affine) NSID_LAUNCH((wgrad3_grouped_kernel<true>), dim3(grid), dim3(W3_THREADS), 0, stream, ga);
  else NSID_LAUNCH((wgrad3_grouped_kernel<false>), dim3(grid), dim3(W3_THREADS), 0, stream, ga);
  return nsid_launch_status();
}

// returns NSID_OK when launched, 1 when the shape is outside this kernel's preconditions (the caller then runs the
// first form), or an error code
int nsid_wgrad2_launch(const void* dout, int ldd, const void* x, int ldx, float* dw, int M, int Nout, int K, int groups,
                       const float* in_scale, const float* in_shift, float slope, hipStream_t stream) {
  if (ldd % 8 != 0 || ldx % 8 != 0) return 1;
  const long w3_wgs = nsid_tune(NSID_T_w3_wgs);
  const long w3_min_tiles = nsid_tune(NSID_T_w3_min_tiles);
  const long tiles3 = (long)(Nout / W3_T) * (K / W3_T) * groups;
  if (w3_wgs > 0 && Nout % W3_T == 0 && K % W3_T == 0 && M % 128 == 0 && tiles3 >= w3_min_tiles) {
    int rc3 = 0;
    const int cands3[] = {4096, 2048, 1024, 512, 256, 128};
    for (int rc : cands3) {
      if (M % rc != 0) continue;
      rc3 = rc;
      if (tiles3 * (M / rc) >= w3_wgs) break;
    }
    if (rc3 != 0) {
      WgArgs p{};
      p.A = static_cast<const __bf16*>(dout); p.lda = ldd; p.a_goff = Nout;
      p.B = static_cast<const __bf16*>(x); p.ldb = ldx; p.b_goff = K;
      p.C = dw; p.ldc = K; p.c_goff = (long)Nout * K;
      p.R = M; p.rchunk = rc3; p.tiles_j = K / W3_T;
      p.b_scale = in_scale; p.b_shift = in_shift; p.b_slope = slope; p.b_aff_goff = K;
      dim3 grid(M / rc3, (Nout / W3_T) * (K / W3_T), groups);
      nsid_count(NSID_C_wgrad3);
      if (in_scale != nullptr) NSID_LAUNCH((wgrad3_kernel<true>), grid, dim3(W3_THREADS), 0, stream, p);
      else NSID_LAUNCH((wgrad3_kernel<false>), grid, dim3(W3_THREADS), 0, stream, p);
      return nsid_launch_status();
    }
  }
  return 1;
}
