// Forward row GEMM on 256x256 tiles with LDS-DMA operand staging (gfx950): out[M][N] = act(X[M][K] W[N][K]^T + bias) (+ residual),
// bf16 operands in HBM, fp32 accumulation, bf16 output, optional BatchNorm partial statistics of the fp32 result.
//
// Why a second GEMM body (gemm.hip keeps every other shape): the 128x128-tile kernels stage operands global -> VGPR -> LDS and
// are bound by the bytes a CU can pull through its L2 -> LDS path (~55-70 GB/s per CU). The vendor library's plain
// bf16 product runs the FFN shapes 25-35 % faster than gemm.hip on a 256x256 macro-tile because a stage then moves HALF the
// operand bytes per flop (tools/gemm_bench.py --cold --blas). This kernel takes that tile and stages with
// global_load_lds_dwordx4 (no staging registers, no ds_write pass, no vmcnt -> ds_write dependency):
//   * 512 threads = 8 waves as 2 (rows) x 4 (columns); wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 tiles = 128 accumulator registers;
//   * the reduction advances in 32-deep sub-tiles through a ring of FOUR 32 KB LDS slots (A 256x32 + B 256x32 bf16): three
//     sub-tiles (96 KB) are in flight per CU while one is being consumed; ONE workgroup barrier per sub-tile;
//   * LDS image: [16 rows][32 k] sub-blocks of 1 KB, each written by ONE wave-instruction (LDS-DMA writes lane-linearly:
//     64 lanes x 16 B); the 16-byte chunk index is XOR-ed with 2*(row >> 3 & 1) on the SOURCE address and on the fragment
//     read, which makes every ds_read_b128 lane group {8 rows at chunk c, the other 8 at chunk c+1} hit 16 distinct 16-byte bank
//     slots (MI355X_MICROARCH.md, LDS table: conflict-free, 4 cycles per wave-read);
//   * fragments are double-buffered in registers: the 12 ds_read_b128 of sub-tile j+1 are issued behind the barrier and land under
//     the 32 MFMAs of sub-tile j;
//   * synchronisation: s_waitcnt vmcnt(N) counts this wave's own LDS-DMA groups (4 per sub-tile), then ONE s_barrier both
//     publishes sub-tile j+1 to every wave and frees slot j % 4 (every wave has its fragments of sub-tile j in registers:
//     lgkmcnt(0) before the barrier) for the LDS-DMA of sub-tile j+4.
// Preconditions (host-checked, nsid_gemm256_fwd_launch returns 1 otherwise): M % 256 == 0, N % 256 == 0, K % 128 == 0.
#include <cstdlib>
#include "nsid_common.h"

namespace {

struct G256Args {
  const __bf16* A; long lda;
  const __bf16* B; long ldb;
  __bf16* C; long ldc;
  int M, N, K;
  const float* bias;
  int relu_out;                   // out = max(acc + bias, 0)
  const __bf16* addend; long ldadd;
  float* stat; long stat_plane; long stat_ld;
  unsigned long long* trace;      // nsid_debug_gemm_trace buffer or nullptr: {start, end of main loop, end, where | prologue << 40}
};

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

#ifndef NSID_G256_ABLATE
#define NSID_G256_ABLATE 0        // diagnosis builds (tools/build_variant.sh): 1 = no LDS-DMA in the loop, 2 = no MFMA, 4 = no fragment reads
#endif
constexpr int SLOT = 32768;         // bytes per ring slot: A image [16 row-blocks][1 KB], then B image [16 row-blocks][1 KB]
constexpr int NSLOT = 4;
constexpr int OLD = 64 + 4;         // floats per row of a wave's epilogue transpose buffer

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N == 0 || N == 4 || N == 8 || N == 12, "whole LDS-DMA groups of four");
  // the builtin, not inline asm: the compiler's own wait-count bookkeeping then knows what has been waited for
  // (s_waitcnt simm16 on gfx9: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8]; the fields not meant are left at their maximum)
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (0xF << 8));
}

template <bool PAIR>
__global__ __launch_bounds__(512, 2) void gemm256_fwd_kernel(const G256Args p) {
  __shared__ __attribute__((aligned(1024))) char lds[NSLOT * SLOT];
  unsigned long long t_start = 0, t_first = 0, t_loop = 0;
  if (p.trace) t_start = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, rq = lane >> 4;

  const int tiles_n = p.N >> 8;
  int bid = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);   // blocks b, b+8 share an XCD: contiguous tile runs per L2
  const int ti = bid / tiles_n, tj = bid % tiles_n;
  const int i0 = ti << 8, j0 = tj << 8;

  // ---- LDS-DMA source addressing: waves 0-3 stage A (row-blocks 4w .. 4w+3), waves 4-7 stage B. Lane l supplies row l >> 2 of
  // the block and the LOGICAL chunk that lives at physical chunk l & 3.
  const int grow = lane >> 2, lc = (lane & 3) ^ (((grow >> 3) & 1) << 1);
  const bool stage_b = wave >= 4;
  const long ld_s = stage_b ? p.ldb : p.lda;
  // uniform 64-bit base (SGPRs) + ONE 32-bit lane offset for every LDS-DMA of the kernel: the loads take the saddr form, the row-block
  // and reduction steps are scalar adds (eight 64-bit vector addresses per stage were 16 registers the 256-register budget lacks)
  const char* ubase = reinterpret_cast<const char*>(stage_b ? p.B + (long)(j0 + (wave & 3) * 64) * p.ldb
                                                            : p.A + (long)(i0 + (wave & 3) * 64) * p.lda);
  const unsigned voff = (unsigned)(grow * (int)ld_s + lc * 8) * 2u;
  const long rb_bytes = ld_s * 32;                        // 16 rows further, in bytes
  const int dst0 = (stage_b ? 16384 : 0) + (wave & 3) * 4096;

  auto issue = [&](int j, int slot) {                     // sub-tile j (k = 32 j ..) -> ring slot
    const char* s = ubase + (long)j * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_vptr)(s + i * rb_bytes + voff), (lds_vptr)(lds + slot * SLOT + dst0 + i * 1024), 16, 0, 0);
  };

  // ---- fragment addressing
  const int lo = (lr * 64 + rq * 16) ^ (((lr >> 3) & 1) << 5);
  const char* fa_base = lds + wr * 8192 + lo;
  const char* fb_base = lds + 16384 + wc * 4096 + lo;

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 fa[2][8], fb[2][4];
  auto read_frags = [&](int set, int slot) {
    if (NSID_G256_ABLATE & 4) return;
#pragma unroll
    for (int a = 0; a < 8; ++a) fa[set][a] = *reinterpret_cast<const bf16x8*>(fa_base + slot * SLOT + a * 1024);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[set][b] = *reinterpret_cast<const bf16x8*>(fb_base + slot * SLOT + b * 1024);
  };
  auto mfma_block = [&](int set) {
    if (NSID_G256_ABLATE & 2) {
#pragma unroll
      for (int a = 0; a < 8; ++a) asm volatile("" ::"v"(fa[set][a]));
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(fb[set][b]));
      return;
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[set][a], fb[set][b], acc[a][b], 0, 0, 0);
  };

  const int J = p.K >> 5;                                 // 32-deep sub-tiles; a multiple of 4 (K % 128 == 0)
  if constexpr (!PAIR) {
    issue(0, 0);
    issue(1, 1);
    issue(2, 2);
    issue(3, 3);
    wait_vm<12>();
    __builtin_amdgcn_s_barrier();
    if (p.trace) t_first = __builtin_amdgcn_s_memrealtime();
    read_frags(0, 0);

    // one sub-tile: VM = LDS-DMA instructions of this wave that may stay in flight at the wait (the groups younger than j+1)
#define NSID_G256_STEP(U, VM, ISSUE, READ)                                                \
  do {                                                                                    \
    __builtin_amdgcn_s_waitcnt(0xC07F);           /* lgkmcnt(0) */                        \
    wait_vm<VM>();                                                                        \
    __builtin_amdgcn_s_barrier();                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    if (ISSUE) issue(jb + (U) + 4, (U));                                                  \
    if (READ) read_frags(((U) + 1) & 1, ((U) + 1) & 3);                                   \
    mfma_block((U) & 1);                                                                  \
    /* issue order: the 12 fragment reads of sub-tile j+1 first (they land under the MFMAs), then the MFMAs with one */ \
    /* LDS-DMA behind every 8th (a glds holds the wave's issue for 60+ cycles: spread, the other wave of the SIMD fills in) */ \
    if (READ) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                  \
      if (ISSUE) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                       \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)

    for (int jb = 0; jb < J - 4; jb += 4) {          // steady state: sub-tiles j+1 .. j+3 in flight, j+4 issued behind the barrier
      NSID_G256_STEP(0, 8, true, true);
      NSID_G256_STEP(1, 8, true, true);
      NSID_G256_STEP(2, 8, true, true);
      NSID_G256_STEP(3, 8, true, true);
    }
    // last four sub-tiles: nothing left to issue; in flight behind j+1 are j+2, j+3 / j+3 / nothing
    constexpr int jb = 0;                  // (only the issue branch, compiled out here, reads it)
    NSID_G256_STEP(0, 8, false, true);
    NSID_G256_STEP(1, 4, false, true);
    NSID_G256_STEP(2, 0, false, true);
    NSID_G256_STEP(3, 0, false, false);
#undef NSID_G256_STEP
  } else {
    // PAIR: the ring is two 64-deep stages = slot pairs {0,1}, {2,3}; a stage is fetched by EIGHT LDS-DMA per thread issued
    // together, the two 64-byte halves of every 128-byte operand line back to back (the second half then hits the CU's L1
    // instead of crossing the L2 -> L1 path again a sub-tile later), consumed as two 32-deep halves, and refilled from the
    // barrier in the middle of the NEXT stage's... precisely: at the barrier of half-step (t, 1) every wave holds the last
    // fragments of stage t in registers, so stage t's slots take stage t+2; stage t+1 (issued one stage earlier) must have
    // landed there (vmcnt(0): nothing younger is in flight). ONE barrier per 64-deep stage.
    auto issue2 = [&](int t, int pair) {
      const char* s2 = ubase + (long)t * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
          __builtin_amdgcn_global_load_lds((glb_vptr)(s2 + i * rb_bytes + kb * 64 + voff),
                                           (lds_vptr)(lds + (2 * pair + kb) * SLOT + dst0 + i * 1024), 16, 0, 0);
    };
    issue2(0, 0);
    issue2(1, 1);
    wait_vm<8>();
    __builtin_amdgcn_s_barrier();
    if (p.trace) t_first = __builtin_amdgcn_s_memrealtime();
    read_frags(0, 0);
#define NSID_G256_EVEN(U)                                                                 \
  do {                                                                                    \
    read_frags(((U) + 1) & 1, ((U) + 1) & 3);                                             \
    mfma_block((U) & 1);                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                                   \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                  \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#define NSID_G256_ODD(U, ISSUE, READ)                                                     \
  do {                                                                                    \
    __builtin_amdgcn_s_waitcnt(0xC07F);           /* lgkmcnt(0) */                        \
    wait_vm<0>();                                                                         \
    __builtin_amdgcn_s_barrier();                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    if (ISSUE && !(NSID_G256_ABLATE & 1)) issue2(((hb + (U)) >> 1) + 2, (U) >> 1);                                   \
    if (READ) read_frags(((U) + 1) & 1, ((U) + 1) & 3);                                   \
    mfma_block((U) & 1);                                                                  \
    if (READ) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                  \
      if (ISSUE) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                       \
    }                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
    for (int hb = 0; hb < J - 4; hb += 4) {
      NSID_G256_EVEN(0);
      NSID_G256_ODD(1, true, true);
      NSID_G256_EVEN(2);
      NSID_G256_ODD(3, true, true);
    }
    constexpr int hb = 0;
    NSID_G256_EVEN(0);
    NSID_G256_ODD(1, false, true);
    NSID_G256_EVEN(2);
    NSID_G256_ODD(3, false, false);
#undef NSID_G256_EVEN
#undef NSID_G256_ODD
  }
  // every wave passed the barrier of the last sub-tile after its last fragment read: the ring is free for the epilogue

  if (p.trace) t_loop = __builtin_amdgcn_s_memrealtime();
  // ---------------- epilogue. C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg.
  float* ldsf = reinterpret_cast<float*>(lds);
  if (p.bias != nullptr) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float bj = p.bias[j0 + wc * 64 + 16 * b + lr];
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a][b] += bj;
    }
  }
  float* red = ldsf + 8 * 32 * OLD;          // [2 sums][2 wave-rows][4 row groups][256 columns] behind the transpose buffers
  if (p.stat != nullptr) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[a][b][r];
          s += v;
          q += v * v;
        }
      const int c = wc * 64 + 16 * b + lr;
      red[((0 * 2 + wr) * 4 + rq) * 256 + c] = s;
      red[((1 * 2 + wr) * 4 + rq) * 256 + c] = q;
    }
  }
  // stores through LDS: a wave transposes 32 rows x 64 columns at a time, every lane then writes 8 consecutive bf16 (16 B):
  // one wave-store = 8 rows x 128 B
  float* ost = ldsf + wave * (32 * OLD);
  const int orow = lane >> 3, oq = (lane & 7) * 8;
  const long crow0 = (long)(i0 + wr * 128 + orow);
  const int ccol = j0 + wc * 64 + oq;
  f32x4 pre[16];
  if (p.addend != nullptr) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      pre[q] = *reinterpret_cast<const f32x4*>(p.addend + (crow0 + 8 * q) * p.ldadd + ccol);
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) ost[(16 * a2 + 4 * rq + r) * OLD + 16 * b + lr] = acc[2 * h + a2][b][r];
    // a wave reads back only its own buffer, and one wave's LDS operations execute in issue order: no barrier
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int rr = pass * 8 + orow;
      float v[8];
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(ost + rr * OLD + oq);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(ost + rr * OLD + oq + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
      if (p.relu_out) {                    // uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.addend != nullptr) {
        const bf16x8 ad = __builtin_bit_cast(bf16x8, pre[4 * h + pass]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)ad[e];
      }
      Chunk<__bf16>::store(p.C + (crow0 + 32 * h + 8 * pass) * p.ldc + ccol, v);
    }
  }
  if (p.stat != nullptr) {
    __syncthreads();                       // the parked sums of all eight waves
    if (tid < 256) {
      const long col = j0 + tid;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {     // one row of partial sums per 128-row statistics tile = per wave-row
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s += red[((0 * 2 + t2) * 4 + k) * 256 + tid];
          q += red[((1 * 2 + t2) * 4 + k) * 256 + tid];
        }
        p.stat[((long)ti * 2 + t2) * p.stat_ld + col] = s;
        p.stat[p.stat_plane + ((long)ti * 2 + t2) * p.stat_ld + col] = q;
      }
    }
  }
  if (p.trace && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.trace[4 * blockIdx.x + 0] = t_start;
    p.trace[4 * blockIdx.x + 1] = t_loop;
    p.trace[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    p.trace[4 * blockIdx.x + 3] = ((t_first - t_start) << 40) | ((unsigned long long)(xcc & 0xF) << 32) | hw;
  }
}

}  // namespace

// returns NSID_OK / NSID_ELAUNCH, or 1 when the shape is outside the kernel's preconditions (the caller then takes gemm.hip)
__attribute__((visibility("hidden")))
int nsid_gemm256_fwd_launch(const void* x, int ldx, const void* w, const float* bias, const void* addend, int ldadd, void* out,
                            int ldo, int M, int Nout, int K, bool relu_out, float* stat, long stat_plane, long stat_ld,
                            hipStream_t stream) {
  if (M % 256 != 0 || Nout % 256 != 0 || K % 128 != 0 || ldx % 8 != 0 || ldo % 8 != 0 || (addend && ldadd % 8 != 0)) return 1;
  if (!nsid_aligned16(x) || !nsid_aligned16(w) || !nsid_aligned16(out) || (addend && !nsid_aligned16(addend))) return 1;
  G256Args p{};
  p.A = static_cast<const __bf16*>(x); p.lda = ldx;
  p.B = static_cast<const __bf16*>(w); p.ldb = K;
  p.C = static_cast<__bf16*>(out); p.ldc = ldo;
  p.M = M; p.N = Nout; p.K = K;
  p.bias = bias;
  p.relu_out = relu_out ? 1 : 0;
  p.addend = static_cast<const __bf16*>(addend); p.ldadd = ldadd;
  p.stat = stat; p.stat_plane = stat_plane; p.stat_ld = stat_ld;
  p.trace = static_cast<unsigned long long*>(g_gemm_trace_host);
  const dim3 grid((M / 256) * (Nout / 256));
  static const int pair = getenv("NSID_G256_PAIR") ? atoi(getenv("NSID_G256_PAIR")) : 1;
  if (pair) NSID_LAUNCH((gemm256_fwd_kernel<true>), grid, dim3(512), 0, stream, p);
  else NSID_LAUNCH((gemm256_fwd_kernel<false>), grid, dim3(512), 0, stream, p);
  return nsid_launch_status();
}
