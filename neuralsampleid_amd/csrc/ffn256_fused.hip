// Eval-mode FFN in ONE launch with the x tile in registers (forward-only fingerprint extraction, BASELINE config 5):
//     out = x + W2 relu(W1 x + b1) + b2      (encoder/graph_encoder.py:82-89 with both BatchNorms folded into the convs), H = 4C.
// Written for the C = 256 stage (described below with its numbers), since templated on C and serving C = 128 / 64 too (see the kernel).
// Six of the twelve blocks sit at C = 256, and their un-fused FFN pair is a quarter of a micro-batch: gemm256 writes the hidden
// tensor (M x 1024 bf16: 268 MB at a 2 048-clip micro-batch) in 116 us and reads it back in 87 us — both launches are bound by those
// bytes, not by the matrix pipe (0.7 PFLOP/s). ffn_fused.hip keeps the hidden chunk on the CU for C = 64 / 128 with the x tile in
// LDS; at C = 256 that form needs 161 KB. Here the x tile lives in REGISTERS instead:
//   * 512 threads = 8 waves, a workgroup owns 256 rows, a wave 32 of them (two 16-row MFMA tiles). Its x rows are its B-operand
//     fragments for the whole kernel: 8 k-steps x 2 row tiles x 4 registers = 64 VGPRs, loaded once straight from global memory.
//   * The hidden dimension is walked in 32 chunks of 32. Per chunk the workgroup needs W1[chunk] (32 x 256) and W2[:, chunk]
//     (256 x 32): 16 KB each, fetched by LDS-DMA (global_load_lds_dwordx4, four per wave and chunk) into a ring of four 32 KB slots,
//     two chunks ahead; ONE workgroup barrier per chunk hands a slot over. GEMM 2 of chunk ch - 1 runs interleaved with GEMM 1 of
//     chunk ch (software pipeline: the hidden chunk's bias / ReLU / pack / LDS round trip is off the matrix pipe's critical path).
//     Every wave reads the whole chunk (each weight fragment feeds two MFMAs: both row tiles), so the weights cross L2 -> LDS once per
//     256 rows: 32 KB per 128 MFMAs of a wave pair.
//   * GEMM 1 (K = 256): D[hidden][row] with the operands swapped (weights first), so a lane ends up with four consecutive hidden values
//     of one row: bias + ReLU + bf16 pack, one 8-byte LDS store into the wave's PRIVATE 2 KB hidden buffer (no barrier: a wave's LDS
//     operations execute in order), read back as the B operand of GEMM 2 (K = 32), which accumulates D[channel][row] in 128 registers.
//   * LDS images: [16 rows][32 k] sub-blocks of 1 KB, the 16-byte chunk index XOR-ed with 2 * (row >> 3 & 1) -- gemm256.hip's format:
//     one LDS-DMA writes a sub-block, every ds_read_b128 of a fragment is conflict-free.
//   * Epilogue: (acc + b2) + x in fp32 with ONE rounding (the two-launch form's arithmetic) from registers: the LDS-DMA of W2 permutes
//     the output channels so that a lane's accumulators are the channels of its own x fragments. x is read once, the output leaves in
//     16-byte stores, nothing is staged (measured, profiles/r04b: the un-overlapped x read / staged epilogue / x re-read of the first
//     version were 32 of its 155 us).
// Per chunk a wave issues 32 + 32 MFMAs (2 048 matrix cycles per SIMD with two waves on it) against 34 fragment reads (256 KB of LDS
// reads per workgroup and chunk: half the LDS array's rate); the M x 1024 hidden tensor never exists.
#include "nsid_common.h"
#include <type_traits>

namespace {

struct F256Args {
  const __bf16* x; const __bf16* w1; const float* b1; const __bf16* w2; const float* b2; __bf16* out;
  int M;
};

typedef __attribute__((address_space(3))) void* lds_vptr;

constexpr int F_HC = 32;                      // hidden units per chunk
constexpr int F_NS = 4;                       // ring slots: chunk ch (GEMM 1), chunk ch - 1 (GEMM 2), chunks ch + 1 and ch + 2 in flight
// per channel count C (H = 4 C): a chunk's W1 image is 32 hidden rows x C = C / 16 sub-blocks of [16 rows][32 k] (1 KB each), its W2
// image C rows x 32 hidden = C / 16 sub-blocks: a ring slot is C / 8 KB (32 KB at C = 256)
constexpr int F_PF = 1;            // fragment prefetch distance in steps (one step = 2 LDS reads, 4 MFMAs)

template <int N>
__device__ __forceinline__ void f_wait_vm() {
  static_assert(N >= 0 && N < 64, "6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (0xF << 8));
}
// one LDS-DMA: 64 lanes x 16 B from (uniform base) + (lane offset) to LDS bytes [dst, dst + 1 KB)   (gemm256.hip glds16)
__device__ __forceinline__ void f_glds16(const char* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

// acc += A B with the accumulator in AGPRs. With one wave per SIMD a wave owns 256 VGPRs + 256 AGPRs; left to itself the register
// allocator spreads x fragments and accumulators over both files and pays for it in v_accvgpr moves inside the chunk loop (measured:
// 187 us against 147 us for the 8-wave form). The constraint settles it: all 64 output accumulator tiles live in a[0:255] for the whole
// tile, the VGPRs hold x fragments, GEMM 1 accumulators and operands.
__device__ __forceinline__ void f_mfma_acc(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  // (the MFMA itself stays an intrinsic so that the compiler's hazard recognizer sees it: an MFMA inside inline asm gets no wait
  // states before a v_accvgpr_read of its result, and produced wrong sums; the empty asm only fixes the register file)
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  asm volatile("" : "+a"(acc));
}

// NW waves per workgroup, each owning RT = 16 / NW row tiles of 16 rows (256 rows per workgroup either way):
//   NW = 8: 32 rows per wave, two waves per SIMD at <= 256 registers;
//   NW = 4: 64 rows per wave, ONE wave per SIMD with the whole 512-register budget (128 of x fragments, 256 of output accumulators):
//           every weight fragment read from LDS feeds four MFMAs instead of two, which halves the LDS reads per matrix cycle.
//
// (Round 4 also widened this launch BACKWARDS -- the Grapher's fc2 + shortcut as 16 more ring iterations, then the max-relative graph conv
// on top: bit-identical, measured slower (docs/experiments.md); those forms live in tools/variants/ffn256_fused_block_forms.hip.)
//
// C = 128 / 64 (round 4): the same kernel with RT = 2 / 4 row tiles per wave (256- / 512-row workgroup tiles; C = 128 with RT = 4 needs
// 55 registers more than a wave has at two per SIMD); a chunk is 16 / 8 LDS-DMA pieces. They replace ffn_fused.hip's
// x-tile-in-LDS form for these widths (tuning key ffn_regs).
template <int F_C, int NW, int RT>
__global__ __attribute__((amdgpu_flat_work_group_size(64 * NW, 64 * NW), amdgpu_waves_per_eu(NW / 4, NW / 4)))
void ffn256_fused_kernel(const F256Args p) {
  constexpr int F_H = 4 * F_C, F_NCH = F_H / F_HC;
  constexpr int KS1 = F_C / 32;               // k-steps of GEMM 1 = x fragments per row tile
  constexpr int EP = 2 * KS1 * RT; // global stores (+ next-x loads) a wave issues in a tile's epilogue (when another tile follows)
  constexpr int CT = F_C / 16;                // output-channel tiles of GEMM 2 (= GEMM 1 fragments per chunk: 2 hidden tiles x KS1)
  constexpr int W2OFF = CT * 1024;            // a slot: [W1 chunk image | W2 chunk image]
  constexpr int F_SLOT = 2 * CT * 1024;
  constexpr int TR = NW * RT * 16;            // rows per workgroup tile
  constexpr int F_HB = NW * RT * 1024;        // every wave its own hidden sub-blocks (one per row tile)
  constexpr int F_LDS = F_NS * F_SLOT + F_HB + F_H * 4 + F_C * 4;
  constexpr int PW = 2 * CT / NW;             // LDS-DMA pieces per wave and chunk
  static_assert(2 * CT % NW == 0 && F_NCH % F_NS == 0, "whole pieces per wave; a tile's chunks start on slot 0");
  __shared__ __attribute__((aligned(1024))) char lds[F_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, rq = lane >> 4;
  const int ntiles = p.M / TR;
  char* const hb = lds + F_NS * F_SLOT + wave * (RT * 1024);
  float* const b1s = reinterpret_cast<float*>(lds + F_NS * F_SLOT + F_HB);
  float* const b2s = b1s + F_H;
  const unsigned lds0 = (unsigned)(size_t)(lds_vptr)lds;

  // ---- x fragments: lane (lr, rq) of row tile b holds x[row0 + 16 b + lr][32 ks + 8 rq .. + 7]
  // (the first tile's here; every later tile's are fetched by the epilogue of the tile before, register by register as it lets go of them)
  bf16x8 xf[KS1][RT];
  // a lane's byte offset inside a tile's x / out rows (uniform 64-bit tile base + 32-bit lane offset + immediate)
  const unsigned xo = (unsigned)((wave * (16 * RT) + lr) * F_C + 8 * rq) * 2u;
  {
    const char* xt = reinterpret_cast<const char*>(p.x + (long)blockIdx.x * TR * F_C);
#pragma unroll
    for (int b = 0; b < RT; ++b)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) xf[ks][b] = *reinterpret_cast<const bf16x8*>(xt + xo + b * (16 * F_C * 2) + ks * 64);
  }
  for (int i = tid; i < F_H / 4; i += 64 * NW) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(p.b1)[i];
  if (tid < F_C / 4) reinterpret_cast<f32x4*>(b2s)[tid] = reinterpret_cast<const f32x4*>(p.b2)[tid];
  // the staged bias vectors are read by OTHER waves later: fence + barrier here, once per launch and before any LDS-DMA is in flight
  // (the bare s_barrier of the chunk loop waits for no LDS store, and gfx950 inserts no lgkmcnt(0) in front of it: ADVICE r4)
  __syncthreads();

  // ---- LDS-DMA addressing (gemm256.hip): lane l supplies row l >> 2 of a 16-row sub-block and the logical 16-byte chunk that lives at
  // physical chunk l & 3
  // W1: row stride F_C elements. W2: row stride F_H elements; sub-block c = 2 ks + half takes its row 4 q + e from output channel
  // 32 ks + 8 q + 4 half + e, so that a lane's accumulators are the channels of its own x fragments (epilogue).
  // The two lane offsets are RE-DERIVED from the lane index at every use (three vector instructions): as kernel-lifetime values the
  // register allocator spilled them (and seven more) around the tile loop of the 256-register instantiation -- 40 bytes of scratch per
  // lane, 7 stores + 7 loads per tile (round 4); the empty asm keeps the compiler from hoisting them back out.
  auto issue = [&](int u, int tile) {                                   // this wave's PW pieces of chunk u (of row tile `tile`)
    int l_ = lane;
    asm volatile("" : "+v"(l_));
    const int grow = l_ >> 2, lc = (l_ & 3) ^ (((grow >> 3) & 1) << 1);
    const unsigned voff1 = (unsigned)(grow * F_C + lc * 8) * 2u;
    const unsigned voff2 = (unsigned)((8 * (grow >> 2) + (grow & 3)) * F_H + lc * 8) * 2u;
    const unsigned dst = lds0 + (u % F_NS) * F_SLOT;
    const int h0 = u * F_HC;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int q = PW * wave + i;            // 0 .. CT - 1: W1 sub-block (hidden tile a, k-step ks); CT .. 2 CT - 1: W2 sub-block (channel tile c)
      if (q < CT) {
        const int a = q / KS1, ks = q % KS1;
        f_glds16(reinterpret_cast<const char*>(p.w1 + (long)(h0 + 16 * a) * F_C + 32 * ks), voff1, dst + q * 1024);
      } else {
        const int c = q - CT;
        f_glds16(reinterpret_cast<const char*>(p.w2 + (long)(32 * (c >> 1) + 4 * (c & 1)) * F_H + h0), voff2, dst + W2OFF + c * 1024);
      }
    }
  };
  issue(0, blockIdx.x);
  issue(1, blockIdx.x);
  const int lo = (lr * 64 + rq * 16) ^ (((lr >> 3) & 1) << 5);          // fragment read offset inside a weight sub-block
  // the wave-private hidden sub-blocks use their own swizzle, chunk ^ (row >> 1 & 3): the 8-byte stores of a 16-lane group (one column
  // of 16 rows) then fall on 8 different 16-byte slots (2-way; 4-way with the weight images' swizzle, measured as 23 % of the
  // kernel's LDS cycles), and the 16-byte fragment reads stay conflict-free
  // ---- persistent over the row tiles: workgroup g takes tiles g, g + grid, ... (one workgroup per CU: 148 KB of LDS)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    char* const ot = reinterpret_cast<char*>(p.out + (long)tile * TR * F_C);
    const char* const xn = reinterpret_cast<const char*>(p.x + ((long)tile + gridDim.x) * TR * F_C);    // the next tile's x (if any)
    const bool more = tile + (int)gridDim.x < ntiles;                   // uniform
    const bool first = tile == (int)blockIdx.x;
    f32x4 acc2[CT][RT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int b = 0; b < RT; ++b) acc2[c][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 hf[RT];                            // the hidden chunk of the PREVIOUS iteration as GEMM 2's B fragments
#pragma unroll
    for (int b = 0; b < RT; ++b) hf[b] = bf16x8{};


    // Software pipeline over the chunks: iteration ch runs GEMM 1 of chunk ch (W1 image of slot ch % 4) interleaved with GEMM 2 of
    // chunk ch - 1 (W2 image of slot (ch - 1) % 4, hidden fragments in registers): the two are independent, so the matrix pipe never
    // waits for the bias / ReLU / pack / LDS round trip of the hidden chunk, and 16 steps of {2 fragment reads, 4 RT/2 MFMAs} hide the
    // LDS latency. Iteration F_NCH only drains GEMM 2 of the last chunk -- and starts the NEXT tile's first two chunks: after its
    // barrier every wave is past iteration F_NCH - 1, so slots 0 and 1 (chunks F_NCH - 4, F_NCH - 3) are free.
    constexpr int NCH = F_NCH;
    auto iter = [&](auto G1, auto G2, const int ch) {
      constexpr bool g1 = decltype(G1)::value, g2 = decltype(G2)::value;     // GEMM 1 of chunk ch / GEMM 2 of chunk ch - 1 in this iteration
      // chunk ch has landed for THIS wave when at most the PW pieces of chunk ch + 1 are still in flight; the barrier then says so
      // for every wave, and that every wave is done with the slot chunk ch + 2 is about to overwrite (chunk ch - 2: its W2 image was
      // read in iteration ch - 1)
      // (vmcnt counts loads, stores and LDS-DMA together, in issue order. On a later tile of a persistent workgroup the epilogue of the
      // tile before issued EP operations -- the next x fragments and the output stores -- AFTER chunks 0 and 1: they are younger than
      // what iterations 0 and 1 wait for and stay in flight; waiting them out here would put every tile's store tail on the critical
      // path)
      if (ch < 2 && !first) f_wait_vm<(PW + EP < 63 ? PW + EP : 63)>();
      else if (ch + 1 < F_NCH) f_wait_vm<PW>();
      else f_wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (ch + 2 < F_NCH) issue(ch + 2, tile);
      else if (!g1 && more) { issue(0, tile + gridDim.x); issue(1, tile + gridDim.x); }
      const char* s1 = lds + (ch % F_NS) * F_SLOT;                       // W1 image of chunk ch
      const char* s2 = lds + ((ch + F_NS - 1) % F_NS) * F_SLOT + W2OFF;    // W2 image of chunk ch - 1
      f32x4 acc1[2][RT];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fragment reads run F_PF steps ahead of their MFMAs
      auto rd1 = [&](int st) { return *reinterpret_cast<const bf16x8*>(s1 + (((st & 1) * KS1) + (st >> 1)) * 1024 + lo); };
      auto rd2 = [&](int st) { return *reinterpret_cast<const bf16x8*>(s2 + st * 1024 + lo); };
      bf16x8 q1[F_PF], q2[F_PF];
#pragma unroll
      for (int i = 0; i < F_PF; ++i) {
        q1[i] = bf16x8{};
        q2[i] = bf16x8{};
        if constexpr (g1) q1[i] = rd1(i);
        if constexpr (g2) q2[i] = rd2(i);
      }
#pragma unroll
      for (int st = 0; st < CT; ++st) {       // GEMM 1 step (ks = st >> 1, a = st & 1) beside GEMM 2 step (channel tile st)
        const bf16x8 f1 = q1[st % F_PF], f2 = q2[st % F_PF];
        if (st + F_PF < CT) {
          if constexpr (g1) q1[st % F_PF] = rd1(st + F_PF);
          if constexpr (g2) q2[st % F_PF] = rd2(st + F_PF);
        }
        if constexpr (g1) {
#pragma unroll
          for (int b = 0; b < RT; ++b)
            acc1[st & 1][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1, xf[st >> 1][b], acc1[st & 1][b], 0, 0, 0);
        }
        if constexpr (g2) {
#pragma unroll
          for (int b = 0; b < RT; ++b) {
            if constexpr (NW == 4)            // the output accumulators are PINNED to the 256 AGPRs (see f_mfma_acc)
              f_mfma_acc(acc2[st][b], f2, hf[b]);
            else
              acc2[st][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2, hf[b], acc2[st][b], 0, 0, 0);
          }
        }
      }
      if constexpr (g1) {
        // relu(acc + b1) -> bf16: four consecutive hidden values of one row per lane, into the wave's own sub-block of row tile b;
        // read back at once as the next iteration's B fragments (a wave's LDS operations execute in order: no barrier)
        // (lane-derived LDS offsets re-derived here: see issue())
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int lr = l2 & 15, rq = l2 >> 4, hsw = (lr >> 1) & 3;
        const int loh = lr * 64 + ((rq ^ hsw) << 4);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const f32x4 bj = *reinterpret_cast<const f32x4*>(b1s + ch * F_HC + 16 * a + 4 * rq);
          const int off = lr * 64 + (((2 * a + (rq >> 1)) ^ hsw) << 4) + (rq & 1) * 8;
#pragma unroll
          for (int b = 0; b < RT; ++b) {
            const f32x4 v = acc1[a][b] + bj;
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (__bf16)fmaxf(v[e], 0.f);
            *reinterpret_cast<bf16x4*>(hb + b * 1024 + off) = h;
          }
        }
#pragma unroll
        for (int b = 0; b < RT; ++b) hf[b] = *reinterpret_cast<const bf16x8*>(hb + b * 1024 + loh);
      }
    };
    iter(std::true_type{}, std::false_type{}, 0);
    for (int ch = 1; ch < NCH; ++ch) iter(std::true_type{}, std::true_type{}, ch);
    iter(std::false_type{}, std::true_type{}, NCH);
    // ---- epilogue: out = (acc2 + b2) + x in fp32, ONE rounding (as the two-launch form: gemm256.hip's residual epilogue), straight
    // from registers. GEMM 2's output-channel order was chosen for this (see issue()): accumulator tile c = 2 ks + half, element e of
    // lane (lr, rq) is channel 32 ks + 8 rq + 4 half + e of row 16 b + lr -- the very channels whose x the lane holds in xf[ks][b] --
    // so the residual needs no second read of x, no LDS staging and no barrier, and a lane stores 8 consecutive channels (16 bytes).
    int l3 = lane;
    asm volatile("" : "+v"(l3));
    const int rq3 = l3 >> 4;
    const unsigned xo3 = (unsigned)((wave * (16 * RT) + (l3 & 15)) * F_C + 8 * rq3) * 2u;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const f32x4 bc0 = *reinterpret_cast<const f32x4*>(b2s + 32 * ks + 8 * rq3);
      const f32x4 bc1 = *reinterpret_cast<const f32x4*>(b2s + 32 * ks + 8 * rq3 + 4);
#pragma unroll
      for (int b = 0; b < RT; ++b) {
        const f32x4 y0 = acc2[2 * ks][b] + bc0, y1 = acc2[2 * ks + 1][b] + bc1;
        const bf16x8 xr = xf[ks][b];
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (__bf16)(y0[e] + (float)xr[e]);
          o[4 + e] = (__bf16)(y1[e] + (float)xr[4 + e]);
        }
        *reinterpret_cast<bf16x8*>(ot + xo3 + b * (16 * F_C * 2) + ks * 64) = o;
        if (more) xf[ks][b] = *reinterpret_cast<const bf16x8*>(xn + xo3 + b * (16 * F_C * 2) + ks * 64);
      }
    }
  }  // tile
}

}  // namespace

// returns NSID_OK / NSID_ELAUNCH, or 1 (nothing launched) outside {C = 256 or 128, M % 256 == 0} / {C = 64, M % 512 == 0} with H = 4 C.
__attribute__((visibility("hidden")))
int nsid_ffn256_fused_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out, int M, int C,
                             int H, hipStream_t stream) {
  if (!(C == 256 || C == 128 || C == 64) || H != 4 * C || M <= 0) return 1;
  const int tr = C == 64 ? 512 : 256;         // rows per workgroup tile
  if (M % tr != 0) return 1;
  F256Args p{static_cast<const __bf16*>(x), static_cast<const __bf16*>(w1), b1, static_cast<const __bf16*>(w2), b2,
             static_cast<__bf16*>(out), M};
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, n = 0;               // an attribute query: legal whatever the stream is doing (capture)
    n_cu = (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) ? n : 256;
    if (n_cu <= 0) n_cu = 256;
  }
  const int ntiles = M / tr;
  // tuning key ffn256: 1 = persistent, one workgroup of 8 waves per CU; 2 = 8 waves, one workgroup per tile; 4 = persistent, 4 waves of
  // 64 rows with the output accumulators in AGPRs (C = 256 only; measured 154 us against 146 us: docs/experiments.md)
  const int wgs = (nsid_tune(NSID_T_ffn256) == 2 || ntiles < n_cu) ? ntiles : n_cu;
  const bool w4 = nsid_tune(NSID_T_ffn256) == 4;
  if (C == 128) {
    NSID_LAUNCH((ffn256_fused_kernel<128, 8, 2>), dim3(wgs), dim3(512), 0, stream, p);
  } else if (C == 64) {
    NSID_LAUNCH((ffn256_fused_kernel<64, 8, 4>), dim3(wgs), dim3(512), 0, stream, p);
  } else {
    if (w4) NSID_LAUNCH((ffn256_fused_kernel<256, 4, 4>), dim3(wgs), dim3(256), 0, stream, p);
    else NSID_LAUNCH((ffn256_fused_kernel<256, 8, 2>), dim3(wgs), dim3(512), 0, stream, p);
  }
  return nsid_launch_status();
}
