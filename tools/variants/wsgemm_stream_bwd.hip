// NOT PART OF THE PRODUCT LIBRARY (round 5 experiment: 8-14 % faster than the 128x128 tile kernels stand-alone, NEUTRAL in the two-stream
// step: docs/experiments.md, "Round 5", profiles/r05_wss/). This is the text that sat in csrc/wsgemm.hip between ws_bwd_kernel and
// ws_bwd_launch_t; to time it again paste it back, add `int N, K;` to WsBwdArgs, the tuning key ws_stream and the counter ws_bwd_stream
// to nsid_common.h, and at the end of nsid_ws_bwd_data_launch:
//   if ((nsid_tune(NSID_T_ws_stream) & 2) && groups == 1 && abn_r == nullptr && Nout % 256 == 0 && K % WSS_CW == 0 && (M / 128) % 8 == 0)
//     { p.N = Nout; p.K = K; return ws_bwds_launch(p, stream); }
// tests/test_wsgemm_gpu.py's BWD_STREAM_SHAPES (35 cases) were green with it.
// ---------------------------------------------------------------------------------------------------------------- backward-data, wide layers
// The same row-owner form for layers whose weight matrix does NOT fit LDS (the C = 256 / 512 stages): a workgroup owns 128 rows and a
// chunk of WSS_CW = 128 output columns, and W[n][chunk] streams through a ring of four 64-row stages by LDS-DMA (16 KB each, three in
// flight), one workgroup barrier per stage = per 16 MFMAs of a wave. dy fragments still go global -> registers, requested three stages
// ahead; every VMEM operation of the main loop is inline assembly with explicit vmcnt accounting (in issue order per stage: 4 LDS-DMA,
// then 4 fragment loads), so the compiler's own waits never drain the ring; four statically indexed fragment sets in a loop unrolled
// four times (N % 256 == 0). Accumulators: 4 tiles x 16 registers, channel on the lane, epilogue as above (ADD / BNR per tile through the
// wave's staging image) with its LDS buffers aliased onto the ring. Grid: 8 consecutive row tiles x all chunks form a run of blocks
// whose chunks of one row tile sit 8 apart = on one XCD (they re-read the same dy rows from that L2).
constexpr int WSS_CW = 128, WSS_SN = 64, WSS_SLOT = WSS_SN * WSS_CW * 2, WSS_NSLOT = 4;

template <int N>
__device__ __forceinline__ void wss_wait_vm() {
  static_assert(N >= 0 && N < 64, "6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (0xF << 8));
}
__device__ __forceinline__ void wss_glds16(const char* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
template <int IMM>
__device__ __forceinline__ void wss_gload(bf16x8& v, const char* sbase, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}

template <bool ADD, bool BNR>
__global__ __launch_bounds__(256, 2) void ws_bwds_kernel(const WsBwdArgs p) {
  constexpr int KT = WSS_CW / 32;                                     // 4 output tiles of 32 columns
  constexpr int RING = WSS_NSLOT * WSS_SLOT;
  __shared__ __attribute__((aligned(1024))) char lds[RING];
  // epilogue buffers alias the ring (free once the last stage has been read)
  char* const tb0 = lds;
  char* const side0 = lds + 4 * WS_TB;
  float* const red = reinterpret_cast<float*>(lds + 4 * WS_TB + 4 * 2 * 2048);
  static_assert(4 * WS_TB + 4 * 2 * 2048 + 4 * 2 * 2 * WSS_CW * 4 <= RING, "the epilogue fits the ring");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int nch = p.K / WSS_CW;
  const int bx = blockIdx.x, run = 8 * nch;
  const int tile = 8 * (bx / run) + (bx % run) % 8, ch = (bx % run) / 8;
  const long row0 = (long)tile * 128 + wave * 32;
  const int c0 = ch * WSS_CW;                                         // first output column of the chunk
  const int nst = p.N / WSS_SN;                                       // a multiple of 4

  // ---- LDS-DMA of a W stage [64 n][128 k]: 16 pieces of 4 rows, 4 per wave; 64-byte blocks of a row XOR (row & 3) on the source
  const int drl = lane >> 4, dpc = lane & 15;
  const int dlc = (((dpc >> 2) ^ drl) << 2) | (dpc & 3);
  const unsigned wvoff = (unsigned)((drl * p.K + dlc * 8) * 2);
  const char* const wsrc0 = reinterpret_cast<const char*>(p.w + (long)(16 * wave) * p.K + c0);
  const long wpiece = (long)p.K * 8, wstage = (long)p.K * 2 * WSS_SN;
  const unsigned lds0 = (unsigned)(size_t)(lds_vptr)lds;
  auto issue_w = [&](int st, int slot) {
    const char* sb = wsrc0 + (long)st * wstage;
    const unsigned d = lds0 + (unsigned)slot * WSS_SLOT + (unsigned)wave * 4096u;
#pragma unroll
    for (int i = 0; i < 4; ++i) wss_glds16(sb + i * wpiece, wvoff, d + i * 1024);
  };
  // ---- dy fragments of a stage: lane (row j, half h), k-step s: 16 bytes at dy[row0 + j][64 st + 16 s + 8 h ..]
  const unsigned xvoff = (unsigned)((j * (int)p.ldd + 8 * h) * 2);
  const char* const xsrc0 = reinterpret_cast<const char*>(p.dy + row0 * p.ldd);
  bf16x8 xf[4][4];                                                    // [set = stage & 3][k-step]
#define WSS_ISSUE_X(SET, ST)                                     \
  do {                                                           \
    const char* xb_ = xsrc0 + (long)(ST) * (WSS_SN * 2);         \
    wss_gload<0>(xf[SET][0], xb_, xvoff);                        \
    wss_gload<32>(xf[SET][1], xb_, xvoff);                       \
    wss_gload<64>(xf[SET][2], xb_, xvoff);                       \
    wss_gload<96>(xf[SET][3], xb_, xvoff);                       \
  } while (0)

  // ---- B fragment addressing (ws_bwd_kernel): rows 8 h + qq (+ 4) of the 16-deep k-step, 64-byte block kt ^ (row & 3) = kt ^ qq
  const int G = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  unsigned boff[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) boff[kt] = (unsigned)((8 * h + qq) * 256 + ((kt ^ qq) << 6) + 32 * (G & 1) + 8 * pp);

  f32x16 acc[KT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;

  // prologue: stages 0, 1, 2 (per stage: the wave's 4 LDS-DMA, then its 4 fragment loads); stage st lives in ring slot st & 3 and in
  // fragment set st & 3
  issue_w(0, 0);
  WSS_ISSUE_X(0, 0);
  issue_w(1, 1);
  WSS_ISSUE_X(1, 1);
  issue_w(2, 2);
  WSS_ISSUE_X(2, 2);
#define WSS_READ_B(SET, S)                                                                                  \
  _Pragma("unroll") for (int kt_ = 0; kt_ < KT; ++kt_) {                                                    \
    const bf16x4 w0_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img_ + (S) * 4096 + boff[kt_]));          \
    const bf16x4 w1_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img_ + (S) * 4096 + 1024 + boff[kt_]));   \
    fb_[SET][kt_] = __builtin_shufflevector(w0_, w1_, 0, 1, 2, 3, 4, 5, 6, 7);                              \
  }
#define WSS_STEP(U)                                                                                         \
  do {                                                                                                      \
    const int st_ = st0 + (U);                                                                              \
    /* own pieces and fragments of stage st_ have landed: behind them in the queue are at most stages st_ + 1, st_ + 2 */ \
    if (st_ + 2 < nst) wss_wait_vm<16>(); else if (st_ + 1 < nst) wss_wait_vm<8>(); else wss_wait_vm<0>();  \
    __builtin_amdgcn_s_barrier();                                                                           \
    if (st_ + 3 < nst) {                                        /* into the slot and the set read last step */ \
      issue_w(st_ + 3, ((U) + 3) & 3);                                                                      \
      WSS_ISSUE_X(((U) + 3) & 3, st_ + 3);                                                                  \
    }                                                                                                       \
    asm volatile("" : "+v"(xf[U][0]), "+v"(xf[U][1]), "+v"(xf[U][2]), "+v"(xf[U][3]));                     \
    const char* img_ = lds + (U) * WSS_SLOT;                                                                \
    /* B fragments double-buffered: the 8 transposing reads of k-step s + 1 are issued before the 4 MFMAs of k-step s, so that a   \
       wave alone on its SIMD (256 workgroups on 256 CUs) does not sit out an LDS round trip in front of every MFMA */                 \
    bf16x8 fb_[2][KT];                                                                                      \
    WSS_READ_B(0, 0);                                                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                      \
      if (s_ + 1 < 4) WSS_READ_B((s_ + 1) & 1, s_ + 1);                                                     \
      __builtin_amdgcn_sched_barrier(0);                        /* nothing crosses: reads first, then the MFMAs */ \
      _Pragma("unroll") for (int kt_ = 0; kt_ < KT; ++kt_)                                                  \
        acc[kt_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[U][s_], fb_[s_ & 1][kt_], acc[kt_], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
  } while (0)
  for (int st0 = 0; st0 < nst; st0 += 4) {
    WSS_STEP(0);
    WSS_STEP(1);
    WSS_STEP(2);
    WSS_STEP(3);
  }
#undef WSS_STEP
#undef WSS_READ_B
#undef WSS_ISSUE_X
  __builtin_amdgcn_s_barrier();                                       // every wave has read the last stage: the ring is free

  // ---- epilogue (ws_bwd_kernel's, per 32-column tile)
  const int srow = lane >> 2, scol = 8 * (lane & 3);
  const __bf16* const aptr = ADD ? p.addend + (row0 + srow) * p.ldadd + c0 + scol : nullptr;
  const __bf16* const rptr = BNR ? p.bn_r + (row0 + srow) * p.bn_ldr + c0 + scol : nullptr;
  f32x4 sa[2][2], sr[2][2];
  auto side_load = [&](int t, int par) {
    if constexpr (ADD) {
      sa[par][0] = *reinterpret_cast<const f32x4*>(aptr + t * 32);
      sa[par][1] = *reinterpret_cast<const f32x4*>(aptr + 16 * p.ldadd + t * 32);
    }
    if constexpr (BNR) {
      sr[par][0] = *reinterpret_cast<const f32x4*>(rptr + t * 32);
      sr[par][1] = *reinterpret_cast<const f32x4*>(rptr + 16 * p.bn_ldr + t * 32);
    }
  };
  if constexpr (ADD || BNR) side_load(0, 0);
  char* const tb = tb0 + wave * WS_TB;
  char* const tw = tb + j * WS_TP + 8 * h;
  const char* const tr = tb + (8 * G + qq) * WS_TP + (4 * pp) * 2;
  __bf16* const orow = p.dx + (row0 + (lane & 15)) * p.ldi + c0 + 8 * G;
  char* const simg = side0 + wave * 4096;
  char* const sw_ = simg + lane * 16;
  const char* const sread = simg + (4 * h + qq) * 64 + 32 * (G & 1) + 8 * pp;
  float s0[KT], s1[KT];
  const bool bn_unit = p.bn_slope == 1.f;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if constexpr (ADD || BNR) {
      if (t + 1 < KT) side_load(t + 1, (t + 1) & 1);
    }
    if constexpr (ADD) {
      *reinterpret_cast<f32x4*>(sw_) = sa[t & 1][0];
      *reinterpret_cast<f32x4*>(sw_ + 1024) = sa[t & 1][1];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bf16x4 a4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(sread + 512 * g));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][4 * g + e] += (float)a4[e];
      }
    }
    bf16x4 o4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      o4[g] = __builtin_convertvector((f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]}), bf16x4);
      *reinterpret_cast<bf16x4*>(tw + 16 * g) = o4[g];
    }
    if constexpr (BNR) {
      *reinterpret_cast<f32x4*>(sw_ + 2048) = sr[t & 1][0];
      *reinterpret_cast<f32x4*>(sw_ + 2048 + 1024) = sr[t & 1][1];
      const long chn = (long)c0 + t * 32 + j;
      const float bsc = p.bn_scale[chn], bsh = p.bn_shift[chn], bmu = p.bn_mean[chn], bis = p.bn_invstd[chn];
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const bf16x4 r4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(sread + 2048 + 512 * g));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dyv = (float)o4[g][e], xx = (float)r4[e];
          float gg = dyv;
          if (!bn_unit) gg = (bsc * xx + bsh) > 0.f ? dyv : dyv * p.bn_slope;
          a0 += gg;
          a1 += gg * ((xx - bmu) * bis);
        }
      }
      s0[t] = a0; s1[t] = a1;
    }
    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tr));
    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tr + 4 * WS_TP));
    const bf16x4 v2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tr + 32));
    const bf16x4 v3 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tr + 4 * WS_TP + 32));
    *reinterpret_cast<bf16x8*>(orow + t * 32) = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(orow + 16 * p.ldi + t * 32) = __builtin_shufflevector(v2, v3, 0, 1, 2, 3, 4, 5, 6, 7);
  }
  if constexpr (BNR) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      red[((wave * 2 + h) * 2 + 0) * WSS_CW + t * 32 + j] = s0[t];
      red[((wave * 2 + h) * 2 + 1) * WSS_CW + t * 32 + j] = s1[t];
    }
    __syncthreads();
    for (int c = tid; c < WSS_CW; c += 256) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { a0 += red[(k * 2 + 0) * WSS_CW + c]; a1 += red[(k * 2 + 1) * WSS_CW + c]; }
      const long col = (long)c0 + c;
      p.bn_partial[(long)tile * p.bn_ld + col] = a0;
      p.bn_partial[p.bn_plane + (long)tile * p.bn_ld + col] = a1;
    }
  }
}

int ws_bwds_launch(const WsBwdArgs& p, hipStream_t s) {
  const dim3 grid((p.M / 128) * (p.K / WSS_CW)), block(256);
  const bool add = p.addend != nullptr, bnr = p.bn_r != nullptr;
  if (add) {
    if (bnr) NSID_LAUNCH((ws_bwds_kernel<true, true>), grid, block, 0, s, p);
    else NSID_LAUNCH((ws_bwds_kernel<true, false>), grid, block, 0, s, p);
  } else {
    if (bnr) NSID_LAUNCH((ws_bwds_kernel<false, true>), grid, block, 0, s, p);
    else NSID_LAUNCH((ws_bwds_kernel<false, false>), grid, block, 0, s, p);
  }
  nsid_count(NSID_C_ws_bwd_stream);
  return nsid_launch_status();
}

