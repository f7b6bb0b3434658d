// Max-relative neighbour aggregation (MRConv2d without its grouped conv) — HBM-bound gather kernels.
// forward : one thread per (row, channel quad); the k neighbour rows of a clip are re-read from L2 (a clip is 64 KB),
//           so HBM sees x once in, u once out (2*N*C*4 + N*k*4 bytes per clip).
// backward: one workgroup per clip accumulates dy in LDS (ds_add_f32), so the scatter never touches HBM atomics.
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

template <typename T>
__global__ __launch_bounds__(256) void mr_bwd_kernel(const T* __restrict__ du, const int32_t* __restrict__ idx,
                                                     const uint8_t* __restrict__ argmax, int N, int C, int k,
                                                     T* __restrict__ dy) {
  extern __shared__ __attribute__((aligned(16))) float acc[];   // [N][C] fp32 accumulator for the whole clip
  constexpr int NV = Chunk<T>::N;          // du elements per chunk = NV/2 channel pairs (even, odd)
  const int b = blockIdx.x;
  const long row0 = (long)b * N;
  const int CH = C / (NV / 2);             // chunks of du per node row
  const int total = N * CH;
  // pass 1: dy = du_even - du_odd (pass-through of the interleave, and the -1 on the centre of the max-relative)
  for (int q = threadIdx.x; q < total; q += blockDim.x) {
    const int n = q / CH, c = (q % CH) * (NV / 2);
    float g[NV];
    Chunk<T>::load(du + (row0 + n) * (2L * C) + 2 * c, g);
#pragma unroll
    for (int e = 0; e < NV / 2; ++e) acc[n * C + c + e] = g[2 * e] - g[2 * e + 1];
  }
  __syncthreads();
  // pass 2: +du_odd to the arg-max neighbour of every (node, channel)
  for (int q = threadIdx.x; q < total; q += blockDim.x) {
    const int n = q / CH, c = (q % CH) * (NV / 2);
    float g[NV];
    Chunk<T>::load(du + (row0 + n) * (2L * C) + 2 * c, g);
    const uint8_t* am = argmax + (row0 + n) * C + c;
    const int32_t* nb = idx + (row0 + n) * k;
#pragma unroll
    for (int e = 0; e < NV / 2; ++e) {
      const int tgt = min(max(nb[min((int)am[e], k - 1)], 0), N - 1);
      atomicAdd(&acc[tgt * C + c + e], g[2 * e + 1]);
    }
  }
  __syncthreads();
  const int CV = C / NV;
  for (int q = threadIdx.x; q < N * CV; q += blockDim.x) {
    float v[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = acc[q * NV + e];
    Chunk<T>::store(dy + row0 * C + (long)q * NV, v);
  }
}

}  // namespace

extern "C" int nsid_mr_aggregate_fwd(const void* r, int ldr, const float* scale, const float* shift,
                                     const int32_t* idx, int B, int N, int C, int k, void* u, uint8_t* argmax,
                                     int dtype, void* stream) {
  NSID_REQUIRE(r && idx && u && B > 0 && N > 0 && C > 0 && k > 0 && k <= 255 && NSID_DTYPE_OK(dtype));
  const int nv = dtype == NSID_BF16 ? 8 : 4;
  NSID_REQUIRE(C % nv == 0 && ldr % nv == 0 && ldr >= C && nsid_aligned16(r) && nsid_aligned16(u));
  NSID_REQUIRE((scale == nullptr) == (shift == nullptr));
  const long total = (long)B * N * (C / nv);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  NSID_DISPATCH_DTYPE(dtype, T, {
    NSID_LAUNCH((mr_fwd_kernel<T>), dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                static_cast<const T*>(r), (long)ldr, scale, shift, idx, (long)B * N, N, C, k, static_cast<T*>(u), argmax);
  });
  return nsid_launch_status();
}

extern "C" int nsid_mr_aggregate_bwd(const void* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C,
                                     int k, void* dy, int dtype, void* stream) {
  NSID_REQUIRE(du && idx && argmax && dy && B > 0 && N > 0 && C > 0 && k > 0 && k <= 255 && NSID_DTYPE_OK(dtype));
  NSID_REQUIRE(C % (dtype == NSID_BF16 ? 8 : 4) == 0 && nsid_aligned16(du) && nsid_aligned16(dy));
  const size_t bytes = (size_t)N * C * sizeof(float);
  NSID_REQUIRE(bytes <= 160 * 1024);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_kernel<float>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_kernel<__bf16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  NSID_DISPATCH_DTYPE(dtype, T, {
    NSID_LAUNCH((mr_bwd_kernel<T>), dim3(B), dim3(256), bytes, static_cast<hipStream_t>(stream),
                static_cast<const T*>(du), idx, argmax, N, C, k, static_cast<T*>(dy));
  });
  return nsid_launch_status();
}
