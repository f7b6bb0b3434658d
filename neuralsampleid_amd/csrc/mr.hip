// Max-relative neighbour aggregation (MRConv2d without its grouped conv) — HBM-bound gather kernels.
// forward : one thread per (row, channel quad); the k neighbour rows of a clip are re-read from L2 (a clip is 64 KB),
//           so HBM sees x once in, u once out (2*N*C*4 + N*k*4 bytes per clip).
// backward: one workgroup per clip accumulates dy in LDS (ds_add_f32), so the scatter never touches HBM atomics.
#include "nsid_common.h"

namespace {

__global__ __launch_bounds__(256) void mr_fwd_kernel(const float* __restrict__ r, long ldr,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const int32_t* __restrict__ idx, long rows_total, int N, int C,
                                                     int k, float* __restrict__ u, uint8_t* __restrict__ argmax) {
  const int C4 = C >> 2;
  const long total = rows_total * C4;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long row = q / C4;
    const int c = (int)(q % C4) * 4;
    const long clip0 = (row / N) * N;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale != nullptr) {
      sc = *reinterpret_cast<const f32x4*>(scale + c);
      sh = *reinterpret_cast<const f32x4*>(shift + c);
    }
    f32x4 y = *reinterpret_cast<const f32x4*>(r + row * ldr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = sc[e] * y[e] + sh[e];
    f32x4 best = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    int arg[4] = {0, 0, 0, 0};
    const int32_t* nb = idx + row * k;
    for (int j = 0; j < k; ++j) {
      const long nrow = clip0 + min(max(nb[j], 0), N - 1);     // ids come from the caller: never fault on them
      f32x4 v = *reinterpret_cast<const f32x4*>(r + nrow * ldr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = (sc[e] * v[e] + sh[e]) - y[e];
        if (d > best[e]) { best[e] = d; arg[e] = j; }       // strict: first maximum wins, as torch.max
      }
    }
    float* dst = u + row * (2L * C) + 2 * c;                  // interleave: channel 2c = y, 2c+1 = max-relative
    *reinterpret_cast<f32x4*>(dst) = f32x4{y[0], best[0], y[1], best[1]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{y[2], best[2], y[3], best[3]};
    if (argmax != nullptr)
      *reinterpret_cast<uchar4*>(argmax + row * C + c) =
          make_uchar4((unsigned char)arg[0], (unsigned char)arg[1], (unsigned char)arg[2], (unsigned char)arg[3]);
  }
}

__global__ __launch_bounds__(256) void mr_bwd_kernel(const float* __restrict__ du, const int32_t* __restrict__ idx,
                                                     const uint8_t* __restrict__ argmax, int N, int C, int k,
                                                     float* __restrict__ dy) {
  extern __shared__ __attribute__((aligned(16))) float acc[];   // [N][C]
  const int b = blockIdx.x;
  const long row0 = (long)b * N;
  const int C2 = C >> 1;                 // pairs of channels: one float4 of du = (e0, o0, e1, o1)
  const int total = N * C2;
  // pass 1: dy = du_even - du_odd (pass-through of the interleave, and the -1 on the centre of the max-relative)
  for (int q = threadIdx.x; q < total; q += blockDim.x) {
    const int n = q / C2, c = (q % C2) * 2;
    const f32x4 g = *reinterpret_cast<const f32x4*>(du + (row0 + n) * (2L * C) + 2 * c);
    acc[n * C + c] = g[0] - g[1];
    acc[n * C + c + 1] = g[2] - g[3];
  }
  __syncthreads();
  // pass 2: +du_odd to the arg-max neighbour of every (node, channel)
  for (int q = threadIdx.x; q < total; q += blockDim.x) {
    const int n = q / C2, c = (q % C2) * 2;
    const f32x4 g = *reinterpret_cast<const f32x4*>(du + (row0 + n) * (2L * C) + 2 * c);
    const uint8_t* am = argmax + (row0 + n) * C + c;
    const int32_t* nb = idx + (row0 + n) * k;
    const int t0 = min(max(nb[min((int)am[0], k - 1)], 0), N - 1);
    const int t1 = min(max(nb[min((int)am[1], k - 1)], 0), N - 1);
    atomicAdd(&acc[t0 * C + c], g[1]);
    atomicAdd(&acc[t1 * C + c + 1], g[3]);
  }
  __syncthreads();
  const int C4 = C >> 2;
  for (int q = threadIdx.x; q < N * C4; q += blockDim.x)
    reinterpret_cast<f32x4*>(dy + row0 * C)[q] = reinterpret_cast<const f32x4*>(acc)[q];
}

}  // namespace

extern "C" int nsid_mr_aggregate_fwd(const float* r, int ldr, const float* scale, const float* shift,
                                     const int32_t* idx, int B, int N, int C, int k, float* u, uint8_t* argmax,
                                     void* stream) {
  NSID_REQUIRE(r && idx && u && B > 0 && N > 0 && C > 0 && k > 0 && k <= 255);
  NSID_REQUIRE(C % 4 == 0 && ldr % 4 == 0 && ldr >= C && nsid_aligned16(r) && nsid_aligned16(u));
  NSID_REQUIRE((scale == nullptr) == (shift == nullptr));
  const long total = (long)B * N * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  NSID_LAUNCH(mr_fwd_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), r, (long)ldr,
                     scale, shift, idx, (long)B * N, N, C, k, u, argmax);
  return nsid_launch_status();
}

extern "C" int nsid_mr_aggregate_bwd(const float* du, const int32_t* idx, const uint8_t* argmax, int B, int N, int C,
                                     int k, float* dy, void* stream) {
  NSID_REQUIRE(du && idx && argmax && dy && B > 0 && N > 0 && C > 0 && k > 0 && k <= 255);
  NSID_REQUIRE(C % 4 == 0 && nsid_aligned16(du) && nsid_aligned16(dy));
  const size_t bytes = (size_t)N * C * sizeof(float);
  NSID_REQUIRE(bytes <= 160 * 1024);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mr_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return NSID_ELAUNCH;
    configured = true;
  }
  NSID_LAUNCH(mr_bwd_kernel, dim3(B), dim3(256), bytes, static_cast<hipStream_t>(stream), du, idx, argmax, N, C,
                     k, dy);
  return nsid_launch_status();
}
