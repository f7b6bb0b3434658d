// Does the 2-way bank conflict of gemm.hip's staging WRITES cost anything? ("What's weak" item 6: SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.22-0.33 on the 32-deep-stage kernels; csrc/gemm.hip TileGeom argues the replays hide under the 13 cycles a
// ds_write_b128 spends moving its operands.)
// A workgroup of 256 threads runs the staging pattern of a 128-row, 32-deep bf16 stage -- thread t writes the 16-byte chunk (t & 3) of
// row (t >> 2) and of row 64 + (t >> 2): two ds_write_b128 per stage -- then the fragment reads and MFMAs of that stage (8 ds_read_b128
// and 16 MFMAs per wave), REPS times, with the row pitch
//   96 bytes  (the kernel's: reads conflict-free, writes 2-way),
//   80 bytes  (writes conflict-free in the 8-lane groups, reads 3-way: what round 1 had),
//   and a write-only / read-only split of the same loops.
// Cycles per stage from s_memtime (median over workgroups), one and three workgroups per CU.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/lds_stage_write_bench tools/lds_stage_write_bench.hip && tools/_bin/lds_stage_write_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PITCH, bool WR, bool RD>
__global__ __launch_bounds__(256) void stage_kernel(unsigned long long* stamps, float* sink, int reps) {
  __shared__ __attribute__((aligned(16))) char lds[2][128 * 160];
  const int t = threadIdx.x, lane = t & 63, lr = lane & 15, rq = lane >> 4, wave = t >> 6;
  f32x4 v = {(float)t, 1.f, 2.f, 3.f};
  f32x4 acc[4] = {};
  for (int i = t; i < 2 * 128 * 160 / 16; i += 256) reinterpret_cast<f32x4*>(&lds[0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    char* img = lds[r & 1];
    if (WR) {
      *reinterpret_cast<f32x4*>(img + (t >> 2) * PITCH + (t & 3) * 16) = v;
      *reinterpret_cast<f32x4*>(img + (64 + (t >> 2)) * PITCH + (t & 3) * 16) = v;
      v[0] += 1.f;
    }
    __syncthreads();
    if (RD) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(img + (32 * wave + 16 * a + lr) * PITCH + rq * 16);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bf16x8 fb = *reinterpret_cast<const bf16x8*>(img + (16 * ((b + 2 * wave) & 7) + lr) * PITCH + rq * 16);
          acc[(2 * a + b) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[(2 * a + b) & 3], 0, 0, 0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) stamps[blockIdx.x] = t1 - t0;
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v[0] == -1.f) sink[t] = acc[0][0];
}

template <int PITCH, bool WR, bool RD>
double run(int wgs, int reps, unsigned long long* d_st, float* d_sink) {
  std::vector<unsigned long long> h(wgs);
  for (int w = 0; w < 3; ++w) stage_kernel<PITCH, WR, RD><<<wgs, 256>>>(d_st, d_sink, reps);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), d_st, wgs * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return (double)h[wgs / 2] / reps;
}

int main() {
  unsigned long long* d_st; float* d_sink;
  CK(hipMalloc(&d_st, 4096 * 8)); CK(hipMalloc(&d_sink, 4096));
  const int reps = 4000;
  for (int per_cu : {1, 3}) {
    const int wgs = 256 * per_cu;
    printf("{\"workgroups_per_cu\": %d, \"cycles_per_stage\": {", per_cu);
    printf("\"pitch96_write+read\": %.1f, ", run<96, true, true>(wgs, reps, d_st, d_sink));
    printf("\"pitch80_write+read\": %.1f, ", run<80, true, true>(wgs, reps, d_st, d_sink));
    printf("\"pitch96_write_only\": %.1f, ", run<96, true, false>(wgs, reps, d_st, d_sink));
    printf("\"pitch80_write_only\": %.1f, ", run<80, true, false>(wgs, reps, d_st, d_sink));
    printf("\"pitch128_write_only\": %.1f, ", run<128, true, false>(wgs, reps, d_st, d_sink));
    printf("\"pitch96_read_only\": %.1f, ", run<96, false, true>(wgs, reps, d_st, d_sink));
    printf("\"pitch80_read_only\": %.1f}}\n", run<80, false, true>(wgs, reps, d_st, d_sink));
  }
  return 0;
}
