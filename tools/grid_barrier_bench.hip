// Gate for a clip-resident persistent training kernel (task 1): what does ONE grid-wide BatchNorm-statistics exchange
// cost inside a launch, in the regime that matters — one workgroup per clip, activations in LDS (nothing dirty behind the barrier),
// only 2*C floats per workgroup published — against what it would replace, a kernel boundary plus a reload of the clip?
//
//   A  persistent kernel, G workgroups, P phases back to back: every workgroup publishes C2 floats (write-through, `sc1`), arrives on
//      its group's counter (group = blockIdx % 8: one XCD under round-robin placement, speed only); the group's LAST arriver sums the
//      group's partials in a fixed order and publishes the group sum, arrives on the top counter; every workgroup polls the top
//      counter (one lane, relaxed `sc1` loads + s_sleep), then reads the 8 group sums (`sc1` loads) and adds them in a fixed order:
//      the result is bit-identical in every workgroup and independent of arrival order. Every spin is bounded (abort word).
//   A' the same with `overlap` ticks of work between arrive and wait (split-phase: the other clip's phase hides the latency).
//   A2 two such kernels on two streams (the two views), each G = 256 workgroups with <= 80 KB of LDS: both resident on every CU.
//   B  the launch-per-phase alternative, captured in a hipGraph: P x { phase kernel: G workgroups load their clip (32 KB) into LDS,
//      store it back, write C2 partial floats;  finalize kernel: one workgroup reduces G x C2 floats } — what the step does today.
// Every reduced value is checked in every workgroup (exact integer-valued floats) under uneven load (`skew`), with the consumer's
// lines pre-read (L1-warm), as the guide's hand-off test rules ask.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/grid_barrier_bench tools/grid_barrier_bench.hip
//   tools/_bin/grid_barrier_bench            (prints one JSON object per configuration)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NGROUP = 8;
constexpr int LINE = 32;                        // u32 words per 128-byte line: every polled word on a line of its own
// control block (zeroed by a memset node before every launch): [0 .. 8) group counters, 8 top counter, 9 abort, 10 mismatches, 11 timeouts
constexpr int CTRL_WORDS = 16 * LINE;
constexpr unsigned SPIN_LIMIT = 1u << 21;

struct Args {
  unsigned* ctrl;          // CTRL_WORDS
  float* part;             // [P][G][C2]
  float* xpart;            // [P][8][C2]
  unsigned long long* stamps;   // [G][2]: summed wait ticks, max wait ticks (100 MHz)
  int P, G, C2;
  int work, skew, overlap; // ticks of the 100 MHz clock: work per phase, extra work of every 13th workgroup-phase, work between arrive and wait
  int verify;
};

__device__ __forceinline__ void spin_ticks(int ticks) {
  if (ticks <= 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ float val(int w, int c, int p) { return (float)((w * 31 + c * 7 + p * 3) & 1023); }

// ---- the exchange. Called by wave 0 only (64 lanes); C2 in {128, 256}: a lane owns C2 / 64 consecutive floats.
template <int C2>
__device__ __forceinline__ void arrive(const Args& a, int p, int w, const float* mine /* this lane's C2/64 floats */, int lane) {
  constexpr int V = C2 / 64;                    // 2 or 4 floats per lane
  const int g = w % NGROUP, gsize = a.G / NGROUP;
  float* dst = a.part + ((long)p * a.G + w) * C2 + lane * V;
#pragma unroll
  for (int e = 0; e < V; e += 2) {              // 8-byte write-through stores (global_store_dwordx2 sc1)
    unsigned long long bits;
    const f32x2 v2{mine[e], mine[e + 1]};
    __builtin_memcpy(&bits, &v2, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + e), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(a.ctrl + g * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old == (unsigned)(gsize * (p + 1) - 1)) { // last arriver of the group: fixed-order sum of the group's partials
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int j = 0; j < gsize; ++j) {
      const float* src = a.part + ((long)p * a.G + g + NGROUP * j) * C2 + lane * V;
#pragma unroll
      for (int e = 0; e < V; e += 2) {
        const unsigned long long bits =
            __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x2 v2;
        __builtin_memcpy(&v2, &bits, 8);
        acc[e] += v2[0];
        acc[e + 1] += v2[1];
      }
    }
    float* xd = a.xpart + ((long)p * NGROUP + g) * C2 + lane * V;
#pragma unroll
    for (int e = 0; e < V; e += 2) {
      unsigned long long bits;
      const f32x2 v2{acc[e], acc[e + 1]};
      __builtin_memcpy(&bits, &v2, 8);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(xd + e), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(a.ctrl + NGROUP * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// returns false on timeout / abort; out: this lane's C2/64 reduced floats
template <int C2>
__device__ __forceinline__ bool wait_reduce(const Args& a, int p, float* out, int lane) {
  constexpr int V = C2 / 64;
  const unsigned want = (unsigned)(NGROUP * (p + 1));
  bool ok = true;
  if (lane == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(a.ctrl + NGROUP * LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 63u) == 0 &&
          (spins > SPIN_LIMIT || __hip_atomic_load(a.ctrl + (NGROUP + 1) * LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        __hip_atomic_store(a.ctrl + (NGROUP + 1) * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.ctrl + (NGROUP + 3) * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
    }
  }
  ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
  if (!ok) return false;
#pragma unroll
  for (int e = 0; e < V; ++e) out[e] = 0.f;
  for (int g = 0; g < NGROUP; ++g) {            // fixed order: identical bits in every workgroup
    const float* src = a.xpart + ((long)p * NGROUP + g) * C2 + lane * V;
#pragma unroll
    for (int e = 0; e < V; e += 2) {
      const unsigned long long bits =
          __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f32x2 v2;
      __builtin_memcpy(&v2, &bits, 8);
      out[e] += v2[0];
      out[e + 1] += v2[1];
    }
  }
  return true;
}

template <int C2>
__global__ void persist_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);          // [C2] the reduced vector, for the other waves
  const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  constexpr int V = C2 / 64;
  unsigned long long wsum = 0, wmax = 0;
  bool dead = false;
  for (int p = 0; p < a.P && !dead; ++p) {
    spin_ticks(a.work + (((w * 7 + p) % 13 == 0) ? a.skew : 0));
    if (wave == 0) {
      if (a.verify) {        // pre-read the lines this wave will be handed (L1-warm consumer): a stale copy would then be seen
        float sink = 0.f;
        for (int g = 0; g < NGROUP; ++g) sink += a.xpart[((long)p * NGROUP + g) * C2 + lane * V];
        if (sink == 12345.678f) a.stamps[0] = 1;
      }
      float mine[V], out[V];
#pragma unroll
      for (int e = 0; e < V; ++e) mine[e] = val(w, lane * V + e, p);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      arrive<C2>(a, p, w, mine, lane);
      const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
      spin_ticks(a.overlap);
      const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
      const bool ok = wait_reduce<C2>(a, p, out, lane);
      const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
      const unsigned long long cost = (ta - t0) + (t2 - t1);       // what the exchange cost THIS workgroup (arrival skew included)
      wsum += cost;
      wmax = cost > wmax ? cost : wmax;
      if (ok) {
#pragma unroll
        for (int e = 0; e < V; ++e) red[lane * V + e] = out[e];
      } else {
        red[0] = -1.f;
      }
    }
    __syncthreads();
    if (red[0] < 0.f) dead = true;
    if (a.verify && !dead && t < C2) {
      float expect = 0.f;
      for (int ww = 0; ww < a.G; ++ww) expect += val(ww, t, p);     // exact: integers below 2^24
      if (red[t] != expect) __hip_atomic_fetch_add(a.ctrl + (NGROUP + 2) * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (t == 0) { a.stamps[2 * w] = wsum; a.stamps[2 * w + 1] = wmax; }
}

// ---- B: launch per phase
__global__ void phase_kernel(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ part, int C2, int p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* img = reinterpret_cast<float4*>(smem);                    // 32 KB = 2048 float4
  const int w = blockIdx.x, t = threadIdx.x;
  const float4* src = reinterpret_cast<const float4*>(in) + (long)w * 2048;
  float4* dst = reinterpret_cast<float4*>(out) + (long)w * 2048;
  for (int i = t; i < 2048; i += blockDim.x) img[i] = src[i];
  __syncthreads();
  for (int i = t; i < 2048; i += blockDim.x) { float4 v = img[i]; v.x += 1.f; dst[i] = v; }
  if (t < C2) part[(long)w * C2 + t] = val(w, t, p);
}
__global__ void finalize_kernel(const float* __restrict__ part, float* __restrict__ red, int G, int C2) {
  const int t = threadIdx.x;
  if (t < C2) {
    float s = 0.f;
    for (int w = 0; w < G; ++w) s += part[(long)w * C2 + t];
    red[t] = s;
  }
}

struct Res { float us_total; double wait_avg_us, wait_max_us; unsigned mism, tmo; };

template <int C2>
static Res run_persist(int G, int threads, int lds, int P, int work, int skew, int overlap, int verify, int nstreams, int reps) {
  std::vector<hipStream_t> st(nstreams);
  std::vector<Args> args(nstreams);
  for (int s = 0; s < nstreams; ++s) {
    CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    Args& a = args[s];
    CK(hipMalloc(&a.ctrl, CTRL_WORDS * 4));
    CK(hipMalloc(&a.part, (size_t)P * G * C2 * 4));
    CK(hipMalloc(&a.xpart, (size_t)P * NGROUP * C2 * 4));
    CK(hipMalloc(&a.stamps, (size_t)G * 16));
    CK(hipMemset(a.part, 0xff, (size_t)P * G * C2 * 4));            // poison
    CK(hipMemset(a.xpart, 0xff, (size_t)P * NGROUP * C2 * 4));
    a.P = P; a.G = G; a.C2 = C2; a.work = work; a.skew = skew; a.overlap = overlap; a.verify = verify;
  }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persist_kernel<C2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Res r{};
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    for (int s = 0; s < nstreams; ++s) CK(hipMemsetAsync(args[s].ctrl, 0, CTRL_WORDS * 4, st[s]));
    for (int s = 0; s < nstreams; ++s) CK(hipStreamSynchronize(st[s]));
    CK(hipEventRecord(e0, st[0]));
    if (nstreams > 1) CK(hipStreamWaitEvent(st[1], e0, 0));
    for (int s = 0; s < nstreams; ++s) hipLaunchKernelGGL(persist_kernel<C2>, dim3(G), dim3(threads), lds, st[s], args[s]);
    hipEvent_t ej;
    if (nstreams > 1) { CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming)); CK(hipEventRecord(ej, st[1])); CK(hipStreamWaitEvent(st[0], ej, 0)); }
    CK(hipEventRecord(e1, st[0]));
    CK(hipStreamSynchronize(st[0]));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
    for (int s = 0; s < nstreams; ++s) {
      unsigned ctrl[CTRL_WORDS];
      CK(hipMemcpy(ctrl, args[s].ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
      r.mism += ctrl[(NGROUP + 2) * LINE];
      r.tmo += ctrl[(NGROUP + 3) * LINE];
    }
  }
  r.us_total = best * 1e3f;
  std::vector<unsigned long long> stamps(2 * G);
  CK(hipMemcpy(stamps.data(), args[0].stamps, (size_t)G * 16, hipMemcpyDeviceToHost));
  double s = 0, m = 0;
  for (int w = 0; w < G; ++w) { s += stamps[2 * w]; m = stamps[2 * w + 1] > m ? stamps[2 * w + 1] : m; }
  r.wait_avg_us = s / G / P / 100.0;
  r.wait_max_us = m / 100.0;
  for (int s2 = 0; s2 < nstreams; ++s2) {
    CK(hipFree(args[s2].ctrl)); CK(hipFree(args[s2].part)); CK(hipFree(args[s2].xpart)); CK(hipFree(args[s2].stamps));
    CK(hipStreamDestroy(st[s2]));
  }
  return r;
}

static void report(const char* name, int G, int threads, int lds, int P, int C2, int work, int skew, int overlap, int nstreams, int verify) {
  Res r = C2 == 128 ? run_persist<128>(G, threads, lds, P, work, skew, overlap, verify, nstreams, verify ? 2 : 5)
                    : run_persist<256>(G, threads, lds, P, work, skew, overlap, verify, nstreams, verify ? 2 : 5);
  const double per = (r.us_total - (double)P * (work + overlap) / 100.0) / P;
  printf("{\"cfg\": \"%s\", \"G\": %d, \"threads\": %d, \"lds\": %d, \"P\": %d, \"C2\": %d, \"work_us\": %.2f, \"skew_us\": %.2f, \"overlap_us\": %.2f, "
         "\"streams\": %d, \"verify\": %d, \"total_us\": %.1f, \"per_phase_minus_work_us\": %.2f, \"wait_avg_us\": %.2f, \"wait_max_us\": %.2f, "
         "\"mismatches\": %u, \"timeouts\": %u}\n",
         name, G, threads, lds, P, C2, work / 100.0, skew / 100.0, overlap / 100.0, nstreams, verify, r.us_total, per, r.wait_avg_us, r.wait_max_us,
         r.mism, r.tmo);
  fflush(stdout);
}

static void run_launches(int G, int P, int C2, bool with_finalize) {
  float *a, *b, *part, *red;
  CK(hipMalloc(&a, (size_t)G * 32768)); CK(hipMalloc(&b, (size_t)G * 32768));
  CK(hipMalloc(&part, (size_t)G * C2 * 4)); CK(hipMalloc(&red, C2 * 4));
  CK(hipMemset(a, 0, (size_t)G * 32768));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int p = 0; p < P; ++p) {
    hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(512), 32768, s, (p & 1) ? b : a, (p & 1) ? a : b, part, C2, p);
    if (with_finalize) hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, part, red, G, C2);
  }
  CK(hipStreamEndCapture(s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("{\"cfg\": \"B launch-per-phase%s\", \"G\": %d, \"P\": %d, \"C2\": %d, \"total_us\": %.1f, \"per_phase_us\": %.2f}\n",
         with_finalize ? " + finalize launch" : "", G, P, C2, best * 1e3f, best * 1e3f / P);
  fflush(stdout);
  CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(s));
  CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(part)); CK(hipFree(red));
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 64;
  // correctness first: uneven load, L1-warm consumers, every word checked
  report("A verify 256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 300, 1500, 0, 1, 1);
  report("A verify 512x512thr 72KB", 512, 512, 72 * 1024, P, 128, 300, 1500, 0, 1, 1);
  report("A2 verify 2x256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 300, 1500, 0, 2, 1);
  report("A verify 256x512thr 150KB C2=256", 256, 512, 150 * 1024, P, 256, 300, 1500, 0, 1, 1);
  // timing: barriers back to back
  report("A back-to-back 256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 0, 0, 0, 1, 0);
  report("A back-to-back 256x512thr 150KB", 256, 512, 150 * 1024, P, 128, 0, 0, 0, 1, 0);
  report("A back-to-back 256x512thr 72KB C2=256", 256, 512, 72 * 1024, P, 256, 0, 0, 0, 1, 0);
  report("A back-to-back 512x512thr 72KB", 512, 512, 72 * 1024, P, 128, 0, 0, 0, 1, 0);
  report("A back-to-back 512x256thr 72KB", 512, 256, 72 * 1024, P, 128, 0, 0, 0, 1, 0);
  report("A2 back-to-back 2x256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 0, 0, 0, 2, 0);
  // with 5 us of work per phase, even and uneven
  report("A work5 256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 500, 0, 0, 1, 0);
  report("A work5 skew3 256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 500, 300, 0, 1, 0);
  report("A2 work5 2x256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 500, 0, 0, 2, 0);
  report("A2 work5 skew3 2x256x512thr 72KB", 256, 512, 72 * 1024, P, 128, 500, 300, 0, 2, 0);
  // split phase: 5 us of work before, 5 us of independent work between arrive and wait
  report("A' split work5 overlap5 256x512thr 150KB", 256, 512, 150 * 1024, P, 128, 500, 0, 500, 1, 0);
  report("A' split work5 overlap3 256x512thr 150KB", 256, 512, 150 * 1024, P, 128, 500, 0, 300, 1, 0);
  report("A' split work5 skew3 overlap5 256x512thr 150KB", 256, 512, 150 * 1024, P, 128, 500, 300, 500, 1, 0);
  // B
  run_launches(256, P, 128, false);
  run_launches(256, P, 128, true);
  run_launches(512, P, 128, true);
  return 0;
}
