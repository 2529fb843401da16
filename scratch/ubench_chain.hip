// Microbenchmark (round 6): can the ~4.5 us boundary between two DEPENDENT kernels be beaten by launching them on two alternating streams
// and carrying the dependency in a device word?  Kernel k+1 is then dispatched, resident and through its independent prologue while kernel k still runs; its
// waves spin on a word that kernel k's last workgroup publishes (release fence per workgroup, done counter, acquire in the consumer).  The KRN step is a
// chain of ~125 dependent launches of 8-25 us on small maps; this is what a "no kernel boundary" version of it could gain per launch.
//   mode 0  one stream, plain launches (what the plan does)
//   mode 1  one stream, every kernel also runs the publish epilogue (cost of the epilogue alone)
//   mode 2  two alternating streams, dependency by device word (entry spin + publish)
//   mode 3  as 1, but the device-wide release is made once per XCD by its last workgroup (other workgroups: workgroup-scope release only)
//   mode 4  as 2 with the per-XCD release
// Every kernel reads the previous kernel's output buffer and writes its own (value + 1): the final value proves order and visibility.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench_chain.hip -o scratch/ubench_chain && ./scratch/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef SLEEP
#define SLEEP 8
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Sync { unsigned word; unsigned pad0[31]; unsigned done; unsigned pad1[31]; unsigned err; unsigned pad2[31]; unsigned xdone[8][32]; unsigned xall; unsigned pad3[31]; };

// each workgroup: PER_WG floats in, PER_WG floats out, `iters` dependent FMAs per element (pads the kernel to a small-map launch's duration)
template <int PER_T>
__global__ __launch_bounds__(256) void link(const float* __restrict__ in, float* __restrict__ out, int iters, Sync* sy, unsigned wait_val, unsigned pub_val,
                                            int publish, int nblocks, float* __restrict__ tab) {
  __shared__ float lt[256];
  lt[threadIdx.x] = tab[threadIdx.x];                 // the independent prologue (weights -> LDS in a real kernel)
  __syncthreads();
  if (wait_val) {                                      // entry: one thread of the workgroup spins, then the workgroup acquires
    if (threadIdx.x == 0) {                            // (first version: lane 0 of every wave, s_sleep 1: 22.6 / 32.2 / 51.8 us per link in mode 4)
      int spins = 0;
      while ((int)(__hip_atomic_load(&sy->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - wait_val) < 0) {
        __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > (1 << 22)) { sy->err = 1; break; }       // ~ seconds: never hang the box
      }
    }
    __syncthreads();
#ifndef NOACQ   // -DNOACQ: upper bound only (no device-wide acquire: correct here only because nothing re-caches the ping-pong buffers between two launches)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * PER_T;
  float v[PER_T];
#pragma unroll
  for (int i = 0; i < PER_T; i += 4) *reinterpret_cast<float4*>(v + i) = *reinterpret_cast<const float4*>(in + base + i);
  const float one = lt[threadIdx.x];
#pragma unroll
  for (int i = 0; i < PER_T; ++i) {
    float x = v[i];
    for (int k = 0; k < iters; ++k) x = x * one + 0.f;
    v[i] = x + one;
  }
#pragma unroll
  for (int i = 0; i < PER_T; i += 4) *reinterpret_cast<float4*>(out + base + i) = *reinterpret_cast<float4*>(v + i);
  if (publish == 2) {
    // release per XCD instead of per workgroup: a workgroup only waits for its stores to reach its XCD's L2 (workgroup-scope release = s_waitcnt), the LAST
    // workgroup of each XCD (round-robin dispatch: workgroup i runs on XCD i % 8) writes that L2 back once, the last of those publishes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
      const int x = blockIdx.x & 7;
      const unsigned mine = (unsigned)((nblocks - x + 7) >> 3);            // workgroups of this launch on XCD x
      const unsigned d = __hip_atomic_fetch_add(&sy->xdone[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == mine * pub_val - 1u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned a = __hip_atomic_fetch_add(&sy->xall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == 8u * pub_val - 1u) __hip_atomic_store(&sy->word, pub_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else if (publish) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this workgroup's stores are visible device-wide before it counts itself done
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned d = __hip_atomic_fetch_add(&sy->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == (unsigned)nblocks * pub_val - 1u) __hip_atomic_store(&sy->word, pub_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main(int argc, char** argv) {
  const int CHAIN = 120;
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  float* tab; CK(hipMalloc(&tab, 1024)); { std::vector<float> h(256, 1.0f); CK(hipMemcpy(tab, h.data(), 1024, hipMemcpyHostToDevice)); }
  Sync* sy; CK(hipMalloc(&sy, sizeof(Sync)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  constexpr int PER_T = 16;
  for (int wgs : {256, 512, 1024}) {
    const size_t n = (size_t)wgs * 256 * PER_T;      // 4 / 8 / 16 MB per buffer: a small-map tensor
    float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    for (int iters : {40, 160}) {
      for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f; float last = -1.f; unsigned err = 0;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipMemsetAsync(a, 0, n * 4, sa)); CK(hipMemsetAsync(sy, 0, sizeof(Sync), sa)); CK(hipStreamSynchronize(sa));
          CK(hipEventRecord(e0, sa));
          const bool two = mode == 2 || mode == 4;
          if (two) CK(hipStreamWaitEvent(sb, e0, 0));
          for (int k = 0; k < CHAIN; ++k) {
            const float* in = (k & 1) ? b : a; float* out = (k & 1) ? a : b;
            hipStream_t st = (two && (k & 1)) ? sb : sa;
            hipLaunchKernelGGL(link<PER_T>, dim3(wgs), dim3(256), 0, st, in, out, iters, sy, two ? (unsigned)k : 0u, (unsigned)(k + 1), mode == 0 ? 0 : (mode >= 3 ? 2 : 1), wgs, tab);
          }
          if (two) { CK(hipEventRecord(e1, sb)); CK(hipStreamWaitEvent(sa, e1, 0)); }
          CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
          const float* fin = (CHAIN & 1) ? b : a;
          CK(hipMemcpy(&last, fin + n - 1, 4, hipMemcpyDeviceToHost));
          { std::vector<float> hf(n); CK(hipMemcpy(hf.data(), fin, n * 4, hipMemcpyDeviceToHost)); for (size_t i = 0; i < n; ++i) if (hf[i] != (float)CHAIN) { last = -hf[i] - 1000.f; break; } }
          Sync hs; CK(hipMemcpy(&hs, sy, sizeof(Sync), hipMemcpyDeviceToHost)); err |= hs.err;
        }
        printf("wgs %4d iters %3d mode %d: %.2f us per link (chain of %d)  final value %.0f (expected %d)%s\n", wgs, iters, mode, best * 1e3 / CHAIN, CHAIN, last, CHAIN,
               err ? "  SPIN TIME-OUT" : "");
      }
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
