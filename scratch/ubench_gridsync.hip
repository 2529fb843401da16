// Microbenchmark (round 5, groundwork for the 7x7 persistent kernel of DESIGN section 7): what does a grid-wide barrier between two
// dependent phases cost on the MI355X, against a kernel boundary?  Persistent workgroups (1 or 2 per CU) run PHASES phases; each phase
// does a token amount of work (one dependent global load + one atomic, like a BatchNorm statistic) and then meets the others.
//   flat:  one counter, every workgroup adds 1 and spins on a generation word
//   tree:  per-XCD counters (workgroup k runs on XCD k % 8), the last arriver of an XCD adds to the global counter, all spin on the generation
// Baseline: the same token work as PHASES back-to-back kernel launches on one stream.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench_gridsync.hip -o scratch/ubench_gridsync && ./scratch/ubench_gridsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar { unsigned cnt[8 * 32]; unsigned top; unsigned gen; };   // per-XCD counters 128 B apart

__device__ __forceinline__ void token_work(float* data, float* acc, int phase) {
  const float v = data[(blockIdx.x * 64 + (threadIdx.x & 63) + phase * 4096) & 65535];   // one dependent load
  if (threadIdx.x == 0) atomicAdd(acc + (phase & 63), v);                                // one statistic
}

template <int TREE>
__global__ void persistent(float* data, float* acc, Bar* bar, int phases) {
  const unsigned nwg = gridDim.x;
  for (int p = 0; p < phases; ++p) {
    token_work(data, acc, p);
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned target = (unsigned)(p + 1);
      if (TREE) {
        const int xcd = blockIdx.x & 7;
        const unsigned per = (nwg + 7 - xcd) / 8;
        if (atomicAdd(&bar->cnt[xcd * 32], 1u) == per * target - 1) {
          if (atomicAdd(&bar->top, 1u) == 8 * target - 1) __hip_atomic_store(&bar->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        if (atomicAdd(&bar->top, 1u) == nwg * target - 1) __hip_atomic_store(&bar->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      while (__hip_atomic_load(&bar->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

__global__ void one_phase(float* data, float* acc, int phase) { token_work(data, acc, phase); }

int main() {
  float *data, *acc; Bar* bar;
  CK(hipMalloc(&data, 65536 * 4)); CK(hipMalloc(&acc, 64 * 4)); CK(hipMalloc(&bar, sizeof(Bar)));
  CK(hipMemset(data, 0, 65536 * 4)); CK(hipMemset(acc, 0, 256));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int PH = 2000;
  for (int nwg : {256, 512}) {
    for (int tree = 0; tree < 2; ++tree) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(bar, 0, sizeof(Bar)));
        CK(hipEventRecord(a));
        if (tree) hipLaunchKernelGGL(persistent<1>, dim3(nwg), dim3(256), 0, 0, data, acc, bar, PH);
        else hipLaunchKernelGGL(persistent<0>, dim3(nwg), dim3(256), 0, 0, data, acc, bar, PH);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
      }
      printf("persistent, %3d workgroups, %s barrier: %.2f us per phase\n", nwg, tree ? "tree" : "flat", best * 1e3f / PH);
    }
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      for (int p = 0; p < PH; ++p) hipLaunchKernelGGL(one_phase, dim3(nwg), dim3(256), 0, 0, data, acc, p);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    printf("kernel per phase, %3d workgroups:            %.2f us per phase\n", nwg, best * 1e3f / PH);
  }
  return 0;
}
