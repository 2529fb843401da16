// Microbenchmark (round 5, for DESIGN section 7): what does JOINING a side stream cost the launch stream?  The KRN step has two joins (weight copies
// at the start of forward, weight gradients at the end of backward); both are events today and show 6-7 us gaps on the launch queue.
// A launch stream runs groups of five dependent ~10 us kernels; a side stream runs one short kernel per group that has long finished when the
// launch stream reaches the join behind its 4th kernel (so only the MECHANISM is timed, not waiting):
//   0 none        no join
//   1 event       hipEventRecord(side) + hipStreamWaitEvent(launch)           (event: hipEventDisableTiming)
//   2 event-nf    the same, hipEventDisableSystemFence as well                  (what the plan uses now)
//   3 word        a one-wave kernel on the side stream stores a serial, a one-wave gate kernel on the launch stream spins on it
//   4 word-entry  the side stream stores the serial, the launch stream's NEXT kernel spins on it with its first wave before doing its work
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench_join.hip -o scratch/ubench_join && ./scratch/ubench_join
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void work(float* data, int iters, const unsigned* wait_flag, unsigned val) {
  if (wait_flag) {   // every wave waits for the word itself (an acquire per wave: what it then reads was released before the store)
    while ((int)(__hip_atomic_load(wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - val) < 0) __builtin_amdgcn_s_sleep(2);
  }
  float v = data[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  data[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ void set_flag(unsigned* flag, unsigned val) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void gate(const unsigned* flag, unsigned val) {
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - val) < 0) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
  const int GROUPS = argc > 1 ? atoi(argv[1]) : 200;
  const int WGS = 512, ITERS = 650, SIDE_WGS = 64, SIDE_ITERS = 300;
  float *d, *ds; unsigned* flag;
  CK(hipMalloc(&d, WGS * 256 * sizeof(float))); CK(hipMemset(d, 0, WGS * 256 * sizeof(float)));
  CK(hipMalloc(&ds, SIDE_WGS * 256 * sizeof(float))); CK(hipMemset(ds, 0, SIDE_WGS * 256 * sizeof(float)));
  CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256));
  CK(hipDeviceSynchronize());
  hipStream_t st, side; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  std::vector<hipEvent_t> ev(GROUPS), evnf(GROUPS), fk(GROUPS);
  for (int i = 0; i < GROUPS; ++i) {
    CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evnf[i], hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&fk[i], hipEventDisableTiming | hipEventDisableSystemFence));
  }
  const char* names[5] = {"none", "event", "event-nf", "word", "word-entry"};
  double base = 0;
  unsigned serial = 0;
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 5; ++mode) {
    CK(hipDeviceSynchronize());
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(t0, st));
      for (int g = 0; g < GROUPS; ++g) {
        ++serial;
        // the side kernel of this group starts behind the PREVIOUS group's last kernel (a fork that is not part of the measurement: it is the
        // same in every mode, baseline included) and is done long before the join
        hipLaunchKernelGGL(work, dim3(WGS), dim3(256), 0, st, d, ITERS, (const unsigned*)nullptr, 0u);
        CK(hipEventRecord(fk[g], st)); CK(hipStreamWaitEvent(side, fk[g], 0));
        hipLaunchKernelGGL(work, dim3(SIDE_WGS), dim3(256), 0, side, ds, SIDE_ITERS, (const unsigned*)nullptr, 0u);
        if (mode == 1 || mode == 2) CK(hipEventRecord((mode == 2 ? evnf : ev)[g], side));
        if (mode == 3 || mode == 4) hipLaunchKernelGGL(set_flag, dim3(1), dim3(64), 0, side, flag, serial);
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(work, dim3(WGS), dim3(256), 0, st, d, ITERS, (const unsigned*)nullptr, 0u);
        // ---- the join
        if (mode == 1 || mode == 2) CK(hipStreamWaitEvent(st, (mode == 2 ? evnf : ev)[g], 0));
        if (mode == 3) hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, st, flag, serial);
        hipLaunchKernelGGL(work, dim3(WGS), dim3(256), 0, st, d, ITERS, mode == 4 ? flag : (const unsigned*)nullptr, serial);
      }
      CK(hipEventRecord(t1, st));
      CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(side));
      float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
      if (pass == 1) {
        const double per = ms * 1000.0 / GROUPS;
        if (mode == 0) base = per;
        printf("%-12s %8.2f us per group of 5 launches   join cost %6.2f us\n", names[mode], per, per - base);
      }
    }
  }
  return 0;
}
