// Microbenchmark (round 5): what does handing work to a second stream cost the LAUNCH stream?  The KRN plan forks its weight gradients
// 22 times per step; the kernel trace (profiles/r5_krn_chain.txt) shows a ~4.8 us hole on the launch queue behind every fork and none
// anywhere else.  Here: a launch stream runs GROUPS x 5 dependent ~10 us kernels; after the 5th kernel of every group a short kernel
// is handed to a side stream, ordered behind that 5th kernel by one of
//   0 none      no side work at all (baseline)
//   1 record    hipEventRecord(launch) + hipStreamWaitEvent(side)                      event: hipEventDisableTiming
//   2 record-nf the same, event created with hipEventDisableSystemFence as well
//   3 stop      the 5th kernel launched with hipExtLaunchKernelGGL(stopEvent) + hipStreamWaitEvent(side)
//   4 stop-nf   the same, event with hipEventDisableSystemFence
//   5 value     hipStreamWriteValue32(launch) + hipStreamWaitValue32(side)
//   6 flag      NO packet on the launch queue: the last workgroup of the 5th kernel stores a flag (agent-scope release), the side stream
//               runs a one-wave kernel that spins on it
//  11 flag-kernel  a one-wave kernel on the launch queue stores the flag;  12 flag-next: the NEXT launch-stream kernel stores it at its entry
//               (it runs behind a barrier bit: its first instruction proves the producer complete); 13 / 14: those two with no side kernel
//   7 unordered the side kernels with no ordering at all (the price of the side WORK);  8-10: orderings 4 / 2 / 6 with no side kernel
// Reported: launch-stream time per group minus the baseline = cost of one fork.
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench_fork.hip -o scratch/ubench_fork && ./scratch/ubench_fork
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ~10 us of dependent work on 512 workgroups; with `flag` != nullptr the last workgroup to finish publishes `val`
__global__ void work(float* data, int iters, unsigned* ticket, unsigned* flag, unsigned val, unsigned* entry_flag) {
  // entry_flag: this kernel runs behind a barrier bit, so its first instruction already proves that everything before it on the queue
  // has completed and been released at agent scope
  if (entry_flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(entry_flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float v = data[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  data[blockIdx.x * 256 + threadIdx.x] = v;
  if (flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
        *ticket = 0;
        __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__global__ void side_work(float* data, int iters) {
  float v = data[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
  data[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ void set_flag(unsigned* flag, unsigned val) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void poll(const unsigned* flag, unsigned val) {
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - val) < 0) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
  const int GROUPS = argc > 1 ? atoi(argv[1]) : 200;
  const int WGS = 512, ITERS = 650, SIDE_WGS = 64, SIDE_ITERS = 600;
  float *d, *ds; unsigned *ticket, *flag, *sig;
  CK(hipMalloc(&d, WGS * 256 * sizeof(float))); CK(hipMemset(d, 0, WGS * 256 * sizeof(float)));
  CK(hipMalloc(&ds, SIDE_WGS * 256 * sizeof(float))); CK(hipMemset(ds, 0, SIDE_WGS * 256 * sizeof(float)));
  CK(hipMalloc(&ticket, 256)); CK(hipMemset(ticket, 0, 256));
  flag = ticket + 32;
  if (hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) != hipSuccess) { sig = nullptr; (void)hipGetLastError(); }
  hipStream_t st, side; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t t0, t1, join; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  std::vector<hipEvent_t> ev(GROUPS), evnf(GROUPS);
  for (int i = 0; i < GROUPS; ++i) {
    CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&evnf[i], hipEventDisableTiming | hipEventDisableSystemFence));
  }
  const char* names[15] = {"none", "record", "record-nf", "stop", "stop-nf", "value", "flag", "unordered", "stop-nf-only", "record-nf-only", "flag-only", "flag-kernel", "flag-next", "flag-kernel-only", "flag-next-only"};
  double base = 0;
  unsigned serial = 0;
  for (int rep = 0; rep < 2; ++rep)
  for (int mode_ = 0; mode_ < 15; ++mode_) {
    // 7: side kernels with no ordering at all (what the side WORK costs); 8 / 9 / 10: the ordering alone, no side kernel behind it
    const int mode = mode_ == 8 ? 4 : mode_ == 9 ? 2 : mode_ == 10 ? 6 : mode_ == 13 ? 11 : mode_ == 14 ? 12 : mode_;
    const bool side_kernel = mode_ != 0 && (mode_ < 8 || mode_ == 11 || mode_ == 12);
    if (mode == 5 && !sig) { printf("%-10s signal memory unavailable\n", names[mode_]); continue; }
    if (sig) CK(hipMemset(sig, 0, 8));
    CK(hipDeviceSynchronize());
    for (int pass = 0; pass < 2; ++pass) {     // pass 0 warms up
      CK(hipEventRecord(t0, st));
      const auto h0 = std::chrono::steady_clock::now();
      for (int gI = 0; gI < GROUPS; ++gI) {
        // 12 flag-next: the FIRST kernel of the next group publishes the previous fork's serial at its entry
        for (int k = 0; k < 4; ++k)
          hipLaunchKernelGGL(work, dim3(WGS), dim3(256), 0, st, d, ITERS, ticket, (unsigned*)nullptr, serial, (mode == 12 && k == 0) ? flag : (unsigned*)nullptr);
        ++serial;
        std::vector<hipEvent_t>& E = (mode == 2 || mode == 4) ? evnf : ev;
        if (mode == 3 || mode == 4)
          hipExtLaunchKernelGGL(work, dim3(WGS), dim3(256), 0, st, nullptr, E[gI], 0, d, ITERS, ticket, (unsigned*)nullptr, 0u, (unsigned*)nullptr);
        else if (mode == 6)
          hipLaunchKernelGGL(work, dim3(WGS), dim3(256), 0, st, d, ITERS, ticket, flag, serial, (unsigned*)nullptr);
        else
          hipLaunchKernelGGL(work, dim3(WGS), dim3(256), 0, st, d, ITERS, ticket, (unsigned*)nullptr, 0u, (unsigned*)nullptr);
        if (mode == 1 || mode == 2) { CK(hipEventRecord(E[gI], st)); CK(hipStreamWaitEvent(side, E[gI], 0)); }
        if (mode == 3 || mode == 4) CK(hipStreamWaitEvent(side, E[gI], 0));
        if (mode == 5) { CK(hipStreamWriteValue32(st, sig, serial, 0)); CK(hipStreamWaitValue32(side, sig, serial, hipStreamWaitValueGte, 0xffffffffu)); }
        if (mode == 11) hipLaunchKernelGGL(set_flag, dim3(1), dim3(64), 0, st, flag, serial);   // a one-wave kernel on the launch queue
        if (mode == 11 || mode == 12) hipLaunchKernelGGL(poll, dim3(1), dim3(64), 0, side, flag, serial);
        if (mode == 6) hipLaunchKernelGGL(poll, dim3(1), dim3(64), 0, side, flag, serial);
        if (side_kernel) hipLaunchKernelGGL(side_work, dim3(SIDE_WGS), dim3(256), 0, side, ds, SIDE_ITERS);
      }
      if (mode == 12) hipLaunchKernelGGL(set_flag, dim3(1), dim3(64), 0, st, flag, serial);     // the last fork has no next group
      const auto h1 = std::chrono::steady_clock::now();
      CK(hipEventRecord(t1, st));
      CK(hipEventRecord(join, side)); CK(hipStreamWaitEvent(st, join, 0));
      CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(side));
      float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
      if (pass == 1) {
        const double per = ms * 1000.0 / GROUPS;
        if (mode_ == 0) base = per;
        printf("%-14s %8.2f us per group of 5 launches   fork cost %6.2f us   (host enqueue %.2f us per group)\n", names[mode_], per, per - base,
               std::chrono::duration<double, std::micro>(h1 - h0).count() / GROUPS);
      }
    }
  }
  return 0;
}
