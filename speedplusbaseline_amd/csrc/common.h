// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the KRN/SPN/DANN hot path.
// Everything here is written for wave64 + MFMA; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spb_hip.h"
// TIMING EXPERIMENTS ONLY (scratch/build_variant.py; results are wrong): -DSPB_NO_ATOMICS turns every statistics atomic of a
// file into a no-op, -DSPB_NO_ATOMICS_W the weight-gradient atomics (sites marked SPB_ATOMIC_W)
template <typename P, typename V> __device__ __forceinline__ void spb_no_atomic(P p, V v) { *p = v; }   // a plain store keeps the computation alive
#ifdef SPB_NO_ATOMICS_W
#define SPB_ATOMIC_W(p, v) spb_no_atomic(p, v)
#else
#define SPB_ATOMIC_W(p, v) (atomicAdd)(p, v)
#endif
#ifdef SPB_NO_ATOMICS
#define atomicAdd(p, v) spb_no_atomic(p, v)
#endif

// ---------------------------------------------------------------------------------------------
// -DSPB_DET: the REPRODUCIBLE twin of the library (libspb_hip_det.so, speedplusbaseline_amd/build.py).  Every float atomicAdd of
// the KRN kernels -- BatchNorm batch sums, backward sums, split-M weight gradients -- becomes an EXACT accumulation: the target
// float slot has a shadow of four 64-bit integer windows (fixed point with least significant bits 2^0, 2^-40, 2^-80, 2^-120);
// a contribution is split exactly over the windows (a float's 24-bit mantissa straddles at most two) and added with integer
// atomics.  Integer addition is associative, so the accumulated value does not depend on the order in which workgroups arrive:
// the same inputs give the same bits in every run, whatever else ran on the device before.  spb_det_flush (krn_plan.hip) folds
// the windows into the float slot after each launch (plan: Runner::ok) -- kernels and their consumers are unchanged.
// The regions (float range -> shadow) are registered per engine / context (spb_det_register); an atomic outside every region
// falls back to the float atomic and is COUNTED (spb_det_misses), so a test can assert that a run was fully covered.
// Cost: 1-2 integer atomics per float atomic + one flush launch per kernel launch: a test / reproducibility mode, not the bench path.
#ifdef SPB_DET
struct spb_det_region_t { const float* lo; const float* hi; long long* shadow; };
#define SPB_DET_MAX_REGIONS 64    // one per engine (gradient arena) + one per (batch size, slot) context; KrnEngine.drop_context frees a slot
struct spb_det_table_t { spb_det_region_t r[SPB_DET_MAX_REGIONS]; int n; int pad; };
static __device__ spb_det_table_t g_spb_det_table;          // one copy per translation unit (no relocatable device code)
static __device__ unsigned long long g_spb_det_misses;
typedef int (*spb_det_tu_fn)(const spb_det_table_t*, unsigned long long*);
extern "C" __attribute__((visibility("hidden"))) void spb_det_register_tu(spb_det_tu_fn f);       // krn_plan.hip: the list of per-file setters (internal)
// (t != null: upload the table; misses != null: add and clear this file's miss counter)
static int spb_det_tu_sync(const spb_det_table_t* t, unsigned long long* misses) {
  if (t && hipMemcpyToSymbol(HIP_SYMBOL(g_spb_det_table), t, sizeof(*t)) != hipSuccess) { (void)hipGetLastError(); return 1; }
  if (misses) {
    unsigned long long m = 0, z = 0;
    if (hipMemcpyFromSymbol(&m, HIP_SYMBOL(g_spb_det_misses), sizeof(m)) != hipSuccess) { (void)hipGetLastError(); return 1; }
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spb_det_misses), &z, sizeof(z));
    *misses += m;
  }
  return 0;
}
namespace { struct spb_det_tu_init { spb_det_tu_init() { spb_det_register_tu(&spb_det_tu_sync); } }; static spb_det_tu_init spb_det_tu_init_; }

__device__ __forceinline__ void spb_det_add_exact(long long* sh, float v) {
  double r = (double)v;
  const double q0 = trunc(r);           r -= q0;
  const double q1 = trunc(r * 0x1p40);  r -= q1 * 0x1p-40;
  const double q2 = trunc(r * 0x1p80);  r -= q2 * 0x1p-80;
  const double q3 = trunc(r * 0x1p120);
  unsigned long long* u = reinterpret_cast<unsigned long long*>(sh);
  if (q0 != 0.0) (atomicAdd)(u + 0, (unsigned long long)(long long)q0);
  if (q1 != 0.0) (atomicAdd)(u + 1, (unsigned long long)(long long)q1);
  if (q2 != 0.0) (atomicAdd)(u + 2, (unsigned long long)(long long)q2);
  if (q3 != 0.0) (atomicAdd)(u + 3, (unsigned long long)(long long)q3);
}
template <typename P, typename V> __device__ __forceinline__ auto spb_det_atomic(P* p, V v) -> decltype((atomicAdd)(p, v)) { return (atomicAdd)(p, v); }
__device__ __forceinline__ float spb_det_atomic(float* p, float v) {
  if (v == 0.f) return 0.f;
  if (fabsf(v) < 0x1p39f) {            // (non-finite or huge: the float slot takes it directly and stays non-finite through the flush)
    const int n = g_spb_det_table.n;
    for (int i = 0; i < n; ++i) {
      const spb_det_region_t& R = g_spb_det_table.r[i];
      if (p >= R.lo && p < R.hi) { spb_det_add_exact(R.shadow + 4 * (p - R.lo), v); return 0.f; }
    }
    (atomicAdd)(&g_spb_det_misses, 1ull);
  }
  return (atomicAdd)(p, v);
}
#undef SPB_ATOMIC_W
#define SPB_ATOMIC_W(p, v) spb_det_atomic(p, v)
#define atomicAdd(p, v) spb_det_atomic(p, v)
#endif

typedef unsigned short bf16_t;  // raw 16-bit storage element: bfloat16 bits -- or IEEE half bits in the -DSPB_F16 twin library (below)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
// The 16-bit storage format is a COMPILE-TIME property of the library: libspb_hip.so stores bfloat16, its twin libspb_hip_f16.so
// (the same sources compiled with -DSPB_F16, speedplusbaseline_amd/build.py) stores IEEE half and runs v_mfma_f32_16x16x32_f16 --
// the reference's fp16 autocast recipe for SPN (train.py:101-104, trainer.py:146-181; BASELINE configs[5]).  Everything that
// touches the bits goes through the helpers of this header (bf2f / pack_bf16x2 / cvt8 / ld8 / rnd8) and SPB_MFMA16.
#ifdef SPB_F16
typedef _Float16 spb_h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 spb_h16x8 __attribute__((ext_vector_type(8)));
#define SPB_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(spb_h16x8, (a)), __builtin_bit_cast(spb_h16x8, (b)), (c), 0, 0, 0)
#else
#define SPB_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define SPB_WAVE 64

#define SPB_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)

// stream forks through a device word (elemwise.hip): store `val` behind everything `s` holds / keep `s` waiting until the word reaches `val`
void spb_fork_store(unsigned* flag, unsigned val, hipStream_t s);
void spb_fork_gate(const unsigned* flag, unsigned val, hipStream_t s, unsigned* poison_dev);   // poison_dev: see spb_fork_poison_* (elemwise.hip)
unsigned* spb_fork_poison_alloc();                   // host word a gate raises instead of trapping when it gives up (SPB_FORK_TIMEOUT_S)
void spb_fork_poison_free(unsigned* host);
unsigned* spb_fork_poison_dev(unsigned* host);
// false: order by events (stream capture, SPB_EVENT_FORKS=1, a counter-collecting profiler, or the start-up self-test failed: it runs once
// per process, on the first (from, to) pair it is asked about)
bool spb_fork_by_word(hipStream_t from, hipStream_t to);
// spb_dw_args_t::entry_flag (include/spb_hip.h): the first thread of a launch publishes "everything before me on my stream is complete".
// Any thread would do -- the dispatch sat behind a barrier bit -- and the store needs no fence of its own: the earlier launches' results
// were released at device scope when their dispatch packets completed, and whoever spins on the word only gates later dispatches.
__device__ __forceinline__ void spb_publish_entry(unsigned* flag, unsigned val) {
  if (flag != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// scalar conversions
#ifdef SPB_F16
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
// the two halves of a packed 32-bit word as floats
__device__ __forceinline__ void spb_unpack2(uint32_t u, float& lo, float& hi) {
  const spb_h16x2 h = __builtin_bit_cast(spb_h16x2, u);
  lo = (float)h[0]; hi = (float)h[1];
}
#else
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ void spb_unpack2(uint32_t u, float& lo, float& hi) { lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u); }
#endif
// float -> bf16, round-to-nearest-even: one v_cvt_pk_bf16_f32 per pair on gfx950 (the integer emulation cost 7 VALU
// ops per value and was ~20 % of the instruction stream of the streaming kernels)
typedef __bf16 spb_bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float spb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const spb_f32x2 v = {lo, hi};
#ifdef SPB_F16
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, spb_h16x2));   // v_cvt_pk_f16_f32 (RNE; overflow -> inf, as autocast)
#else
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, spb_bf16x2_hw));
#endif
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }
// value after a round trip through the storage type (what a consumer will read back)
template <typename T> __device__ __forceinline__ float rnd(float v) { return to_f<T>(from_f<T>(v)); }

// ---------------------------------------------------------------------------------------------
// 8-element vector access (16 B for bf16, 2x16 B for float).  p must be 16-byte aligned.
template <typename T> __device__ __forceinline__ void ld8(const T* p, float v[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float v[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  spb_unpack2(u.x, v[0], v[1]); spb_unpack2(u.y, v[2], v[3]); spb_unpack2(u.z, v[4], v[5]); spb_unpack2(u.w, v[6], v[7]);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float v[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float v[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float v[8]) {
  uint4 u;
  u.x = pack_bf16x2(v[0], v[1]);
  u.y = pack_bf16x2(v[2], v[3]);
  u.z = pack_bf16x2(v[4], v[5]);
  u.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
// Raw (unconverted) 8-element vectors: kernels issue all their global loads back to back into these, with clamped
// addresses instead of branches, and convert/mask afterwards -- a load inside a conditional block is waited for
// inside that block, which serialises HBM round trips (measured: 10x on the depthwise kernels).
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { uint4 u; };
template <> struct Raw8<float> { float4 a, b; };
template <typename T> __device__ __forceinline__ Raw8<T> ldraw(const T* p);
template <> __device__ __forceinline__ Raw8<bf16_t> ldraw<bf16_t>(const bf16_t* p) {
  Raw8<bf16_t> r; r.u = *reinterpret_cast<const uint4*>(p); return r;
}
template <> __device__ __forceinline__ Raw8<float> ldraw<float>(const float* p) {
  Raw8<float> r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r;
}
__device__ __forceinline__ void cvt8(const Raw8<bf16_t>& r, float v[8]) {
  spb_unpack2(r.u.x, v[0], v[1]); spb_unpack2(r.u.y, v[2], v[3]); spb_unpack2(r.u.z, v[4], v[5]); spb_unpack2(r.u.w, v[6], v[7]);
}
__device__ __forceinline__ void cvt8(const Raw8<float>& r, float v[8]) {
  v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}

template <typename T> __device__ __forceinline__ void rnd8(float v[8]);
template <> __device__ __forceinline__ void rnd8<float>(float v[8]) {}
template <> __device__ __forceinline__ void rnd8<bf16_t>(float v[8]) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const uint32_t p = pack_bf16x2(v[i], v[i + 1]);
    spb_unpack2(p, v[i], v[i + 1]);
  }
}

// ---------------------------------------------------------------------------------------------
// activations (codes are SPB_ACT_* in the public header)
// All four activations are one branch-free form:  act(u) = min(max(u,0),hi) + ns*min(u,0)
//   none: hi=inf ns=1   relu: hi=inf ns=0   relu6: hi=6 ns=0   leaky: hi=inf ns=slope
// (hi, ns) depend on kernel arguments only, so the compiler keeps them in SGPRs; a per-element switch on `act` cost
// 272 branches and 216 VGPRs in the depthwise kernel.
__device__ __forceinline__ float act_hi(int act) { return act == SPB_ACT_RELU6 ? 6.f : __builtin_inff(); }
__device__ __forceinline__ float act_ns(int act, float slope) {
  return act == SPB_ACT_NONE ? 1.f : (act == SPB_ACT_LEAKY ? slope : 0.f);
}
__device__ __forceinline__ float act_fwd(float u, int act, float slope) {
  return __builtin_amdgcn_fmed3f(u, 0.f, act_hi(act)) + act_ns(act, slope) * fminf(u, 0.f);
}
__device__ __forceinline__ float act_grad(float u, int act, float slope) {
  return u > 0.f ? (u < act_hi(act) ? 1.f : 0.f) : act_ns(act, slope);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm reference (spb_bnref_t in the public header).  A consumer derives the per-channel affine
// itself from the raw batch sums the producer's epilogue accumulated, so no "finalize" launch sits
// between a convolution and the next one.
typedef spb_bnref_t BNRef;

// The two per-channel sums of a tensor, added over its R <= SPB_MAX_REPLICAS replicas.  All 2 * SPB_MAX_REPLICAS loads are
// issued back to back on clamped replica indices and masked afterwards: a loop with the runtime bound R makes every load
// wait for the previous one (one memory round trip per replica -- measured: ~1 us per kernel per 14 loads), which is what
// kept the replica count of the small tensors at 1 and their atomics serialised on one address.
#define SPB_MAX_REPLICAS 8
__device__ __forceinline__ void bn_replica_sums(const float* base, int R, int C, int c, float& a, float& b) {
  if (R == 1) {   // (uniform) most tensors -- every map below 56x56 at bs=48 -- have one replica: 2 loads instead of 16 clamped ones
    a = base[c]; b = base[C + c];
    return;
  }
  float va[SPB_MAX_REPLICAS], vb[SPB_MAX_REPLICAS];
#pragma unroll
  for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
    const size_t o = (size_t)(i < R ? i : 0) * 2 * C + c;
    va[i] = base[o]; vb[i] = base[o + C];
  }
  a = 0.f; b = 0.f;
#pragma unroll
  for (int i = 0; i < SPB_MAX_REPLICAS; ++i) { a += i < R ? va[i] : 0.f; b += i < R ? vb[i] : 0.f; }
}

// mean / inverse std of channel c
__device__ __forceinline__ void bn_moments(const BNRef& r, int c, float& mean, float& invstd) {
  if (r.moments) {  // eval: sums holds [mean | var]
    mean = r.sums[c];
    invstd = rsqrtf(r.sums[r.C + c] + r.eps);
    return;
  }
  float s, q;
  bn_replica_sums(r.sums, r.R, r.C, c, s, q);
  mean = s * r.inv_n;
  float var = fmaxf(q * r.inv_n - mean * mean, 0.f);
  invstd = rsqrtf(var + r.eps);
}
// The coefficient helpers below request EVERYTHING a channel needs before the first value is used.  Every `if` on the way (eval moments,
// replica count, missing affine) is uniform, but a load behind a branch is only issued once the branch is decided and everything before
// its merge point has been waited for: gamma / beta after the sums cost a second memory round trip in the prologue of every launch, the
// BatchNorm-backward coefficients (sums, backward sums, gamma) a third -- ~0.7 us each, ~175 launches per step.
// forward affine: a = act(z*scale + shift)
__device__ __forceinline__ void bn_fwd_coef(const BNRef& r, int c, float& scale, float& shift) {
  if (r.gamma == nullptr) { scale = 1.f; shift = 0.f; return; }
  const float gm = r.gamma[c], bt = r.beta[c];      // in flight together with the sums
  float mean, is;
  bn_moments(r, c, mean, is);
  scale = gm * is;
  shift = bt - mean * scale;
}
// backward: dz = g*p0 + z*p1 + p2   (training-mode BN input gradient; g = dL/d(BN output))
__device__ __forceinline__ void bn_bwd_coef(const BNRef& r, int c, float& p0, float& p1, float& p2) {
  if (r.gamma == nullptr) { p0 = 1.f; p1 = 0.f; p2 = 0.f; return; }
  const float gm = r.gamma[c];
  float mean, is, s1, s2;
  if (r.R == 1 && !r.moments) {   // (uniform) the common case -- every map below 28x28 at bs=48: five loads, one round trip
    const float s = r.sums[c], q = r.sums[r.C + c];
    s1 = r.bsums[c]; s2 = r.bsums[r.C + c];
    mean = s * r.inv_n;
    is = rsqrtf(fmaxf(q * r.inv_n - mean * mean, 0.f) + r.eps);
  } else if (!r.moments) {        // replicas: the 4 x SPB_MAX_REPLICAS clamped loads of both pairs of sums first
    float va[SPB_MAX_REPLICAS], vb[SPB_MAX_REPLICAS], vc[SPB_MAX_REPLICAS], vd[SPB_MAX_REPLICAS];
#pragma unroll
    for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
      const size_t o = (size_t)(i < r.R ? i : 0) * 2 * r.C + c;
      va[i] = r.sums[o]; vb[i] = r.sums[o + r.C]; vc[i] = r.bsums[o]; vd[i] = r.bsums[o + r.C];
    }
    float s = 0.f, q = 0.f;
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
      s += i < r.R ? va[i] : 0.f; q += i < r.R ? vb[i] : 0.f; s1 += i < r.R ? vc[i] : 0.f; s2 += i < r.R ? vd[i] : 0.f;
    }
    mean = s * r.inv_n;
    is = rsqrtf(fmaxf(q * r.inv_n - mean * mean, 0.f) + r.eps);
  } else {
    bn_moments(r, c, mean, is);
    bn_replica_sums(r.bsums, r.R, r.C, c, s1, s2);
  }
  s1 *= r.inv_n; s2 *= r.inv_n;
  p0 = gm * is;
  p1 = -p0 * is * s2;
  p2 = p0 * (mean * is * s2 - s1);
}

// Split form for prologues that need several BatchNorms at once: bn_issue requests gamma, beta and the replica sums of a channel (clamped
// replica indices: no branch), bn_issue_bwd the backward sums, bn_finish turns them into mean / inverse std.  Training statistics only
// (r.moments == 0) and r.gamma != nullptr: the caller tests that once, uniformly, and otherwise uses bn_fwd_coef / bn_bwd_coef.
struct BNLoad { float gm, bt, a[SPB_MAX_REPLICAS], b[SPB_MAX_REPLICAS]; };
struct BNLoadB { float c[SPB_MAX_REPLICAS], d[SPB_MAX_REPLICAS]; };
__device__ __forceinline__ void bn_issue(const BNRef& r, int c, BNLoad& L) {
  L.gm = r.gamma[c]; L.bt = r.beta[c];
#pragma unroll
  for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
    const size_t o = (size_t)(i < r.R ? i : 0) * 2 * r.C + c;
    L.a[i] = r.sums[o]; L.b[i] = r.sums[o + r.C];
  }
}
__device__ __forceinline__ void bn_issue_bwd(const BNRef& r, int c, BNLoadB& L) {
#pragma unroll
  for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
    const size_t o = (size_t)(i < r.R ? i : 0) * 2 * r.C + c;
    L.c[i] = r.bsums[o]; L.d[i] = r.bsums[o + r.C];
  }
}
__device__ __forceinline__ void bn_finish(const BNRef& r, const BNLoad& L, float& mean, float& is) {
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < SPB_MAX_REPLICAS; ++i) { s += i < r.R ? L.a[i] : 0.f; q += i < r.R ? L.b[i] : 0.f; }
  mean = s * r.inv_n;
  is = rsqrtf(fmaxf(q * r.inv_n - mean * mean, 0.f) + r.eps);
}
__device__ __forceinline__ void bn_finish_bwd(const BNRef& r, const BNLoad& L, const BNLoadB& B, float& p0, float& p1, float& p2) {
  float mean, is, s1 = 0.f, s2 = 0.f;
  bn_finish(r, L, mean, is);
#pragma unroll
  for (int i = 0; i < SPB_MAX_REPLICAS; ++i) { s1 += i < r.R ? B.c[i] : 0.f; s2 += i < r.R ? B.d[i] : 0.f; }
  s1 *= r.inv_n; s2 *= r.inv_n;
  p0 = L.gm * is;
  p1 = -p0 * is * s2;
  p2 = p0 * (mean * is * s2 - s1);
}

// BatchNorm-backward coefficients of channel c of `r` and, if `has_epi`, the forward affine + moments of the same channel of `e` (the
// depthwise backward kernels need both: dz of their output's BatchNorm and the activation mask / xhat of their input's).  Common case
// (one replica each, training statistics): the nine loads of both are requested together.
__device__ __forceinline__ void bn_bwd_epi_coef(const BNRef& r, const BNRef& e, bool has_epi, int c, float& p0, float& p1, float& p2,
                                                float& sc, float& sh, float& mu, float& is) {
  sc = 1.f; sh = 0.f; mu = 0.f; is = 0.f;
  if (has_epi && r.gamma != nullptr && r.R == 1 && !r.moments && e.R == 1 && !e.moments) {   // (uniform)
    const float gm = r.gamma[c], s = r.sums[c], q = r.sums[r.C + c], s1 = r.bsums[c] * r.inv_n, s2 = r.bsums[r.C + c] * r.inv_n;
    const float eg = e.gamma[c], eb = e.beta[c], es = e.sums[c], eq = e.sums[e.C + c];
    const float mean = s * r.inv_n;
    const float ris = rsqrtf(fmaxf(q * r.inv_n - mean * mean, 0.f) + r.eps);
    p0 = gm * ris;
    p1 = -p0 * ris * s2;
    p2 = p0 * (mean * ris * s2 - s1);
    mu = es * e.inv_n;
    is = rsqrtf(fmaxf(eq * e.inv_n - mu * mu, 0.f) + e.eps);
    sc = eg * is;
    sh = eb - mu * sc;
    return;
  }
  bn_bwd_coef(r, c, p0, p1, p2);
  if (has_epi) {
    bn_moments(e, c, mu, is);
    sc = e.gamma[c] * is;
    sh = e.beta[c] - mu * sc;
  }
}

// Per-channel prologue table of a 256-thread workgroup: coef[0..Kp) = c0, coef[Kp..2Kp) = c1, coef[2Kp..3Kp) = c2 for
// MODE 1 (forward: scale, shift, 0) or MODE 2 (BN backward: p0, p1, p2); channels >= K get zeros.  A thread owns channels
// t, t+256, ...; it derives G of them per pass from clamped, branch-free loads, so a 960-channel table costs two memory
// round trips instead of four (a `for (c = t; c < K; c += 256)` loop waits for every channel's loads before the next
// channel's are issued: measured ~1.5 us per iteration on the 7x7 layers).
template <int MODE>
__device__ __forceinline__ void bn_coef_table(const BNRef& r, int K, int Kp, float* coef, int t) {
  // Common case (affine present, one replica, training statistics): the uniform tests are taken ONCE, outside the channel loop, and the 4-5
  // loads of up to GF channels per thread are requested together -- with the tests inside bn_fwd_coef / bn_bwd_coef every channel of the
  // unrolled loop is its own basic-block chain and its loads are waited for before the next channel's are issued (three memory round trips
  // per pass of a 768-channel table, measured in the prologue of the 7x7 GEMMs).
  if (r.gamma != nullptr && r.R == 1 && !r.moments) {   // (uniform)
    constexpr int GF = 5;                               // 1280 channels in one pass
    for (int cb = t; cb < Kp; cb += 256 * GF) {
      float gm[GF], bt[GF], s[GF], q[GF], s1[GF], s2[GF];
#pragma unroll
      for (int j = 0; j < GF; ++j) {
        const int c = cb + 256 * j;
        const int cc = c < K ? c : K - 1;
        gm[j] = r.gamma[cc]; s[j] = r.sums[cc]; q[j] = r.sums[r.C + cc];
        if (MODE == 1) { bt[j] = r.beta[cc]; s1[j] = 0.f; s2[j] = 0.f; }
        else { bt[j] = 0.f; s1[j] = r.bsums[cc]; s2[j] = r.bsums[r.C + cc]; }
      }
#pragma unroll
      for (int j = 0; j < GF; ++j) {
        const int c = cb + 256 * j;
        if (c < Kp) {
          const bool ok = c < K;
          const float mean = s[j] * r.inv_n;
          const float is = rsqrtf(fmaxf(q[j] * r.inv_n - mean * mean, 0.f) + r.eps);
          float v0, v1, v2;
          if (MODE == 1) { v0 = gm[j] * is; v1 = bt[j] - mean * v0; v2 = 0.f; }
          else {
            const float m1 = s1[j] * r.inv_n, m2 = s2[j] * r.inv_n;
            v0 = gm[j] * is; v1 = -v0 * is * m2; v2 = v0 * (mean * is * m2 - m1);
          }
          coef[c] = ok ? v0 : 0.f; coef[Kp + c] = ok ? v1 : 0.f; coef[2 * Kp + c] = ok ? v2 : 0.f;
        }
      }
    }
    return;
  }
  constexpr int G = MODE == 1 ? 3 : 2;
  if (r.gamma != nullptr && !r.moments) {               // (uniform) replicas (the 14x14 maps and larger): the same, 2-4 x 8 clamped loads per channel
    for (int cb = t; cb < Kp; cb += 256 * G) {
      float gm[G], bt[G], va[G][SPB_MAX_REPLICAS], vb[G][SPB_MAX_REPLICAS], vc[MODE == 1 ? 1 : G][SPB_MAX_REPLICAS], vd[MODE == 1 ? 1 : G][SPB_MAX_REPLICAS];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int c = cb + 256 * j;
        const int cc = c < K ? c : K - 1;
        gm[j] = r.gamma[cc];
        bt[j] = MODE == 1 ? r.beta[cc] : 0.f;
#pragma unroll
        for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
          const size_t o = (size_t)(i < r.R ? i : 0) * 2 * r.C + cc;
          va[j][i] = r.sums[o]; vb[j][i] = r.sums[o + r.C];
          if (MODE != 1) { vc[MODE == 1 ? 0 : j][i] = r.bsums[o]; vd[MODE == 1 ? 0 : j][i] = r.bsums[o + r.C]; }
        }
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int c = cb + 256 * j;
        float sv = 0.f, qv = 0.f, m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
          sv += i < r.R ? va[j][i] : 0.f; qv += i < r.R ? vb[j][i] : 0.f;
          if (MODE != 1) { m1 += i < r.R ? vc[MODE == 1 ? 0 : j][i] : 0.f; m2 += i < r.R ? vd[MODE == 1 ? 0 : j][i] : 0.f; }
        }
        if (c < Kp) {
          const bool ok = c < K;
          const float mean = sv * r.inv_n;
          const float is = rsqrtf(fmaxf(qv * r.inv_n - mean * mean, 0.f) + r.eps);
          float v0, v1, v2;
          if (MODE == 1) { v0 = gm[j] * is; v1 = bt[j] - mean * v0; v2 = 0.f; }
          else {
            m1 *= r.inv_n; m2 *= r.inv_n;
            v0 = gm[j] * is; v1 = -v0 * is * m2; v2 = v0 * (mean * is * m2 - m1);
          }
          coef[c] = ok ? v0 : 0.f; coef[Kp + c] = ok ? v1 : 0.f; coef[2 * Kp + c] = ok ? v2 : 0.f;
        }
      }
    }
    return;
  }
  for (int cb = t; cb < Kp; cb += 256 * G) {
    float c0[G], c1[G], c2[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = cb + 256 * j;
      const int cc = c < K ? c : K - 1;
      c0[j] = 0.f; c1[j] = 0.f; c2[j] = 0.f;
      if (MODE == 1) bn_fwd_coef(r, cc, c0[j], c1[j]);
      else bn_bwd_coef(r, cc, c0[j], c1[j], c2[j]);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = cb + 256 * j;
      if (c < Kp) {
        const bool ok = c < K;
        coef[c] = ok ? c0[j] : 0.f; coef[Kp + c] = ok ? c1[j] : 0.f; coef[2 * Kp + c] = ok ? c2[j] : 0.f;
      }
    }
  }
}

// scale / shift of every channel into two arrays (LDS), G channels of a thread per pass with all of their loads in flight before
// anything is stored: `for (c = t; c < C; c += 256) bn_fwd_coef(...)` makes one memory round trip per iteration (four for the 1024
// channels of the head and of the concat's bn_apply: ~1.5 us each on the launch stream).
template <int G>
__device__ __forceinline__ void bn_fwd_table(const BNRef& r, int C, float* sc, float* sh, int t, int nthr) {
  if (r.gamma != nullptr && r.R == 1 && !r.moments) {   // (uniform) as in bn_coef_table: the tests once, the loads of G channels together
    for (int cb = t; cb < C; cb += nthr * G) {
      float gm[G], bt[G], s[G], q[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int c = cb + nthr * j, cc = c < C ? c : C - 1;
        gm[j] = r.gamma[cc]; bt[j] = r.beta[cc]; s[j] = r.sums[cc]; q[j] = r.sums[r.C + cc];
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int c = cb + nthr * j;
        const float mean = s[j] * r.inv_n;
        const float a = gm[j] * rsqrtf(fmaxf(q[j] * r.inv_n - mean * mean, 0.f) + r.eps);
        if (c < C) { sc[c] = a; sh[c] = bt[j] - mean * a; }
      }
    }
    return;
  }
  for (int cb = t; cb < C; cb += nthr * G) {
    float a[G], b[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = cb + nthr * j;
      bn_fwd_coef(r, c < C ? c : C - 1, a[j], b[j]);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = cb + nthr * j;
      if (c < C) { sc[c] = a[j]; sh[c] = b[j]; }
    }
  }
}

// Output-side coefficients of the input-gradient epilogue (activation mask / xhat of the W columns n0 .. n0+W-1 of `epi`): ecoef[0..W) scale,
// [ld..ld+W) shift and, with FULL, [2ld..) mean, [3ld..) inverse std.  Split in two so that a prologue can request these sums BEFORE it
// builds its reduction-side table (bn_coef_table) and finish them after: everything rides one memory round trip (as two plain calls the
// second one's loads were issued only after the first table's had been waited for).  W <= 512, 256 threads.
struct BNEpiPre { BNLoad l[2]; bool fast; };
__device__ __forceinline__ void bn_epi_issue(const BNRef& epi, int n0, int N, int W, int t, BNEpiPre& P) {
  P.fast = epi.gamma != nullptr && !epi.moments && W <= 512;   // (uniform)
  if (P.fast) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = n0 + t + 256 * u;
      bn_issue(epi, n < N ? n : N - 1, P.l[u]);
    }
  }
}
template <bool FULL>
__device__ __forceinline__ void bn_epi_finish(const BNRef& epi, int n0, int N, int W, int ld, float* ecoef, int t, const BNEpiPre& P) {
  if (P.fast) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = t + 256 * u;
      float mu, is;
      bn_finish(epi, P.l[u], mu, is);
      if (c < W) {
        const bool ok = n0 + c < N;
        const float sc = P.l[u].gm * is;
        ecoef[c] = ok ? sc : 1.f; ecoef[ld + c] = ok ? P.l[u].bt - mu * sc : 0.f;
        if (FULL) { ecoef[2 * ld + c] = ok ? mu : 0.f; ecoef[3 * ld + c] = ok ? is : 0.f; }
      }
    }
    return;
  }
  for (int c = t; c < W; c += 256) {
    float sc = 1.f, sh = 0.f, mu = 0.f, is = 0.f;
    if (n0 + c < N && epi.gamma != nullptr) {
      bn_moments(epi, n0 + c, mu, is);
      sc = epi.gamma[n0 + c] * is;
      sh = epi.beta[n0 + c] - mu * sc;
    }
    ecoef[c] = sc; ecoef[ld + c] = sh;
    if (FULL) { ecoef[2 * ld + c] = mu; ecoef[3 * ld + c] = is; }
  }
}

// Prologue table of the residual join a = bn(A) + bn2(A2) (spb_gemm_args_t pro_mode 3; neither side has an activation):
// coef[0..Kp) = scale of A, coef[Kp..2Kp) = scale of A2, coef[2Kp..3Kp) = shift + shift2 -- the a*c0 + a2*c1 + c2 form the
// BatchNorm-backward prologue already runs on.  Same launch shape as bn_coef_table (256 threads, clamped branch-free loads).
__device__ __forceinline__ void bn_join_table(const BNRef& r, const BNRef& r2, int K, int Kp, float* coef, int t) {
  constexpr int G = 2;
  if (r.gamma != nullptr && !r.moments && r2.gamma != nullptr && !r2.moments) {   // (uniform) the sums of both sides requested together
    for (int cb = t; cb < Kp; cb += 256 * G) {
      BNLoad la[G], lb[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int c = cb + 256 * j;
        const int cc = c < K ? c : K - 1;
        bn_issue(r, cc, la[j]); bn_issue(r2, cc, lb[j]);
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int c = cb + 256 * j;
        float m1, i1, m2, i2;
        bn_finish(r, la[j], m1, i1); bn_finish(r2, lb[j], m2, i2);
        if (c < Kp) {
          const bool ok = c < K;
          const float sc = la[j].gm * i1, sc2 = lb[j].gm * i2;
          coef[c] = ok ? sc : 0.f; coef[Kp + c] = ok ? sc2 : 0.f;
          coef[2 * Kp + c] = ok ? (la[j].bt - m1 * sc) + (lb[j].bt - m2 * sc2) : 0.f;
        }
      }
    }
    return;
  }
  for (int cb = t; cb < Kp; cb += 256 * G) {
    float c0[G], c1[G], c2[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = cb + 256 * j;
      const int cc = c < K ? c : K - 1;
      float sc, sh, sc2, sh2;
      bn_fwd_coef(r, cc, sc, sh);
      bn_fwd_coef(r2, cc, sc2, sh2);
      c0[j] = sc; c1[j] = sc2; c2[j] = sh + sh2;
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int c = cb + 256 * j;
      if (c < Kp) {
        const bool ok = c < K;
        coef[c] = ok ? c0[j] : 0.f; coef[Kp + c] = ok ? c1[j] : 0.f; coef[2 * Kp + c] = ok ? c2[j] : 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wave64 reductions via DPP-lowered shuffles
// sum over the 16 lanes of a DPP row, result in every lane: VALU-only butterflies (quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror).  __shfl_xor compiles to ds_bpermute_b32 (LDS crossbar), several times slower.
template <int CTRL> __device__ __forceinline__ float spb_dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += spb_dpp_f<0xB1>(v); v += spb_dpp_f<0x4E>(v); v += spb_dpp_f<0x141>(v); v += spb_dpp_f<0x140>(v);
  return v;
}
// v[lane] + v[lane ^ 32] and v[lane] + v[lane ^ 16] in every lane, on the vector ALU: gfx950's v_permlane32_swap / v_permlane16_swap
// exchange the upper half (the odd 16-lane rows) of the first operand with the lower half (the even rows) of the second.
// Inline asm on purpose: __builtin_amdgcn_permlane32_swap / 16_swap are miscompiled by this hipcc when both operands are the same
// value (the second result register is dropped: scratch/probe_permlane.hip, checked on the GPU), and __shfl_xor is a
// ds_bpermute_b32 -- an LDS round trip per step that serialises reductions inside loops (a 9-tap loop with 16 of them per tap:
// 1.2 us per tap, measured in the depthwise plane kernel).
__device__ __forceinline__ float xor32_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor16_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware remap of a linear workgroup id: the dispatcher places block b on XCD b%8, so give
// each XCD a contiguous chunk of the logical id space (neighbouring tiles then share one L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;      // XCD x runs blocks x, x+8, ...: q of them, one more if x < r
  return x * q + (x < r ? x : r) + (bid >> 3);
}

static inline int spb_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// LDS-DMA (global_load_lds_dwordx4): lane i's 16 bytes land at lds_base + 16*i with no VGPR in between.
// Inline asm on purpose: with the builtin hipcc knows the instruction writes LDS and drains vmcnt(0) before the next
// ds_read, which serialises every ring built on it; hidden in asm, a counted s_waitcnt (wait_vmcnt) is the only
// ordering (and it is required: nothing else orders a ds_read behind a pending DMA).  M0 is compiler-reserved: saved
// and restored.  lds_base must be wave-uniform.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
// same with a wave-uniform 64-bit base in SGPRs and a 32-bit byte offset per lane: a ring whose source advances by a constant per stage
// then needs no vector arithmetic at all for its addresses (the base moves on the scalar unit)
__device__ __forceinline__ void dma16s(const void* sbase, unsigned voff, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: every load prefetched across it and
// every store issued before it costs its full round trip at the barrier.  For workgroups whose threads talk through LDS alone
// (global data only ever reaches LDS through registers, i.e. behind its own data dependency).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
