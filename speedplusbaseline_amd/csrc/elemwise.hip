// Streaming (HBM-bound) passes around the convolutions:
//   * bn_apply / bn_bwd_prep : materialise act(bn(z)) (+ residual, concat, RouterV2 reorg) and its backward
//       (MobileNetV2 residual adds; reference park2019.py:74-80 reorg + cat)
//   * table-driven whole-network maintenance: BN running statistics, BN parameter gradients, compute-dtype weight
//       copies (bf16 W, W^T, permuted head weight)
//   * global-norm clip + optimiser update fused over the flat parameter arena
//       (reference trainer.py:90,97  clip_grad_norm_; build.py:60-78 sgd/rmsprop/adam/adamw)
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include "common.h"
#include "optim_math.h"

namespace {

// ------------------------------------------------------------------------------------------------ bn_apply
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const spb_bnapply_args_t a) {
  extern __shared__ float cf[];  // [4][C]: scale, shift, res scale, res shift
  const int C = a.C, CG = C >> 3;
  bn_fwd_table<4>(a.bn, C, cf, cf + C, threadIdx.x, 256);
  if (a.res) bn_fwd_table<4>(a.bn_res, C, cf + 2 * C, cf + 3 * C, threadIdx.x, 256);
  __syncthreads();
  const T* Z = reinterpret_cast<const T*>(a.Z);
  const T* R = reinterpret_cast<const T*>(a.res);
  T* Y = reinterpret_cast<T*>(a.Y);
  const long long items = (long long)a.B * a.H * a.W * CG;
  const int s = a.reorg;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long long)gridDim.x * 256) {
    const int cg = (int)(it % CG);
    const long long p = it / CG;
    const int c0 = cg * 8;
    float v[8];
    ld8<T>(Z + (size_t)p * C + c0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_fwd(v[j] * cf[c0 + j] + cf[C + c0 + j], a.bn.act, a.bn.slope);
    if (R) {
      float r[8];
      ld8<T>(R + (size_t)p * C + c0, r);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] += act_fwd(r[j] * cf[2 * C + c0 + j] + cf[3 * C + c0 + j], a.bn_res.act, a.bn_res.slope);
    }
    size_t o;
    if (s > 0) {
      const int w = (int)(p % a.W), h = (int)((p / a.W) % a.H), b = (int)(p / ((long long)a.W * a.H));
      const int OH = a.H / s, OW = a.W / s;
      o = ((size_t)(b * OH + h / s) * OW + w / s) * a.ldc + a.coff + ((h % s) * s + (w % s)) * C + c0;
    } else {
      o = (size_t)p * a.ldc + a.coff + c0;
    }
    st8<T>(Y + o, v);
  }
}

// ------------------------------------------------------------------------------------------------ bn_bwd_prep
// thread keeps a fixed channel group (grid size is a multiple of CG/gcd(CG,256)) so sums stay in registers
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_prep_kernel(const spb_bnbwd_args_t a) {
  extern __shared__ float cf[];  // [4][C] scale, shift, mean, invstd ; then [2][C] reduction
  const int C = a.C, CG = C >> 3;
  float* red = cf + 4 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float mu = 0.f, is = 0.f, sc = 1.f, sh = 0.f;
    if (a.bn.gamma) {
      bn_moments(a.bn, c, mu, is);
      sc = a.bn.gamma[c] * is;
      sh = a.bn.beta[c] - mu * sc;
    }
    cf[c] = sc; cf[C + c] = sh; cf[2 * C + c] = mu; cf[3 * C + c] = is;
    red[c] = 0.f; red[C + c] = 0.f;
  }
  __syncthreads();
  const T* dY = reinterpret_cast<const T*>(a.dY);
  const T* Z = reinterpret_cast<const T*>(a.Z);
  T* G = reinterpret_cast<T*>(a.G);
  const long long items = (long long)a.B * a.H * a.W * CG;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(gid % CG), c0 = cg * 8;
  const int s = a.reorg;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  for (long long it = gid; it < items; it += (long long)gridDim.x * 256) {
    const long long p = it / CG;
    size_t o;
    if (s > 0) {
      const int w = (int)(p % a.W), h = (int)((p / a.W) % a.H), b = (int)(p / ((long long)a.W * a.H));
      const int OH = a.H / s, OW = a.W / s;
      o = ((size_t)(b * OH + h / s) * OW + w / s) * a.ldc + a.coff + ((h % s) * s + (w % s)) * C + c0;
    } else {
      o = (size_t)p * a.ldc + a.coff + c0;
    }
    float d[8], z[8];
    ld8<T>(dY + o, d);
    ld8<T>(Z + (size_t)p * C + c0, z);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = z[j] * cf[c0 + j] + cf[C + c0 + j];
      d[j] = rnd<T>(d[j] * act_grad(u, a.bn.act, a.bn.slope));
      s1[j] += d[j];
      s2[j] += d[j] * ((z[j] - cf[2 * C + c0 + j]) * cf[3 * C + c0 + j]);
    }
    st8<T>(G + (size_t)p * C + c0, d);
  }
#ifdef SPB_DET   // reproducible twin (common.h): no LDS float atomics (their order is not fixed); every thread adds exactly
  {
    const int rep = blockIdx.x % a.oR;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(a.osums + (size_t)rep * 2 * C + c0 + j, s1[j]);
      atomicAdd(a.osums + (size_t)rep * 2 * C + C + c0 + j, s2[j]);
    }
  }
#else
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    atomicAdd(&red[c0 + j], s1[j]);
    atomicAdd(&red[C + c0 + j], s2[j]);
  }
  __syncthreads();
  const int rep = blockIdx.x % a.oR;
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(a.osums + (size_t)rep * 2 * C + i, red[i]);
#endif
}

// Row-parallel form (round 3): grid (ceil(C/64), row ranges of BBP_ROWS).  A workgroup owns 64 channels (8 lanes x 8-channel
// 16-byte vectors) x 32 row lanes with BBP_U rows of each lane in flight; every global load of a thread -- operands and the
// BatchNorm sums / affine of its 8 channels -- is issued before anything is waited for: one memory round trip on a few hundred
// workgroups.  The per-channel sums meet in LDS by plain stores (32 row lanes), then one f32 atomic per channel, sum and workgroup.
// (The kernel above walks the tensor with 147 / 37 workgroups, reduces through 16 LDS float atomics per thread and ends with 2*C
// global atomics per workgroup: 300 k of them on the 1024-channel concat branch, 22 + 19 us for two launches that move 18 MB.)
constexpr int BBP_U = 4, BBP_ROWS = 32 * BBP_U;
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_prep_rows_kernel(const spb_bnbwd_args_t a) {
  __shared__ float red[2][32][65];
  const int t = threadIdx.x, v = t & 7, sub = t >> 3;
  const int C = a.C, cb = blockIdx.x * 64, c0 = cb + v * 8;
  const bool cok = c0 < C;
  const int cc = cok ? c0 : 0;
  const long long rows = (long long)a.B * a.H * a.W;
  const long long r0 = (long long)blockIdx.y * BBP_ROWS, r1 = r0 + BBP_ROWS < rows ? r0 + BBP_ROWS : rows;
  const T* dY = reinterpret_cast<const T*>(a.dY);
  const T* Z = reinterpret_cast<const T*>(a.Z);
  T* G = reinterpret_cast<T*>(a.G);
  const int s = a.reorg;
  Raw8<T> dr[BBP_U], zr[BBP_U];
#pragma unroll
  for (int u = 0; u < BBP_U; ++u) {
    const long long ru = r0 + sub + 32 * u;
    const long long p = ru < r1 ? ru : r1 - 1;
    size_t o;
    if (s > 0) {
      const int w = (int)(p % a.W), h = (int)((p / a.W) % a.H), b = (int)(p / ((long long)a.W * a.H));
      const int OH = a.H / s, OW = a.W / s;
      o = ((size_t)(b * OH + h / s) * OW + w / s) * a.ldc + a.coff + ((h % s) * s + (w % s)) * C + cc;
    } else {
      o = (size_t)p * a.ldc + a.coff + cc;
    }
    dr[u] = ldraw<T>(dY + o);
    zr[u] = ldraw<T>(Z + (size_t)p * C + cc);
  }
  // BatchNorm coefficients of the workgroup's 64 channels: ONE channel per thread, so all of their loads are one memory round trip
  // (eight bn_moments calls per thread, each behind its uniform branches, were eight: the 7x7 launches took 13-18 us for 4-14 MB)
  float* tab = &red[0][0][0];      // [4][64], before `red` is needed
  if (t < 64) {
    const int c = cb + t < C ? cb + t : C - 1;
    float m = 0.f, iv = 0.f, g1 = 1.f, h1 = 0.f;
    if (a.bn.gamma) {   // (uniform)
      const float gm = a.bn.gamma[c], bt = a.bn.beta[c];      // requested with the sums, not after them
      bn_moments(a.bn, c, m, iv);
      g1 = gm * iv;
      h1 = bt - m * g1;
    }
    tab[t] = g1; tab[64 + t] = h1; tab[128 + t] = m; tab[192 + t] = iv;
  }
  __syncthreads();
  float sc[8], sh[8], mu[8], is[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = tab[v * 8 + j]; sh[j] = tab[64 + v * 8 + j]; mu[j] = tab[128 + v * 8 + j]; is[j] = tab[192 + v * 8 + j]; }
  __syncthreads();                 // `red` is written below
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
  for (int u = 0; u < BBP_U; ++u) {
    const long long ru = r0 + sub + 32 * u;
    if (ru < r1 && cok) {
      float d[8], z[8];
      cvt8(dr[u], d); cvt8(zr[u], z);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float uu = z[j] * sc[j] + sh[j];
        d[j] = rnd<T>(d[j] * act_grad(uu, a.bn.act, a.bn.slope));
        s1[j] += d[j];
        s2[j] += d[j] * ((z[j] - mu[j]) * is[j]);
      }
      st8<T>(G + (size_t)ru * C + c0, d);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][sub][v * 8 + j] = s1[j]; red[1][sub][v * 8 + j] = s2[j]; }
  __syncthreads();
  if (t < 128) {
    const int which = t >> 6, c = t & 63;
    if (cb + c < C) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += red[which][i][c];
      atomicAdd(a.osums + (size_t)(blockIdx.y % a.oR) * 2 * C + (size_t)which * C + cb + c, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------ BN tables
__global__ __launch_bounds__(256) void bn_running_update_kernel(const spb_bnupd_entry_t* tab, const float* stats,
                                                                float* buffers, long long* nbt, float momentum) {
  const spb_bnupd_entry_t e = tab[blockIdx.x];
  for (int c = threadIdx.x; c < e.C; c += 256) {
    float s, q;
    bn_replica_sums(stats + e.sums_off, e.R, e.C, c, s, q);     // all replicas in one memory round trip
    const float mean = s * e.inv_n;
    const float var = fmaxf(q * e.inv_n - mean * mean, 0.f);
    float* rm = buffers + e.rm_off;
    rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
    rm[e.C + c] = (1.f - momentum) * rm[e.C + c] + momentum * var * e.unbias;
  }
  if (threadIdx.x == 0 && nbt) nbt[e.bn_index] += 1;
}

__global__ __launch_bounds__(256) void bn_param_grads_kernel(const spb_bnupd_entry_t* tab, const float* stats,
                                                             float* grads) {
  const spb_bnupd_entry_t e = tab[blockIdx.x];
  if (e.bsums_off < 0) return;
  for (int c = threadIdx.x; c < e.C; c += 256) {
    float s1, s2;
    bn_replica_sums(stats + e.bsums_off, e.R, e.C, c, s1, s2);
    grads[e.gamma_off + c] += s2;
    grads[e.beta_off + c] += s1;
  }
}

// the same, and afterwards the block zeroes the batch-sum slots of its own BatchNorm (sums and backward sums, all replicas: one
// contiguous run of 4 R C floats from sums_off): nothing reads them after this kernel, and the next training forward then starts on
// clean accumulators without a memset launch at the head of its launch stream (5.7 us per step)
__global__ __launch_bounds__(256) void bn_param_grads_zero_kernel(const spb_bnupd_entry_t* tab, float* stats, float* grads) {
  const spb_bnupd_entry_t e = tab[blockIdx.x];
  if (e.bsums_off >= 0) {
    // the old gradient values and every replica sum of the pass are requested together (the replica-count test is taken once, outside:
    // inside bn_replica_sums each of the four channels waited for its own loads, then the read-modify-write made a further round trip --
    // this launch sits between the last weight gradient and the optimizer, with nothing beside it)
    const float* bs = stats + e.bsums_off;
    for (int cb = threadIdx.x; cb < e.C; cb += 1024) {
      float s1[4], s2[4], og[4], ob[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cb + 256 * j, cc = c < e.C ? c : e.C - 1;
        og[j] = grads[e.gamma_off + cc]; ob[j] = grads[e.beta_off + cc];
      }
      if (e.R == 1) {   // (uniform)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = cb + 256 * j, cc = c < e.C ? c : e.C - 1; s1[j] = bs[cc]; s2[j] = bs[e.C + cc]; }
      } else {
        float va[4][SPB_MAX_REPLICAS], vb[4][SPB_MAX_REPLICAS];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = cb + 256 * j, cc = c < e.C ? c : e.C - 1;
#pragma unroll
          for (int i = 0; i < SPB_MAX_REPLICAS; ++i) {
            const size_t o = (size_t)(i < e.R ? i : 0) * 2 * e.C + cc;
            va[j][i] = bs[o]; vb[j][i] = bs[o + e.C];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
          for (int i = 0; i < SPB_MAX_REPLICAS; ++i) { s1[j] += i < e.R ? va[j][i] : 0.f; s2[j] += i < e.R ? vb[j][i] : 0.f; }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = cb + 256 * j;
        if (c < e.C) { grads[e.gamma_off + c] = og[j] + s2[j]; grads[e.beta_off + c] = ob[j] + s1[j]; }
      }
    }
  }
  __syncthreads();
  const long long n = (e.bsums_off >= 0 ? 4LL : 2LL) * e.R * e.C;
  for (long long i = threadIdx.x; i < n; i += 256) stats[e.sums_off + i] = 0.f;
}

__global__ __launch_bounds__(256) void bn_load_running_kernel(const spb_bnupd_entry_t* tab, float* stats,
                                                              const float* buffers) {
  const spb_bnupd_entry_t e = tab[blockIdx.x];
  for (int c = threadIdx.x; c < 2 * e.C; c += 256) stats[e.sums_off + c] = buffers[e.rm_off + c];
}

// ------------------------------------------------------------------------------------------------ weight prep
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const spb_prep_entry_t* tab, int n_entries,
                                                          const float* params, T* wc) {
  __shared__ float tile[32][33];
  // entry owning this tile: last i with tab[i].tile0 <= blockIdx.x (tile0 ascending).  Binary search: the linear scan this
  // replaces cost every one of the ~10 000 blocks ~110 serial L2 loads (60 us for 45 MB of copies)
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= tab[mid].tile0) lo = mid; else hi = mid - 1;
  }
  const spb_prep_entry_t e = tab[lo];
  int tl = blockIdx.x - e.tile0;
  int rows = e.rows, cols = e.cols;
  const float* src = params + e.src_off;
  T* dst = wc + e.dst_off;
  size_t dbase = 0;
  if (e.mode == 2) {  // per-j [C][HW] -> [HW][C]
    const int tx_n = (cols + 31) / 32, ty_n = (rows + 31) / 32;
    const int j = tl / (tx_n * ty_n);
    tl -= j * tx_n * ty_n;
    src += (size_t)j * rows * cols;
    dbase = (size_t)j * rows * cols;
  }
  const int tx_n = (cols + 31) / 32;
  const int r0 = (tl / tx_n) * 32, c0 = (tl % tx_n) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  if (e.mode == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + ty + 8 * i, c = c0 + tx;
      if (r < rows && c < cols) dst[(size_t)r * cols + c] = from_f<T>(tile[ty + 8 * i][tx]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;  // destination row = source column
      if (r < rows && c < cols) dst[dbase + (size_t)c * rows + r] = from_f<T>(tile[tx][ty + 8 * i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ optimiser
// Deterministic: every block writes its partial sum, the last block to arrive adds them in block order.  (An atomicAdd per
// block made the clip coefficient differ in the last bit between runs -- and between the ranks of a data-parallel job, whose
// replicas then drift apart by an ulp per step.)
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, long long n, float* out, float* partial,
                                                          unsigned* counter) {
  __shared__ float red[4];
  __shared__ bool last;
  // block b owns the contiguous run of float4 [b*per, (b+1)*per): independent 16-byte loads, 4 in flight per thread
  const long long n4 = n >> 2;
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long long i = lo + threadIdx.x;
  for (; i + 768 < hi; i += 1024) {
    const float4 a = g4[i], b = g4[i + 256], c = g4[i + 512], d = g4[i + 768];
    s0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
    s2 += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
    s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
  }
  for (; i < hi; i += 256) { const float4 a = g4[i]; s0 += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
  float s = (s0 + s1) + (s2 + s3);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    __threadfence();
    last = atomicAdd(counter, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (unsigned j = threadIdx.x; j < gridDim.x; j += 256) t += __builtin_nontemporal_load(partial + j);   // fixed order per lane
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) { *out = red[0] + red[1] + red[2] + red[3]; *counter = 0; }
}

// SPB_SQ_PARTS workgroups, block b owns a contiguous run; 8 independent 16-byte loads per thread in flight; partial[b] by a
// plain store (the consumer adds the partials up)
__global__ __launch_bounds__(256) void grad_sqnorm_partials_kernel(const float* __restrict__ g, long long n, float* partial) {
  __shared__ float red[4];
  const long long n4 = n >> 2;
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 2048) {
    float4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long long j = i + 256 * u; a[u] = g4[j < hi ? j : hi - 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i + 256 * u < hi) s[u] += a[u].x * a[u].x + a[u].y * a[u].y + a[u].z * a[u].z + a[u].w * a[u].w;
  }
  float t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; t += v * v; }
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// VEC = 4: 16-byte accesses (n % 4 == 0, 16-byte aligned arenas); VEC = 1: scalar.  NT: non-temporal accesses (a 150 M
// element arena is a pure stream: nothing is reused from L2 / MALL)
template <bool NT> __device__ __forceinline__ f32x4_t ldv(const float* p, long long i) {
  const f32x4_t* q = reinterpret_cast<const f32x4_t*>(p) + i;
  if constexpr (NT) return __builtin_nontemporal_load(q); else return *q;
}
template <bool NT> __device__ __forceinline__ void stv(float* p, long long i, f32x4_t v) {
  f32x4_t* q = reinterpret_cast<f32x4_t*>(p) + i;
  if constexpr (NT) __builtin_nontemporal_store(v, q); else *q = v;
}
// Work assignment: block b owns the contiguous run of 256*U vectors starting at b*256*U (no grid-stride loop: blocks are
// dispatched in order, so at any moment the chip works on one compact window of each of the seven streams, which keeps
// HBM pages open; measured 5.9 TB/s against 5.0 TB/s for a 2048-block grid-stride loop on the 152 M element SPN arena).
template <int VEC, bool NT, int U>
__global__ __launch_bounds__(256) void optim_step_kernel(const spb_optim_args_t a) {
  if (a.skip && *a.skip != 0.f) return;   // AMP: a non-finite gradient was found -- GradScaler.step() skips optimizer.step()
  float gm = a.gmul ? *a.gmul : 1.f;
  float coef = 1.f;
  if (a.max_norm > 0.f && a.sq_partials && a.n_sq_partials > 0) {   // every workgroup adds the partials up in the same fixed order
    __shared__ float sqred[4];
    float t = (int)threadIdx.x < a.n_sq_partials ? a.sq_partials[threadIdx.x] : 0.f;
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) sqred[threadIdx.x >> 6] = t;
    __syncthreads();
    const float tn = sqrtf((sqred[0] + sqred[1]) + (sqred[2] + sqred[3])) * fabsf(gm);
    coef = fminf(a.max_norm / (tn + 1e-6f), 1.f);
  } else if (a.max_norm > 0.f && a.sqnorm) {
    const float tn = sqrtf(*a.sqnorm) * fabsf(gm);
    coef = fminf(a.max_norm / (tn + 1e-6f), 1.f);
  }
  const float gs = gm * coef;
  const float lr = a.hyper ? a.hyper[0] : a.lr;
  const float bias_c1 = a.hyper ? a.hyper[1] : a.bias_c1;
  const float bias_c2 = a.hyper ? a.hyper[2] : a.bias_c2;
  const bool has_m = a.m && (a.kind >= 2 || (a.kind == 0 && a.beta1 != 0.f));
  const bool has_v = a.v && a.kind >= 1;
  bf16_t* sh = reinterpret_cast<bf16_t*>(a.shadow_bf16);
  const long long nv = a.n / VEC;
  // one pass unless the launch was capped (max_blocks): then every block walks the arena with a grid stride
  for (long long run = blockIdx.x; run * (256 * U) < nv; run += gridDim.x) {
  const long long base = run * (256 * U) + threadIdx.x;
  if constexpr (VEC == 4) {
    f32x4_t p[U], g[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = base + u * 256;
      if (i < nv) {
        p[u] = ldv<NT>(a.params, i);
        g[u] = ldv<NT>(a.grads, i);
        m[u] = has_m ? ldv<NT>(a.m, i) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        v[u] = has_v ? ldv<NT>(a.v, i) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = base + u * 256;
      if (i < nv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float me = m[u][e], ve = v[u][e];
          p[u][e] = optim_one(a, gs, lr, bias_c1, bias_c2, p[u][e], g[u][e], me, ve);
          m[u][e] = me; v[u][e] = ve;
        }
        if (has_m) stv<NT>(a.m, i, m[u]);
        if (has_v) stv<NT>(a.v, i, v[u]);
        stv<NT>(a.params, i, p[u]);
        if (sh) reinterpret_cast<uint2*>(sh)[i] = make_uint2(pack_bf16x2(p[u][0], p[u][1]), pack_bf16x2(p[u][2], p[u][3]));
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long i = base + u * 256;
      if (i < nv) {
        float m = has_m ? a.m[i] : 0.f, v = has_v ? a.v[i] : 0.f;
        const float p = optim_one(a, gs, lr, bias_c1, bias_c2, a.params[i], a.grads[i], m, v);
        if (has_m) a.m[i] = m;
        if (has_v) a.v[i] = v;
        a.params[i] = p;
        if (sh) sh[i] = f2bf(p);
      }
    }
  }
  }
}

int g_opt_vec = 1, g_opt_blocks = 2, g_opt_nt = 0;   // measured best on the 152 M element arena: scalar lanes, 2 per thread

int elem_grid(long long items) {
  long long g = (items + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

}  // namespace

extern "C" int spb_bn_apply(int dtype, const spb_bnapply_args_t* a, spb_stream_t stream) {
  if (!a || !a->Z || !a->Y) return SPB_E_ARG;
  if (a->C <= 0 || (a->C & 7) || (a->ldc & 7) || (a->coff & 7)) return SPB_E_SHAPE;
  if (a->reorg > 0 && ((a->H % a->reorg) || (a->W % a->reorg))) return SPB_E_SHAPE;
  const long long items = (long long)a->B * a->H * a->W * (a->C >> 3);
  const size_t lds = (size_t)4 * a->C * sizeof(float);
  if (dtype == SPB_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(elem_grid(items)), dim3(256), lds, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(elem_grid(items)), dim3(256), lds, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

static int g_bbp_rows = 1;   // spb_debug_set_bn_bwd_prep_rows(0): the walking kernel (kept as the test reference)
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_bn_bwd_prep_rows(int on) { g_bbp_rows = on; return 0; }
#endif
extern "C" int spb_bn_bwd_prep(int dtype, const spb_bnbwd_args_t* a, spb_stream_t stream) {
  if (!a || !a->dY || !a->Z || !a->G || !a->osums || a->oR < 1) return SPB_E_ARG;
  if (a->C <= 0 || (a->C & 7) || (a->ldc & 7) || (a->coff & 7)) return SPB_E_SHAPE;
  if (a->reorg > 0 && ((a->H % a->reorg) || (a->W % a->reorg))) return SPB_E_SHAPE;
  if (g_bbp_rows) {   // row-parallel kernel (default)
    const long long rows = (long long)a->B * a->H * a->W;
    const dim3 grid((unsigned)((a->C + 63) / 64), (unsigned)((rows + BBP_ROWS - 1) / BBP_ROWS));
    if (dtype == SPB_BF16) hipLaunchKernelGGL(bn_bwd_prep_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    else if (dtype == SPB_F32) hipLaunchKernelGGL(bn_bwd_prep_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, *a);
    else return SPB_E_ARG;
    SPB_CHECK_LAUNCH();
    return 0;
  }
  const int CG = a->C >> 3;
  const long long items = (long long)a->B * a->H * a->W * CG;
  const int unit = CG / gcd_i(CG, 256);
  long long want = (items + 256 * 8 - 1) / (256 * 8);
  if (want > 256) want = 256;
  int grid = (int)((want + unit - 1) / unit) * unit;
  if (grid < unit) grid = unit;
  const size_t lds = (size_t)6 * a->C * sizeof(float);
  if (dtype == SPB_BF16) hipLaunchKernelGGL(bn_bwd_prep_kernel<bf16_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, *a);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(bn_bwd_prep_kernel<float>, dim3(grid), dim3(256), lds, (hipStream_t)stream, *a);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_bn_running_update(const spb_bnupd_entry_t* tab, int n_bn, const float* stats, float* buffers,
                                     long long* nbt, float momentum, spb_stream_t stream) {
  if (!tab || !stats || !buffers || n_bn <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(n_bn), dim3(256), 0, (hipStream_t)stream, tab, stats, buffers, nbt, momentum);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_bn_param_grads(const spb_bnupd_entry_t* tab, int n_bn, const float* stats, float* grads,
                                  spb_stream_t stream) {
  if (!tab || !stats || !grads || n_bn <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(n_bn), dim3(256), 0, (hipStream_t)stream, tab, stats, grads);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_bn_param_grads_zero(const spb_bnupd_entry_t* tab, int n_bn, float* stats, float* grads, spb_stream_t stream) {
  if (!tab || !stats || !grads || n_bn <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(bn_param_grads_zero_kernel, dim3(n_bn), dim3(256), 0, (hipStream_t)stream, tab, stats, grads);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_bn_load_running(const spb_bnupd_entry_t* tab, int n_bn, float* stats, const float* buffers,
                                   spb_stream_t stream) {
  if (!tab || !stats || !buffers || n_bn <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(bn_load_running_kernel, dim3(n_bn), dim3(256), 0, (hipStream_t)stream, tab, stats, buffers);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_weight_prep(int dtype, const spb_prep_entry_t* tab, int n_entries, int n_tiles, const float* params,
                               void* wc, spb_stream_t stream) {
  if (!tab || !params || !wc || n_entries <= 0 || n_tiles <= 0) return SPB_E_ARG;
  if (dtype == SPB_BF16) hipLaunchKernelGGL(weight_prep_kernel<bf16_t>, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, tab, n_entries, params, (bf16_t*)wc);
  else if (dtype == SPB_F32) hipLaunchKernelGGL(weight_prep_kernel<float>, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, tab, n_entries, params, (float*)wc);
  else return SPB_E_ARG;
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_grad_sqnorm(const float* grads, long long n, float* out, spb_stream_t stream) {
  if (!grads || !out || n <= 0) return SPB_E_ARG;
  // library-owned scratch: <= 2048 block partials + the arrival counter (zero between launches; one optimizer per process)
  static float* scratch = nullptr;
  if (!scratch) {
    if (hipMalloc(&scratch, 2049 * sizeof(float)) != hipSuccess) return SPB_E_STATE;
    if (hipMemset(scratch, 0, 2049 * sizeof(float)) != hipSuccess) return SPB_E_STATE;
  }
  const long long n4 = n >> 2;
  const int nblk = (int)(n4 >= 512 * 1024 ? 512 : (n4 + 1023) / 1024 > 0 ? (n4 + 1023) / 1024 : 1);
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, grads, n, out, scratch,
                     reinterpret_cast<unsigned*>(scratch + 2048));
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_grad_sqnorm_partials(const float* grads, long long n, float* partials_out, spb_stream_t stream) {
  if (!grads || !partials_out || n <= 0) return SPB_E_ARG;
  hipLaunchKernelGGL(grad_sqnorm_partials_kernel, dim3(SPB_SQ_PARTS), dim3(256), 0, (hipStream_t)stream, grads, n, partials_out);
  SPB_CHECK_LAUNCH();
  return 0;
}

// ---- flat gradient arena maintenance (optimizer.zero_grad(), and the sum of the two DANN passes' arenas, dann.py:95)
__global__ __launch_bounds__(256) void arena_add_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 a = reinterpret_cast<float4*>(dst)[i];
    const float4 b = reinterpret_cast<const float4*>(src)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reinterpret_cast<float4*>(dst)[i] = a;
  }
}
extern "C" int spb_arena_zero(float* arena, long long n, spb_stream_t stream) {
  if (!arena || n <= 0) return SPB_E_ARG;
  const hipError_t e = hipMemsetAsync(arena, 0, (size_t)n * sizeof(float), (hipStream_t)stream);
  return e == hipSuccess ? 0 : (int)e;
}
extern "C" int spb_arena_add(float* dst, const float* src, long long n, spb_stream_t stream) {
  if (!dst || !src || n <= 0 || (n & 3)) return SPB_E_ARG;       // arenas are padded to 4 floats (16-byte aligned tensors)
  const long long n4 = n >> 2;
  const int nblk = (int)(n4 >= 2048LL * 256 ? 2048 : (n4 + 255) / 256);
  hipLaunchKernelGGL(arena_add_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dst, src, n4);
  SPB_CHECK_LAUNCH();
  return 0;
}

extern "C" int spb_optim_step(const spb_optim_args_t* a, spb_stream_t stream) {
  if (!a || !a->params || !a->grads || a->n <= 0) return SPB_E_ARG;
  if (a->kind < 0 || a->kind > 3) return SPB_E_ARG;
  if (a->kind >= 2 && (!a->m || !a->v)) return SPB_E_ARG;
  if (a->kind == 1 && !a->v) return SPB_E_ARG;
  if (a->n_sq_partials < 0 || a->n_sq_partials > 256) return SPB_E_ARG;
  const bool vec = !(a->n & 3) && !(((uintptr_t)a->params | (uintptr_t)a->grads | (uintptr_t)a->m | (uintptr_t)a->v) & 15) &&
                   !((uintptr_t)a->shadow_bf16 & 7);
  const bool capped = a->max_blocks > 0;     // few workgroups: each thread keeps 4 x 16 bytes of every stream in flight
  const bool v4 = vec && (g_opt_vec == 4 || capped);
  const long long items = v4 ? a->n / 4 : a->n;
  const int U = capped ? 4 : (g_opt_blocks > 0 ? g_opt_blocks : 1);      // debug knob reused: vectors per thread
  unsigned blocks = (unsigned)((items + 256 * U - 1) / (256 * U));
  if (a->max_blocks > 0 && blocks > (unsigned)a->max_blocks) blocks = (unsigned)a->max_blocks;
  hipStream_t hs = (hipStream_t)stream;
#define SPB_OPT_LAUNCH(V, NTF, UU) hipLaunchKernelGGL((optim_step_kernel<V, NTF, UU>), dim3(blocks), dim3(256), 0, hs, *a)
  if (v4) {
    if (g_opt_nt) { if (U == 4) SPB_OPT_LAUNCH(4, true, 4); else if (U == 2) SPB_OPT_LAUNCH(4, true, 2); else SPB_OPT_LAUNCH(4, true, 1); }
    else { if (U == 4) SPB_OPT_LAUNCH(4, false, 4); else if (U == 2) SPB_OPT_LAUNCH(4, false, 2); else SPB_OPT_LAUNCH(4, false, 1); }
  } else {
    if (U == 4) SPB_OPT_LAUNCH(1, false, 4); else if (U == 2) SPB_OPT_LAUNCH(1, false, 2); else SPB_OPT_LAUNCH(1, false, 1);
  }
#undef SPB_OPT_LAUNCH
  SPB_CHECK_LAUNCH();
  return 0;
}

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_optim(int vec, int blocks, int nontemporal) {
  g_opt_vec = vec == 4 ? 4 : 1; g_opt_blocks = blocks; g_opt_nt = nontemporal;
  return 0;
}
#endif

// Streams for work that must not compete with the launch stream for the compute units: level -1 = the device's highest
// dispatch priority, 0 = default, +1 = lowest (the workgroup dispatcher serves higher-priority queues first, so a low stream's
// long memory-bound kernel fills the slots the launch stream's kernels leave free instead of taking half the chip from them).
extern "C" int spb_stream_create(int level, spb_stream_t* out) {
  if (!out) return SPB_E_ARG;
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) return (int)e;
  const int pr = level < 0 ? greatest : (level > 0 ? least : (least + greatest) / 2);
  hipStream_t s = nullptr;
  e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr);
  if (e != hipSuccess) return (int)e;
  *out = (spb_stream_t)s;
  return 0;
}
extern "C" int spb_stream_destroy(spb_stream_t s) {
  return s && hipStreamDestroy((hipStream_t)s) == hipSuccess ? 0 : SPB_E_ARG;
}

// ---- stream forks without events (HISTORY.md section 5, "Round 5, second half") -------------------------------------------------------
// Ordering a second stream behind the launch stream with an event costs the LAUNCH stream 5-9 us per fork, whatever the flavour
// (scratch/ubench_fork.hip: event record 8.7 us, completion event on the producer's dispatch packet 6.6 us, either without the system
// fence 6.3 / 5.2 us, stream write/wait value 9.0 us; the forked work itself is free).  A fork needs no event: the second stream runs a
// one-wave GATE kernel that spins on a device word, and the word is stored
//   * by a one-wave kernel on the launch stream (1.6 us per fork: spb_fork_streams, the KRN plan's forks that no depthwise kernel follows), or
//   * by the first thread of the launch stream's NEXT kernel (0.24 us: spb_dw_args_t::entry_flag, spb_publish_entry in common.h).
// Either store runs behind a barrier bit (every dispatch of this runtime carries barrier=1 and device-scope acquire + release fences:
// profiles/r5_fork_aql_headers.txt), so it proves every earlier launch of its stream complete and released at device scope; the
// kernels behind the gate start with their dispatch packet's own device-scope acquire, as behind an event.  Serial numbers only grow: the
// gate of fork n also passes when the word already holds n + k.  The storing kernel is always enqueued BEFORE the gate that waits for it
// (HIP streams may share a hardware queue: in host order every gate then depends on earlier entries only -- no cycle).
__global__ void fork_set_kernel(unsigned* flag, unsigned val) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A gate that gives up does NOT trap any more (round 6; a trap kills the GPU context of the whole process, a Python exception is what a
// caller can handle): it raises the POISON word -- host memory mapped into the device, owned by whoever owns the fork word -- and lets its
// stream go.  The owner reads the word at its next call (a plain host load) and fails that call with SPB_E_TIMEOUT; what the side
// stream computed in between is garbage by then, which is why the error is sticky.
__global__ void fork_gate_kernel(const unsigned* flag, unsigned val, unsigned long long timeout_ticks, unsigned* poison) {
  const unsigned long long t0 = wall_clock64();     // 100 MHz
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - val) < 0) {
    __builtin_amdgcn_s_sleep(8);
    // the storing launch never ran (a failed launch in between, or a tool that serialises the device's kernels): fail loudly
    // instead of hanging the device
    if (timeout_ticks != 0 && wall_clock64() - t0 > timeout_ticks) {
      if (poison) __hip_atomic_store(poison, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      else __builtin_trap();
      return;
    }
  }
}
// How long a gate waits before it gives up: the storing launch is already enqueued when the gate is, so the wait is the launch stream's own
// backlog -- which may hold a collective that waits for a late peer rank.  Default 600 s (the default watchdog of torch's process groups);
// SPB_FORK_TIMEOUT_S overrides it, 0 = wait forever (what an event does).
static unsigned long long fork_timeout_ticks() {
  static const unsigned long long v = [] {
    const char* e = std::getenv("SPB_FORK_TIMEOUT_S");
    const double sec = (e && e[0]) ? std::atof(e) : 600.0;
    return sec <= 0.0 ? 0ull : (unsigned long long)(sec * 1e8);
  }();
  return v;
}
void spb_fork_store(unsigned* flag, unsigned val, hipStream_t s) { hipLaunchKernelGGL(fork_set_kernel, dim3(1), dim3(64), 0, s, flag, val); }
void spb_fork_gate(const unsigned* flag, unsigned val, hipStream_t s, unsigned* poison) {
  hipLaunchKernelGGL(fork_gate_kernel, dim3(1), dim3(64), 0, s, flag, val, fork_timeout_ticks(), poison);
}
// the poison word of a fork: 64 bytes of pinned host memory the device can write (hipHostMalloc is mapped by default on this runtime)
unsigned* spb_fork_poison_alloc() {
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::memset(h, 0, 64);
  return static_cast<unsigned*>(h);
}
void spb_fork_poison_free(unsigned* p) { if (p) (void)hipHostFree(p); }
unsigned* spb_fork_poison_dev(unsigned* host) {
  void* d = nullptr;
  if (!host || hipHostGetDevicePointer(&d, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return static_cast<unsigned*>(d);
}

// ---- start-up self-test of the property the device-word forks rest on (round 6) ----------------------------------------------------
// "The first instruction of launch n + 1 proves launches <= n of its stream complete and their results visible device-wide" holds because
// this runtime gives every kernel dispatch the AQL barrier bit and device-scope acquire / release fences (read once from AMD_LOG_LEVEL=4
// output: profiles/r5_fork_aql_headers.txt).  Nothing in the HIP API promises it.  A runtime that dispatched without the barrier bit would
// turn every fork into a silent race in the weight gradients, so the property is TESTED before the first fork is trusted (at context /
// fork creation, outside any stream capture, ~1 ms once per process and device):
//   stream A:  producer (512 workgroups; each thread waits ~20 us, then writes the round's serial into its words of a 2 MB buffer)
//              -> one-wave kernel whose first thread stores the serial to the fork word      (exactly what spb_publish_entry does)
//   stream B:  gate on the word -> checker: counts the buffer words that do NOT hold the serial
// over 6 rounds.  Without the barrier bit the one-wave kernel overtakes the producer (it needs one free slot, the producer's threads sit
// in their wait), the gate opens and the checker finds stale words.  Any mismatch, or any HIP error on the way: forks are ordered by
// events for the rest of the process and one line says so on stderr.  SPB_FORK_SELFTEST_FAIL=1 forces the failure (test rig).
namespace {
__global__ void forktest_produce_kernel(unsigned* buf, int n, unsigned serial) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000ull) __builtin_amdgcn_s_sleep(16);        // ~20 us at 100 MHz: the dependent launch has every chance to overtake
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = serial;
}
__global__ void forktest_check_kernel(const unsigned* buf, int n, unsigned serial, unsigned* bad) {
  unsigned miss = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) miss += buf[i] != serial;
  if (miss) atomicAdd(bad, miss);
}
int g_fork_selftest = -2;        // -2 not run yet, -1 events forced by the environment, 0 failed (events), 1 passed (device-word forks)
}  // namespace
bool spb_event_forks_forced();
// The test itself, on the two streams it is given.  (Round 6, first version: the test created two streams of its own at fork / context
// creation -- and the SPN step went from 1.67 to 2.9 ms: the HIP runtime deals hardware queues to streams round robin, two more streams
// shifted the deal and the SPN's update stream ended up sharing a queue.  It now runs lazily on the very streams of the first fork.)
static int fork_selftest_run(hipStream_t a, hipStream_t b) {
  const char* ff = std::getenv("SPB_FORK_SELFTEST_FAIL");
  bool ok = !(ff && ff[0] == '1');
  constexpr int N = 512 * 1024;
  unsigned *buf = nullptr, *word = nullptr, *bad = nullptr, *poison = spb_fork_poison_alloc();
  unsigned hbad = 1;
  if (ok) {
    ok = hipMalloc(&buf, N * sizeof(unsigned)) == hipSuccess && hipMalloc(&word, 256) == hipSuccess && hipMalloc(&bad, 256) == hipSuccess &&
         hipMemset(buf, 0, N * sizeof(unsigned)) == hipSuccess && hipMemset(word, 0, 256) == hipSuccess && hipMemset(bad, 0, 256) == hipSuccess &&
         hipStreamSynchronize(nullptr) == hipSuccess && poison != nullptr;
    if (ok) {
      for (unsigned r = 1; r <= 6; ++r) {
        hipLaunchKernelGGL(forktest_produce_kernel, dim3(512), dim3(256), 0, a, buf, N, r);
        hipLaunchKernelGGL(fork_set_kernel, dim3(1), dim3(64), 0, a, word, r);
        hipLaunchKernelGGL(fork_gate_kernel, dim3(1), dim3(64), 0, b, (const unsigned*)word, r, 200000000ull /* 2 s */, spb_fork_poison_dev(poison));
        hipLaunchKernelGGL(forktest_check_kernel, dim3(256), dim3(256), 0, b, (const unsigned*)buf, N, r, bad);
      }
      ok = hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess &&
           hipMemcpy(&hbad, bad, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess && hipGetLastError() == hipSuccess;
      if (ok) ok = hbad == 0 && *reinterpret_cast<volatile unsigned*>(poison) == 0;
    }
  }
  (void)hipGetLastError();
  if (buf) (void)hipFree(buf);
  if (word) (void)hipFree(word);
  if (bad) (void)hipFree(bad);
  spb_fork_poison_free(poison);
  g_fork_selftest = ok ? 1 : 0;
  if (!ok) {
    int rv = 0; (void)hipRuntimeGetVersion(&rv);
    std::fprintf(stderr, "speedplusbaseline_amd: the stream-fork self-test FAILED on this HIP runtime (version %d; %u stale words): a dependent "
                 "launch started before its predecessor's results were visible.  Streams are ordered by events for the rest of this process "
                 "(as with SPB_EVENT_FORKS=1).\n", rv, hbad);
  }
  return g_fork_selftest;
}
// public query (include/spb_hip.h): the cached verdict; if no fork has run the test yet, on two streams of its own
extern "C" int spb_fork_selftest(void) {
  if (g_fork_selftest != -2) return g_fork_selftest;
  if (spb_event_forks_forced()) return g_fork_selftest = -1;
  hipStream_t a = nullptr, b = nullptr;
  if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&b, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    if (a) (void)hipStreamDestroy(a);
    return g_fork_selftest = 0;
  }
  const int r = fork_selftest_run(a, b);
  (void)hipStreamDestroy(a); (void)hipStreamDestroy(b);
  return r;
}
extern "C" int spb_hip_runtime_version(void) { int v = 0; return hipRuntimeGetVersion(&v) == hipSuccess ? v : -1; }

// Tools that let only ONE kernel of the device run at a time cannot run a spinning gate: the launch it waits for would never start
// (measured: `rocprofv3 --pmc ...` hangs until the gate's time-out traps).  Streams are ordered by events instead when
//   * SPB_EVENT_FORKS=1 is in the environment, or
//   * ROCPROF_COUNTER_COLLECTION=1 is (what rocprofv3 exports to the application for --pmc / counter-group runs), or the runtime's own
//     serialising debug switches AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING are,
// and inside a stream capture (a replayed graph would replay the serial numbers).
bool spb_event_forks_forced() {
  static const bool v = [] {
    // (AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING: debug settings of the HIP runtime that make every launch wait for the previous one, on
    // whatever stream -- a gate would then never see the store enqueued behind it)
    for (const char* name : {"SPB_EVENT_FORKS", "ROCPROF_COUNTER_COLLECTION", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING"}) {
      const char* e = std::getenv(name);
      if (e && e[0] && e[0] != '0' && e[0] != 'f' && e[0] != 'F') return true;
    }
    return false;
  }();
  return v;
}
// `to`: the stream the first fork of the process will gate -- the self-test runs on this very pair (once; synchronises both)
bool spb_fork_by_word(hipStream_t from, hipStream_t to) {
  if (spb_event_forks_forced()) return false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(from, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
  if (g_fork_selftest == -2) {
    if (to == nullptr || to == from) return false;
    if (hipStreamIsCapturing(to, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    (void)fork_selftest_run(from, to);
  }
  return g_fork_selftest == 1;
}

struct spb_fork { unsigned* word = nullptr; unsigned serial = 0; hipEvent_t ev = nullptr; unsigned* poison = nullptr; };
extern "C" int spb_fork_create(spb_fork_t** out) {
  if (!out) return SPB_E_ARG;
  spb_fork* f = new spb_fork();
  // the word must BE zero before the first gate can run: hipMemset on device memory returns before the fill has executed, and a gate
  // on a non-blocking stream does not wait for the null stream -- on recycled memory it would read the last serial of a destroyed fork
  if (hipMalloc(&f->word, 256) != hipSuccess || hipMemset(f->word, 0, 256) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess ||
      hipEventCreateWithFlags(&f->ev, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
    if (f->word) hipFree(f->word);
    delete f;
    return SPB_E_STATE;
  }
  f->poison = spb_fork_poison_alloc();
  *out = f;
  return 0;
}
extern "C" void spb_fork_destroy(spb_fork_t* f) {
  if (!f) return;
  if (f->word) hipFree(f->word);
  if (f->ev) hipEventDestroy(f->ev);
  spb_fork_poison_free(f->poison);
  delete f;
}
extern "C" int spb_fork_streams(spb_fork_t* f, spb_stream_t from, spb_stream_t to) {
  if (!f) return SPB_E_ARG;
  hipStream_t a = (hipStream_t)from, b = (hipStream_t)to;
  if (a == b) return 0;
  if (f->poison && *reinterpret_cast<volatile unsigned*>(f->poison)) return SPB_E_TIMEOUT;   // an earlier gate of this fork gave up (sticky)
  if (spb_fork_by_word(a, b)) {
    const unsigned serial = ++f->serial;
    spb_fork_store(f->word, serial, a);
    spb_fork_gate(f->word, serial, b, spb_fork_poison_dev(f->poison));
  } else {
    if (hipEventRecord(f->ev, a) != hipSuccess || hipStreamWaitEvent(b, f->ev, 0) != hipSuccess) return SPB_E_STATE;
  }
  SPB_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Reduce pass of the weight-gradient partials (spb_red_job_t, include/spb_hip.h): dst[i] += sum_p src[p*stride + i].
// One launch takes up to RED_MAX jobs (the table travels in the kernel arguments); a workgroup owns 1024 consecutive
// elements of one job and RED_PC of its parts, all of a thread's loads in flight together; jobs with more than RED_PC parts
// finish with one f32 atomic per element and part chunk (16-32x fewer than one per workgroup of the producing kernel).
namespace {
constexpr int RED_MAX = 16, RED_PC = 16;
struct RedTab { spb_red_job_t j[RED_MAX]; int blk0[RED_MAX + 1]; int njobs; };

__global__ __launch_bounds__(256) void partial_reduce_kernel(const RedTab tab) {
  int ji = 0;
  while (ji + 1 < tab.njobs && (int)blockIdx.x >= tab.blk0[ji + 1]) ++ji;
  const spb_red_job_t jb = tab.j[ji];
  const int eb = (jb.n + 1023) / 1024;
  const int lb = (int)blockIdx.x - tab.blk0[ji];
  const int chunk = lb / eb, i = ((lb - chunk * eb) * 256 + (int)threadIdx.x) * 4;
  if (i >= jb.n) return;
  const int p0 = chunk * RED_PC, p1 = min(jb.nparts, p0 + RED_PC);
  const float* src = jb.src + (size_t)p0 * jb.stride + i;
  float4 v[RED_PC];
#pragma unroll
  for (int p = 0; p < RED_PC; ++p) {   // clamped part index: every load issued before the first add
    const int pc = p0 + p < p1 ? p : p1 - 1 - p0;
    v[p] = *reinterpret_cast<const float4*>(src + (size_t)pc * jb.stride);
  }
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < RED_PC; ++p)
    if (p0 + p < p1) { a.x += v[p].x; a.y += v[p].y; a.z += v[p].z; a.w += v[p].w; }
  float* d = jb.dst + i;
  if (jb.nparts <= RED_PC) {
    float4 o = *reinterpret_cast<float4*>(d);
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    *reinterpret_cast<float4*>(d) = o;
  } else { atomicAdd(d, a.x); atomicAdd(d + 1, a.y); atomicAdd(d + 2, a.z); atomicAdd(d + 3, a.w); }
}
}  // namespace

extern "C" int spb_partial_reduce(const spb_red_job_t* jobs, int njobs, spb_stream_t stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return SPB_E_ARG;
  int done = 0;
  while (done < njobs) {
    RedTab tab; std::memset(&tab, 0, sizeof(tab));
    int nb = 0, k = 0;
    for (; k < RED_MAX && done + k < njobs; ++k) {
      const spb_red_job_t& j = jobs[done + k];
      if (!j.src || !j.dst || j.n <= 0 || (j.n & 3) || j.nparts <= 0 || (j.stride & 3) ||
          (reinterpret_cast<uintptr_t>(j.src) & 15) || (reinterpret_cast<uintptr_t>(j.dst) & 15)) return SPB_E_ARG;
      tab.j[k] = j; tab.blk0[k] = nb;
      nb += ((j.n + 1023) / 1024) * ((j.nparts + RED_PC - 1) / RED_PC);
    }
    tab.blk0[k] = nb; tab.njobs = k;
    hipLaunchKernelGGL(partial_reduce_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, tab);
    SPB_CHECK_LAUNCH();
    done += k;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Dynamic loss scaling on the device (torch.cuda.amp.GradScaler as the reference's fp16 recipe uses it, trainer.py:146-181:
// scaler.scale(loss).backward(); scaler.unscale_(optimizer); clip_grad_value_; scaler.step(optimizer); scaler.update()) --
// without the host synchronisation GradScaler.step() makes to read found_inf.  State: SPB_AMP_STATE floats, see spb_hip.h.
namespace {
__global__ __launch_bounds__(256) void amp_check_kernel(const float* __restrict__ g, long long n, float* state) {
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = g4[i];
    // finite <=> |x| <= FLT_MAX; a NaN compares false
    bad |= !(fabsf(v.x) <= 3.4028234e38f) | !(fabsf(v.y) <= 3.4028234e38f) | !(fabsf(v.z) <= 3.4028234e38f) | !(fabsf(v.w) <= 3.4028234e38f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= !(fabsf(g[(n4 << 2) + threadIdx.x]) <= 3.4028234e38f);
  if (bad) state[SPB_AMP_FOUND_INF] = 1.f;   // racing writers store the same value
}
__global__ void amp_step_kernel(float* st, float lr, float beta1, float beta2, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool found = st[SPB_AMP_FOUND_INF] != 0.f;
  st[SPB_AMP_INV_SCALE] = 1.f / st[SPB_AMP_SCALE];           // the unscale factor of the step in flight
  st[SPB_AMP_SKIP] = found ? 1.f : 0.f;
  st[SPB_AMP_LR] = lr;
  if (!found) {                                               // the optimizer's own step count only advances on steps it takes
    const float t = st[SPB_AMP_STEPS] + 1.f;
    st[SPB_AMP_STEPS] = t;
    st[SPB_AMP_BC1] = beta1 > 0.f ? 1.f - powf(beta1, t) : 1.f;
    st[SPB_AMP_BC2] = beta2 > 0.f ? 1.f - powf(beta2, t) : 1.f;
  }
  // GradScaler.update(): back off at once, grow after `interval` clean steps in a row
  if (found) { st[SPB_AMP_SCALE] *= backoff; st[SPB_AMP_TRACKER] = 0.f; }
  else {
    const float k = st[SPB_AMP_TRACKER] + 1.f;
    if (k >= (float)interval) { st[SPB_AMP_SCALE] *= growth; st[SPB_AMP_TRACKER] = 0.f; } else st[SPB_AMP_TRACKER] = k;
  }
  st[SPB_AMP_FOUND_INF] = 0.f;
}
}  // namespace

extern "C" int spb_amp_check(const float* grads, long long n, float* state, spb_stream_t stream) {
  if (!grads || !state || n <= 0) return SPB_E_ARG;
  const long long n4 = n >> 2;
  const int nblk = (int)(n4 >= 2048LL * 256 * 4 ? 2048 : (n4 + 1023) / 1024 > 0 ? (n4 + 1023) / 1024 : 1);
  hipLaunchKernelGGL(amp_check_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, grads, n, state);
  SPB_CHECK_LAUNCH();
  return 0;
}
// the same check on a 16-bit tensor (n % 8 == 0, 16-byte aligned).  A fully connected layer's weight gradient dW = g^T x (f32 accumulation
// of at most 64 products of 16-bit values: no overflow of its own) is finite exactly when its operands are, so SpnOptimizer checks the six
// [F][MP] gradient operands (a few hundred KB) instead of the 600 MB of f32 weight gradients they produce
namespace {
__global__ __launch_bounds__(256) void amp_check16_kernel(const bf16_t* __restrict__ x, long long n8, float* state) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float v[8];
    ld8<bf16_t>(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) bad |= !(fabsf(v[j]) <= 3.4028234e38f);
  }
  if (bad) state[SPB_AMP_FOUND_INF] = 1.f;
}
}  // namespace
extern "C" int spb_amp_check16(const void* x, long long n, float* state, spb_stream_t stream) {
  if (!x || !state || n <= 0 || (n & 7)) return SPB_E_ARG;
  const long long n8 = n >> 3;
  const int nblk = (int)((n8 + 255) / 256 > 1024 ? 1024 : (n8 + 255) / 256);
  hipLaunchKernelGGL(amp_check16_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16_t*>(x), n8, state);
  SPB_CHECK_LAUNCH();
  return 0;
}
// One launch for the whole decision: up to SPB_AMP_SEGS segments (f32 or 16-bit) are checked, and the LAST workgroup to finish (a ticket in
// state[SPB_AMP_TICKET]) runs spb_amp_step's arithmetic -- nine launches (check x 7, step, ...) on the optimizer's critical path become one.
namespace {
__global__ __launch_bounds__(256) void amp_decide_kernel(const spb_amp_segs_t sg, float* state, float lr, float beta1, float beta2, float growth,
                                                         float backoff, int interval) {
  bool bad = false;
  for (int s = 0; s < sg.nseg; ++s) {
    const long long nv = sg.n[s] >> (sg.is16[s] ? 3 : 2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
      if (sg.is16[s]) {
        float v[8];
        ld8<bf16_t>(reinterpret_cast<const bf16_t*>(sg.ptr[s]) + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) bad |= !(fabsf(v[j]) <= 3.4028234e38f);
      } else {
        const float4 v = reinterpret_cast<const float4*>(sg.ptr[s])[i];
        bad |= !(fabsf(v.x) <= 3.4028234e38f) | !(fabsf(v.y) <= 3.4028234e38f) | !(fabsf(v.z) <= 3.4028234e38f) | !(fabsf(v.w) <= 3.4028234e38f);
      }
    }
  }
  const int any_bad = __syncthreads_or(bad ? 1 : 0);      // one lane publishes the workgroup's finding, then takes the ticket (its own release covers it)
  if (threadIdx.x == 0) {
    if (any_bad) __hip_atomic_store(state + SPB_AMP_FOUND_INF, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bool last;
    const unsigned prev = atomicAdd(reinterpret_cast<unsigned*>(state + SPB_AMP_TICKET), 1u);
    last = prev == gridDim.x - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      *reinterpret_cast<unsigned*>(state + SPB_AMP_TICKET) = 0u;
      float* st = state;
      const bool found = __hip_atomic_load(st + SPB_AMP_FOUND_INF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.f;
      st[SPB_AMP_INV_SCALE] = 1.f / st[SPB_AMP_SCALE];
      st[SPB_AMP_SKIP] = found ? 1.f : 0.f;
      st[SPB_AMP_LR] = lr;
      if (!found) {
        const float t = st[SPB_AMP_STEPS] + 1.f;
        st[SPB_AMP_STEPS] = t;
        st[SPB_AMP_BC1] = beta1 > 0.f ? 1.f - powf(beta1, t) : 1.f;
        st[SPB_AMP_BC2] = beta2 > 0.f ? 1.f - powf(beta2, t) : 1.f;
      }
      if (found) { st[SPB_AMP_SCALE] *= backoff; st[SPB_AMP_TRACKER] = 0.f; }
      else {
        const float k = st[SPB_AMP_TRACKER] + 1.f;
        if (k >= (float)interval) { st[SPB_AMP_SCALE] *= growth; st[SPB_AMP_TRACKER] = 0.f; } else st[SPB_AMP_TRACKER] = k;
      }
      st[SPB_AMP_FOUND_INF] = 0.f;
    }
  }
}
}  // namespace
extern "C" int spb_amp_decide(const spb_amp_segs_t* segs, float* state, float lr, float beta1, float beta2, float growth, float backoff,
                              int interval, spb_stream_t stream) {
  if (!segs || !state || interval < 1 || segs->nseg < 1 || segs->nseg > SPB_AMP_SEGS) return SPB_E_ARG;
  long long work = 0;
  for (int s = 0; s < segs->nseg; ++s) {
    if (!segs->ptr[s] || segs->n[s] <= 0 || (segs->n[s] & (segs->is16[s] ? 7 : 3))) return SPB_E_ARG;
    work += segs->n[s] >> (segs->is16[s] ? 3 : 2);
  }
  const int nblk = (int)((work + 1023) / 1024 > 512 ? 512 : ((work + 1023) / 1024 < 1 ? 1 : (work + 1023) / 1024));
  hipLaunchKernelGGL(amp_decide_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *segs, state, lr, beta1, beta2, growth, backoff, interval);
  SPB_CHECK_LAUNCH();
  return 0;
}
extern "C" int spb_amp_step(float* state, float lr, float beta1, float beta2, float growth, float backoff, int interval, spb_stream_t stream) {
  if (!state || interval < 1) return SPB_E_ARG;
  hipLaunchKernelGGL(amp_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, beta1, beta2, growth, backoff, interval);
  SPB_CHECK_LAUNCH();
  return 0;
}
