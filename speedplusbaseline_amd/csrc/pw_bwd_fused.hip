// Fused backward of a pointwise (1x1) convolution for the wide-and-shallow layers at the top of MobileNetV2
// (112x112 and 56x56 maps: M = B*H*W is 150k..600k rows, N*K is a few thousand weights).
// Replaces, for those layers, the pair spb_pwconv_gemm(pro 2, epi 2) + spb_pwconv_wgrad -- i.e. the autograd backward
// of nn.Conv2d(k=1) + BatchNorm2d + ReLU6 in torchvision's InvertedResidual (reference call site park2019.py:107-108).
//
//   dz[m,n]  = g[m,n]*p0[n] + z[m,n]*p1[n] + p2[n]                 (BN backward of the conv output, rebuilt on the fly)
//   gin[m,k] = round((sum_n dz[m,n] W[n,k] + res[m,k]) * act'(u_in[m,k]))    + its BN-backward sums over m
//   dW[n,k] += sum_m dz[m,n] * a[m,k]                               (a = the conv input, act(bn(x)) or materialised)
//
// The two GEMMs share their big operand: g and z ([M,N], the dominant bytes) are read from HBM once instead of twice,
// and everything else a row needs travels with it.  One wave owns 32-row chunks end to end, nothing is exchanged
// between waves until the final reduction:
//   * an LDS-DMA double buffer (global_load_lds_dwordx4) holds the raw rows of the next chunk: g, z, the input-side z,
//     the conv input and the residual gradient; no VGPRs are spent on loads in flight;
//   * W^T fragments stay in registers for the whole kernel;
//   * input gradient as (gin)^T = W^T * dz^T: the MFMA B operand is then a plain 16-byte row read of dz, and the result
//     leaves the matrix core as 4 consecutive k per lane -- row-major, no transpose through LDS;
//   * weight gradient needs both operands with the reduction axis (m) fastest: dz and a are parked row-major in
//     wave-private LDS tiles and read back with the gfx950 transpose load (ds_read_b64_tr_b16);
//   * wide K (144) is split over blockIdx.y so that accumulators fit; g,z are then read once per split.
// bf16 only (the f32 parity mode keeps the two-kernel path).
//
// Round 4: the kernel was bound by how few waves a CU could hold, not by memory: 32-row double-buffered stages cost a wave
// 34-55 KB of LDS, so the 56x56 layers ran 3-4 waves per CU -- one per SIMD, every dependent instruction at its full latency
// (9 us per 32-row chunk of the 24 -> 144 layer; 1.4-2.0 TB/s).  Now:
//   * CHR = 16-row chunks (template): half the LDS per wave; workgroups of up to 8 waves share the coefficient tables;
//   * RZ (expand layers, K <= 32): the raw conv output z is NOT read -- it is recomputed on the matrix cores from the conv input
//     rows the kernel stages anyway (z = W a: two 16x16x32 MFMAs per 16 rows x 32 columns, W rows permuted so that the result
//     lands in the lane layout the dz fragment needs).  That removes ~45 % of the launch's bytes (115 MB of 270 MB for 16 -> 96
//     at 112x112) and the largest piece of every stage;
//   * the block-level weight-gradient reduction goes wave by wave through ONE [NB][KB][256] tile instead of nw of them.
#include "common.h"

namespace {


// dz tile: when N is a multiple of 16 the transformed dz overwrites the raw g rows it came from (same lane, same
// 16 bytes; leading dimension N); otherwise (N = 24) the 16-column blocks overrun a row and a padded tile is used.
struct PwbLay { int gB, kB, stage, oZ, oZo, oX, oR, oDzt, oAt, ldz, wave_bytes; };
__host__ __device__ inline PwbLay pwb_lay(int CH, int N, int KW, int NPAD, int KPAD, bool hasx, bool hasr, bool hasz) {
  PwbLay L;
  const bool inplace = (N & 15) == 0;
  L.gB = (CH * N * 2 + 1023) & ~1023;     // raw g (and z) rows of a chunk, whole 1 KB DMA instructions
  L.kB = (CH * KW * 2 + 1023) & ~1023;    // one K-side tile
  L.oZ = L.gB; L.oZo = (hasz ? 2 : 1) * L.gB; L.oX = L.oZo + L.kB; L.oR = L.oX + (hasx ? L.kB : 0);
  L.stage = L.oR + (hasr ? L.kB : 0);
  L.oDzt = inplace ? -1 : 2 * L.stage;
  L.ldz = inplace ? N : NPAD;
  L.oAt = 2 * L.stage + (inplace ? 0 : CH * NPAD * 2);
  L.wave_bytes = (L.oAt + CH * KPAD * 2 + 1023) & ~1023;
  return L;
}

__device__ __forceinline__ void unpack4(uint2 r, float v[4]) {
  spb_unpack2(r.x, v[0], v[1]); spb_unpack2(r.y, v[2], v[3]);      // (bf16 or IEEE half: common.h)
}

// transpose-load fragment: 32 rows x 16 columns (c0..c0+15) of a row-major bf16 LDS tile with leading dimension LD
// -> lane (li, lq) gets column c0+li, rows lq*8 .. lq*8+7  (same addressing as pw_wgrad_kernel in gemm_pw.hip)
// ROWS = 16: the tile has 16 rows only; the upper half of the 32-deep reduction (lanes lq >= 2) is zero
template <int ROWS>
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* tile, int LD, int c0, int li, int lq) {
  typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
  const int lqc = ROWS == 32 ? lq : (lq & 1);            // (clamped: the read stays inside the tile, the value is dropped)
  const bf16_t* p = tile + (lqc * 8 + (li >> 2)) * LD + c0 + (li & 3) * 4;
  union { struct { s16x4_t lo, hi; } s; bf16x8_t v; uint4 q; } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * LD));
  if (ROWS != 32 && lq >= 2) u.q = make_uint4(0, 0, 0, 0);
  return u.v;
}

template <int NB, int KB, int CH, bool RZ>
__global__ __launch_bounds__(512, 2) void pwb_kernel(const spb_pwbwd_args_t g, long long nchunks) {
  constexpr int NS = (NB + 1) / 2;               // 32-wide reduction steps of the input-gradient GEMM
  constexpr int NP = NS * 32, NPAD = NP + 8;
  constexpr int KW = KB * 16, KPAD = KW + 8;
  constexpr int GPR = KW / 8;                    // 16-byte granules per K-side tile row
  constexpr int NH = CH / 16;                    // 16-row halves of a chunk
  static_assert(!RZ || KW <= 32, "z is recomputed in one 32-deep MFMA step: expand layers only (K <= 32)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int M = g.M, N = g.N, K = g.K;
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = (int)(blockDim.x >> 6), nthr = (int)blockDim.x;   // 2..8 waves per block, whatever fits the LDS
  const int k0 = blockIdx.y * KW;                // first input channel of this split
  const bool hasx = g.X != g.Zout, hasr = g.res != nullptr;
  const PwbLay L = pwb_lay(CH, N, KW, NPAD, KPAD, hasx, hasr, !RZ);
  float* pdz = reinterpret_cast<float*>(smem);   // [3][NP]  p0, p1, p2 of the conv output's BN backward
  float* pep = pdz + 3 * NP;                     // [4][KW]  scale, shift, mean, invstd of the input-side BN
  float* pa = pep + 4 * KW;                      // [2][KW]  scale, shift that turn X into the conv input
  // RZ: W fragments of the z recomputation (A operand; [NS][2][64 lanes] x 16 B), shared by the workgroup's waves
  uint4* wz = reinterpret_cast<uint4*>(pa + 2 * KW);
  char* wreg = reinterpret_cast<char*>(wz + (RZ ? NS * 2 * 64 : 0)) + (size_t)wave * L.wave_bytes;
  if (g.pro_dz.gamma != nullptr && !g.pro_dz.moments && g.pro_a.gamma != nullptr && !g.pro_a.moments && g.epi.gamma != nullptr && !g.epi.moments) {
    // (uniform, the training step) the sums of all three BatchNorms are requested before any of them is used: one memory round trip
    // instead of three (output-side backward coefficients, then the input side's moments, then its affine)
    const int top = NP > KW ? NP : KW;
    for (int i = threadIdx.x; i < top; i += nthr) {
      BNLoad ldz, lep, lpa; BNLoadB bdz;
      const int in = i < N ? i : N - 1, k = k0 + i < K ? k0 + i : K - 1;
      bn_issue(g.pro_dz, in, ldz); bn_issue_bwd(g.pro_dz, in, bdz);
      bn_issue(g.epi, k, lep);
      bn_issue(g.pro_a, k, lpa);
      float p0, p1, p2, mu, is, amu, ais;
      bn_finish_bwd(g.pro_dz, ldz, bdz, p0, p1, p2);
      bn_finish(g.epi, lep, mu, is);
      bn_finish(g.pro_a, lpa, amu, ais);
      if (i < NP) { const bool ok = i < N; pdz[i] = ok ? p0 : 0.f; pdz[NP + i] = ok ? p1 : 0.f; pdz[2 * NP + i] = ok ? p2 : 0.f; }
      if (i < KW) {
        const bool ok = k0 + i < K;
        const float sc = lep.gm * is, asc = lpa.gm * ais;
        pep[i] = ok ? sc : 1.f; pep[KW + i] = ok ? lep.bt - mu * sc : 0.f; pep[2 * KW + i] = ok ? mu : 0.f; pep[3 * KW + i] = ok ? is : 0.f;
        pa[i] = ok ? asc : 0.f; pa[KW + i] = ok ? lpa.bt - amu * asc : 0.f;
      }
    }
  } else {
  for (int i = threadIdx.x; i < NP; i += nthr) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (i < N) bn_bwd_coef(g.pro_dz, i, p0, p1, p2);
    pdz[i] = p0; pdz[NP + i] = p1; pdz[2 * NP + i] = p2;
  }
  for (int i = threadIdx.x; i < KW; i += nthr) {
    const int k = k0 + i;
    float sc = 1.f, sh = 0.f, mu = 0.f, is = 0.f, asc = 0.f, ash = 0.f;
    if (k < K) {
      if (g.epi.gamma != nullptr) {
        bn_moments(g.epi, k, mu, is);
        sc = g.epi.gamma[k] * is; sh = g.epi.beta[k] - mu * sc;
      }
      bn_fwd_coef(g.pro_a, k, asc, ash);
    }
    pep[i] = sc; pep[KW + i] = sh; pep[2 * KW + i] = mu; pep[3 * KW + i] = is;
    pa[i] = asc; pa[KW + i] = ash;
  }
  }
  const bf16_t* Wt = reinterpret_cast<const bf16_t*>(g.Wt);
  if constexpr (RZ) {
    // z^T tile (32 columns n of 16 rows m) = two MFMAs whose A rows are PERMUTED weight rows: fragment t, row r  <->  output
    // channel n = ns*32 + (r>>2)*8 + t*4 + (r&3).  The C layout then gives lane (li = m, lq) the channels ns*32 + lq*8 + t*4 + i:
    // with t = 0, 1 exactly the 8 consecutive channels of the dz fragment below.  A fragment element: W[n][8*lq + j] = Wt[k][n].
    for (int e = threadIdx.x; e < NS * 2 * 64; e += nthr) {
      const int f = e >> 6, l = e & 63, r = l & 15, q = l >> 4;
      const int n = (f >> 1) * 32 + (r >> 2) * 8 + (f & 1) * 4 + (r & 3);
      bf16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = q * 8 + j;
        v[j] = (n < N && k < K) ? Wt[(size_t)k * N + n] : (bf16_t)0;
      }
      uint4 u;
      u.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); u.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
      u.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16); u.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
      wz[e] = u;
    }
  }
  // W^T fragments (A operand of the input-gradient MFMA), rows permuted: fragment kb, lane row li  <->  input channel
  // k0 + (li>>2)*4*KB + kb*4 + (li&3).  The KB accumulators of a lane then cover 4*KB CONSECUTIVE channels of its row:
  // 16-byte output stores instead of one scattered 8-byte store per accumulator (64 pieces per instruction).
  bf16x8_t Wf[KB][NS];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int k = k0 + (li >> 2) * 4 * KB + kb * 4 + (li & 3), n = ns * 32 + lq * 8;
      uint4 u = make_uint4(0, 0, 0, 0);
      if (k < K && n < N) u = *reinterpret_cast<const uint4*>(Wt + (size_t)k * N + n);
      Wf[kb][ns] = __builtin_bit_cast(bf16x8_t, u);
    }
  __syncthreads();

  const char* Gp = reinterpret_cast<const char*>(g.G);
  const char* Zp = reinterpret_cast<const char*>(g.Zn);
  const char* Zop = reinterpret_cast<const char*>(g.Zout);
  const char* Xp = reinterpret_cast<const char*>(g.X);
  const char* Rp = reinterpret_cast<const char*>(g.res);
  bf16_t* Y = reinterpret_cast<bf16_t*>(g.Y);
  const unsigned wlds = lds_addr(wreg);
  const long long gtot = (long long)M * N * 2;
  const int eact = g.epi.act, aact = g.pro_a.act;
  const float eslope = g.epi.slope, aslope = g.pro_a.slope;

  auto issue = [&](long long c, int s) {
    const unsigned sb = wlds + (unsigned)(s * L.stage);
    const long long m0 = c * CH;
    for (int i = 0; i < (L.gB >> 10); ++i) {
      long long off = m0 * N * 2 + (long long)(i * 64 + lane) * 16;
      off = off > gtot - 16 ? gtot - 16 : off;       // past the tensor: any valid address, the rows are masked
      dma16(Gp + off, sb + (unsigned)(i << 10));
      if constexpr (!RZ) dma16(Zp + off, sb + (unsigned)(L.oZ + (i << 10)));
    }
    for (int i = 0; i < (L.kB >> 10); ++i) {
      const int q = i * 64 + lane;
      long long m = m0 + q / GPR;
      m = m > M - 1 ? M - 1 : m;
      int k = k0 + (q % GPR) * 8;
      k = k > K - 8 ? K - 8 : k;
      const long long off = (m * K + k) * 2;
      dma16(Zop + off, sb + (unsigned)(L.oZo + (i << 10)));
      if (hasx) dma16(Xp + off, sb + (unsigned)(L.oX + (i << 10)));
      if (hasr) dma16(Rp + off, sb + (unsigned)(L.oR + (i << 10)));
    }
  };

  f32x4_t dw[NB][KB];
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int b = 0; b < KB; ++b) dw[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float s1[KB][4], s2[KB][4];
#pragma unroll
  for (int b = 0; b < KB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[b][i] = 0.f; s2[b][i] = 0.f; }

  bf16_t* at = reinterpret_cast<bf16_t*>(wreg + L.oAt);
  const int LDZ = L.ldz;
  const long long cstride = (long long)gridDim.x * nw;
  long long c = (long long)blockIdx.x * nw + wave;
  if (c < nchunks) issue(c, 0);
  int s = 0;
  for (; c < nchunks; c += cstride, s ^= 1) {
    wait_vmcnt<0>();                                   // stage s has landed; nothing else of this wave is in flight
    if (c + cstride < nchunks) issue(c + cstride, s ^ 1);  // the other stage was consumed in the previous pass
    char* st = wreg + s * L.stage;
    bf16_t* dzt = reinterpret_cast<bf16_t*>(L.oDzt < 0 ? st : wreg + L.oDzt);
    const long long m0 = c * CH;
    // ---- a = the conv input (act(bn(X)) or the materialised tensor as stored), row-major bf16 tile: operand of the weight
    //      gradient's transpose loads and (RZ) of the z recomputation
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int row = h * 16 + li;
      const long long m = m0 + row;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int kl = lq * 4 * KB + kb * 4, k = k0 + kl;
        const bool ok = m < M && k < K;
        float xf[4], asc[4], ash[4], a[4];
        unpack4(*reinterpret_cast<const uint2*>(st + (hasx ? L.oX : L.oZo) + (row * KW + kl) * 2), xf);
        *reinterpret_cast<float4*>(asc) = *reinterpret_cast<const float4*>(pa + kl);
        *reinterpret_cast<float4*>(ash) = *reinterpret_cast<const float4*>(pa + KW + kl);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = ok ? act_fwd(xf[i] * asc[i] + ash[i], aact, aslope) : 0.f;
        uint2 ap;
        ap.x = pack_bf16x2(a[0], a[1]); ap.y = pack_bf16x2(a[2], a[3]);
        *reinterpret_cast<uint2*>(at + row * KPAD + kl) = ap;
      }
    }
    // ---- dz: MFMA B fragments (row m = h*16+li, 8 consecutive n) and the row-major copy for the transpose loads
    bf16x8_t bz[NH][NS];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      bf16x8_t afr;
      if constexpr (RZ) {   // this lane's 8 input channels of row m: B operand of z^T = W a^T (zero beyond the KW columns)
        uint4 u = make_uint4(0, 0, 0, 0);
        if (lq * 8 < KW) u = *reinterpret_cast<const uint4*>(at + (h * 16 + li) * KPAD + lq * 8);
        afr = __builtin_bit_cast(bf16x8_t, u);
      }
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const int row = h * 16 + li, n = ns * 32 + lq * 8;
        const bool ok = n < N && m0 + row < M;
        const int nc = n < N ? n : N - 8;
        float gf[8], zf[8], p0[8], p1[8], p2[8], v[8];
        Raw8<bf16_t> gr;
        gr.u = *reinterpret_cast<const uint4*>(st + (row * N + nc) * 2);
        cvt8(gr, gf);
        if constexpr (RZ) {
          const f32x4_t z0 = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, wz[(ns * 2 + 0) * 64 + lane]), afr, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
          const f32x4_t z1 = SPB_MFMA16(__builtin_bit_cast(bf16x8_t, wz[(ns * 2 + 1) * 64 + lane]), afr, ((f32x4_t){0.f, 0.f, 0.f, 0.f}));
#pragma unroll
          for (int i = 0; i < 4; ++i) { zf[i] = z0[i]; zf[4 + i] = z1[i]; }
        } else {
          Raw8<bf16_t> zr;
          zr.u = *reinterpret_cast<const uint4*>(st + L.oZ + (row * N + nc) * 2);
          cvt8(zr, zf);
        }
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
          *reinterpret_cast<float4*>(p0 + j) = *reinterpret_cast<const float4*>(pdz + n + j);
          *reinterpret_cast<float4*>(p1 + j) = *reinterpret_cast<const float4*>(pdz + NP + n + j);
          *reinterpret_cast<float4*>(p2 + j) = *reinterpret_cast<const float4*>(pdz + 2 * NP + n + j);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok ? gf[j] * p0[j] + zf[j] * p1[j] + p2[j] : 0.f;
        uint4 u;
        u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
        u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
        bz[h][ns] = __builtin_bit_cast(bf16x8_t, u);
        if (L.oDzt >= 0 || n < N) *reinterpret_cast<uint4*>(dzt + row * LDZ + n) = u;
      }
    }
    // ---- input gradient, one 16-row half at a time
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x4_t acc[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        acc[kb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[kb] = SPB_MFMA16(Wf[kb][ns], bz[h][ns], acc[kb]);
      }
      const int row = h * 16 + li;
      const long long m = m0 + row;
      uint2 oo[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int kl = lq * 4 * KB + kb * 4, k = k0 + kl;    // lane (li = row, lq): channels lq*4*KB + kb*4 + i
        const bool ok = m < M && k < K;
        float zf[4], sc[4], sh[4], v[4];
        const uint2 zraw = *reinterpret_cast<const uint2*>(st + L.oZo + (row * KW + kl) * 2);
        unpack4(zraw, zf);
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(pep + kl);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(pep + KW + kl);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[kb][i];
        if (hasr) {
          float rf[4];
          unpack4(*reinterpret_cast<const uint2*>(st + L.oR + (row * KW + kl) * 2), rf);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += rf[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= act_grad(zf[i] * sc[i] + sh[i], eact, eslope);
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        oo[kb] = o;
        unpack4(o, v);                                     // the rounded values are what the sums must see
        if (ok) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { s1[kb][i] += v[i]; s2[kb][i] += v[i] * zf[i]; }
        }
      }
      if (m < M) {   // one contiguous run of 4*KB channels per lane (16-byte aligned when KB is even)
        bf16_t* dst = Y + (size_t)m * K + k0 + lq * 4 * KB;
        const int kbase = k0 + lq * 4 * KB;
        if constexpr ((KB & 1) == 0) {
#pragma unroll
          for (int kb = 0; kb < KB; kb += 2) {
            if (kbase + kb * 4 + 4 < K) *reinterpret_cast<uint4*>(dst + kb * 4) = make_uint4(oo[kb].x, oo[kb].y, oo[kb + 1].x, oo[kb + 1].y);
            else if (kbase + kb * 4 < K) *reinterpret_cast<uint2*>(dst + kb * 4) = oo[kb];
          }
        } else {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
            if (kbase + kb * 4 < K) *reinterpret_cast<uint2*>(dst + kb * 4) = oo[kb];
        }
      }
    }
    // ---- weight gradient: dW[n,k] += sum over the chunk's rows
    asm volatile("" ::: "memory");   // the dz / a tile stores above stay ahead of the transpose loads
    bf16x8_t bf[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) bf[kb] = tr_frag<CH>(at, KPAD, kb * 16, li, lq);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const bf16x8_t af = tr_frag<CH>(dzt, LDZ, nb * 16, li, lq);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) dw[nb][kb] = SPB_MFMA16(af, bf[kb], dw[nb][kb]);
    }
  }
  wait_vmcnt<0>();

  // ---- reductions over the block: weight gradient (C layout: column k = li, rows n = lq*4+i) and the BN sums.
  //      ONE [NB][KB][256] tile: the waves add themselves in turn (nw barriers) -- nw tiles would be 147 KB for 8 waves
  __syncthreads();
  float* red = reinterpret_cast<float*>(wz + (RZ ? NS * 2 * 64 : 0));          // the wave regions are idle now
  for (int w = 0; w < nw; ++w) {
    if (wave == w) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float* q = red + (nb * KB + kb) * 256 + (lq * 4 + i) * 16 + li;
            *q = w == 0 ? dw[nb][kb][i] : *q + dw[nb][kb][i];
          }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < NB * KB * 256; e += nthr) {
    const int blk = e >> 8, r = (e >> 4) & 15, cidx = e & 15;
    const int nb = blk / KB, kb = blk % KB;
    const int n = nb * 16 + r, k = k0 + kb * 16 + cidx;
    if (n < N && k < K) SPB_ATOMIC_W(g.dW + (size_t)n * K + k, red[e]);
  }
  __syncthreads();
  // BN-backward sums of gin: lanes of a 16-lane row hold different m for the same 4 k -> butterfly, then waves in LDS
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = row16_sum(s1[kb][i]), b = row16_sum(s2[kb][i]);
      if (li == 0) {
        red[(wave * 2 + 0) * KW + lq * 4 * KB + kb * 4 + i] = a;
        red[(wave * 2 + 1) * KW + lq * 4 * KB + kb * 4 + i] = b;
      }
    }
  __syncthreads();
  for (int e = threadIdx.x; e < KW; e += nthr) {
    const int k = k0 + e;
    if (k < K) {
      float a = 0.f, b = 0.f;
      for (int w = 0; w < nw; ++w) { a += red[(w * 2 + 0) * KW + e]; b += red[(w * 2 + 1) * KW + e]; }
      const float mu = pep[2 * KW + e], is = pep[3 * KW + e];
      float* dst = g.osums + (size_t)(blockIdx.x % g.oR) * 2 * K;
      atomicAdd(dst + k, a);
      atomicAdd(dst + K + k, is * (b - mu * a));   // sum g*xhat from sum g*z (see dwconv_rows.hip)
    }
  }
}

static int g_pwb_ch = 16;      // rows per chunk (16 | 32) -- spb_debug_set_pwb(ch, rz, max_waves)
static int g_pwb_rz = 1;       // expand layers (K <= 32): recompute z instead of reading it
static int g_pwb_maxw = 8;     // waves per workgroup

template <int NB, int KB, int CH, bool RZ>
int pwb_launch2(const spb_pwbwd_args_t& g, hipStream_t stream) {
  constexpr int NS = (NB + 1) / 2, NP = NS * 32, NPAD = NP + 8, KW = KB * 16, KPAD = KW + 8;
  const bool hasx = g.X != g.Zout, hasr = g.res != nullptr;
  const PwbLay L = pwb_lay(CH, g.N, KW, NPAD, KPAD, hasx, hasr, !RZ);
  const size_t tables = (size_t)(3 * NP + 6 * KW) * sizeof(float) + (RZ ? (size_t)NS * 2 * 64 * 16 : 0);
  const size_t red = (size_t)NB * KB * 256 * sizeof(float);
  // waves per workgroup x workgroups per CU: whatever holds the most waves on a CU (LDS, and the register file: 8 waves per CU
  // at up to 256 registers, 12 at up to 168, 16 at up to 128)
  static int reg_waves = 0;
  if (!reg_waves) {
    hipFuncAttributes fa;
    reg_waves = 8;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&pwb_kernel<NB, KB, CH, RZ>)) == hipSuccess)
      reg_waves = fa.numRegs <= 128 ? 16 : (fa.numRegs <= 168 ? 12 : 8);
  }
  int nw = g_pwb_maxw < 2 ? 2 : (g_pwb_maxw > 8 ? 8 : g_pwb_maxw), per_cu = 1;
  size_t lds = 0;
  int best = 0;
  for (int cand = nw; cand >= 2; --cand) {
    size_t l = tables + (size_t)cand * L.wave_bytes;
    if (tables + red > l) l = tables + red;     // the final reductions reuse the wave regions
    if (l > 160 * 1024) continue;
    int pc = (int)((160 * 1024) / l);
    if (pc * cand > reg_waves) pc = reg_waves / cand;
    if (pc < 1) pc = 1;
    if (pc * cand > best) { best = pc * cand; nw = cand; per_cu = pc; lds = l; }
  }
  if (best == 0) return SPB_E_UNSUPPORTED;
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pwb_kernel<NB, KB, CH, RZ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  const long long nchunks = ((long long)g.M + CH - 1) / CH;
  long long blocks = (nchunks + nw - 1) / nw;
  const long long cap = 256LL * per_cu;
  if (blocks > cap) blocks = cap;
  const int nsplit = (g.K + KW - 1) / KW;
  hipLaunchKernelGGL((pwb_kernel<NB, KB, CH, RZ>), dim3((unsigned)blocks, (unsigned)nsplit), dim3(64 * nw), lds, stream, g, nchunks);
  return 0;
}
template <int NB, int KB>
int pwb_launch(const spb_pwbwd_args_t& g, hipStream_t stream) {
  constexpr bool can_rz = KB * 16 <= 32;
  if constexpr (can_rz) {
    if (g_pwb_rz && g.K <= KB * 16) return g_pwb_ch == 32 ? pwb_launch2<NB, KB, 32, true>(g, stream) : pwb_launch2<NB, KB, 16, true>(g, stream);
  }
  return g_pwb_ch == 32 ? pwb_launch2<NB, KB, 32, false>(g, stream) : pwb_launch2<NB, KB, 16, false>(g, stream);
}

}  // namespace

#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_pwb(int chunk_rows, int recompute_z, int max_waves) {
  if (chunk_rows == 16 || chunk_rows == 32) g_pwb_ch = chunk_rows;
  if (recompute_z >= 0) g_pwb_rz = recompute_z != 0;
  if (max_waves >= 2) g_pwb_maxw = max_waves;
  return 0;
}
#endif

// 0 on launch, SPB_E_UNSUPPORTED when this shape / dtype has no fused instance (the caller then uses
// spb_pwconv_gemm + spb_pwconv_wgrad)
extern "C" int spb_pwconv_bwd_fused(int dtype, const spb_pwbwd_args_t* a, spb_stream_t stream) {
  if (!a || !a->G || !a->Wt || !a->X || !a->Zout || !a->Y || !a->dW || !a->osums) return SPB_E_ARG;
  if (!a->Zn && !(g_pwb_rz && a->K <= 32)) return SPB_E_ARG;   // z may be absent only where it is recomputed (RZ: expand layers, K <= 32)
  if (a->M <= 0 || a->K <= 0 || a->N <= 0 || (a->K & 7) || (a->N & 7) || a->oR < 1) return SPB_E_SHAPE;
  if (dtype != SPB_BF16) return SPB_E_UNSUPPORTED;
  const int NB = (a->N + 15) / 16, KBt = (a->K + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  int e = SPB_E_UNSUPPORTED;
  if (NB == 1 && KBt <= 2) e = pwb_launch<1, 2>(*a, s);           // 32 -> 16 @112
  else if (NB <= 6 && NB > 2 && KBt == 1) e = pwb_launch<6, 1>(*a, s);  // 16 -> 96 @112
  else if (NB <= 2 && KBt <= 6 && KBt > 2) e = pwb_launch<2, 6>(*a, s); // 96 -> 24 @56
  else if (NB <= 9 && NB > 6 && KBt <= 2) e = pwb_launch<9, 2>(*a, s);  // 24 -> 144 @56
  else if (NB <= 2 && KBt <= 10 && KBt > 6) e = pwb_launch<2, 5>(*a, s); // 144 -> 24 @56, 144 -> 32 @28 (two K splits)
  if (e != 0) return e;
  SPB_CHECK_LAUNCH();
  return 0;
}
