// Style-transfer decoder, the ten 128 -> 128 3x3 convolutions of the residual blocks (src/styleaug/ghiasi.py:62-104; half of
// the decoder's time).  Same contract as spb_gconv (ghiasi.hip): raw NHWC bf16 output + per-(image, channel) sums, the
// producer's instance norm + style affine + ReLU applied to the input while it is staged, reflection padding.
//
// What round 3's ablation of gconv_slab_kernel (scratch/ubench_gconv.hip, 85 us per layer at B=48, 56x56) said, and what this
// kernel does about it (43 us, 1.04 PFLOP/s, matrix cores 41 % busy: profiles/r3_ghiasi_*):
//   * 31 us were the OUTPUT STORES: a lane holds 32 channels of one pixel, so every store instruction wrote 64 separate 16-byte
//     pieces 64 B apart (a quarter of 64 cache lines each).  Here the accumulators go to LDS first (the halo is dead by then)
//     and leave as whole pixels: 16 lanes x 16 B = one 256-byte pixel, 1 KB contiguous per wave instruction.  The per-channel
//     sums are taken on that pass too (packed f32 pairs, 12 v_permlane swaps per lane) instead of 64 DPP butterflies per lane.
//   * the K loop was bound by LDS traffic and barriers, not by the matrix cores.  ds_read_b128 is served in four groups of 16
//     lanes that are NOT lane-contiguous ({0-3, 12-15, 20-27}, ... : MI355X_MICROARCH.md, LDS): the halo is laid out as 16
//     CHANNEL-CHUNK PLANES [chunk][10x10 pixels] x 16 B, odd chunks a multiple of 256 B after the even ones, and a pixel fragment
//     is tile rows (r, r+4): pixel indices r*10 + c and (r+4)*10 + c differ by 40 = 8 mod 16, so the 16 pixels of a fragment
//     always fall on 16 different slots whatever the tap.  The weights do not touch LDS at all: a wave owns 32 output channels
//     of all 128 pixels and loads its two weight fragments per reduction step straight from L2 into the registers the matrix
//     cores read (pre-packed by spb_gconv_wide_pack: 1 KB contiguous per wave instruction, six steps ahead), so the reduction
//     loop has NO barrier and NO LDS store -- per step 8 ds_read_b128 + 2 global loads feed 16 MFMAs.
//   * workgroups are persistent (two per CU); the next group's halo and coefficient loads are issued in the epilogue, the weight
//     stream never stops, and the workgroup barriers order LDS only (lds_barrier: __syncthreads() also drains vmcnt, i.e. waits
//     for the prefetched halo and for the output stores).
// Registers are the budget that decides everything here: 96 B of scratch per lane (a double-buffered pixel-fragment array) made
// the kernel 25 % SLOWER -- a scratch reload is a vmcnt-counted load, and vmcnt retires in order, so each reload waited for the
// whole weight prefetch in front of it.  Two 8x8 tiles per group, 57 KB of LDS, 220 VGPRs, no scratch.
// Measured and dropped: tile-granular work split (5 | 4 tiles per workgroup instead of 6 | 4: single-tile units pay the whole
// staging/epilogue cost, 46 us), staggered entry into the weight cycle and a delayed second workgroup per CU (no effect), pixel
// fragments double-buffered in registers (GW_BF2: no effect once it fits without scratch), three workgroups per CU (GW_OCC3: a first
// 168-register build kept 104 B of scratch between the halo loads -- commit 12 us, 55 us per layer; scratch-free with the weight
// prefetch 3 steps deep it is 43.0 against 45.1 us in the microbenchmark and nothing in the decoder itself: kept as an option).
#include "common.h"

#ifndef GW_OCC3
#define GW_OCC3 0   // 1: three workgroups per CU: 168 registers (weight prefetch 3 steps deep), 51.7 KB of LDS (unstaggered planes,
#endif              //    the reduction scratch inside the staging area)
#ifndef GW_BF2
#define GW_BF2 0   // 1: pixel fragments double-buffered in registers (one reduction step ahead)
#endif
#ifndef GWABL
#define GWABL 0   // ablation bits for scratch/ubench_gconv.hip: 1 no K loop, 2 no halo staging, 4 no slab traffic, 8 no MFMA, 64 no output
#endif

#ifdef GW_TS
__device__ unsigned long long* g_gw_ts;     // scratch/ubench_gconv.hip: [workgroup][group 0..3][5] wall clock (100 MHz)
#define GW_STAMP(i) do { if (t == 0 && g_gw_ts && gi < 4) g_gw_ts[(blockIdx.x * 4 + gi) * 5 + (i)] = wall_clock64(); } while (0)
#else
#define GW_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int GW_PLANE = GW_OCC3 ? 1600 : 1632;   // bytes per chunk plane: 100 pixels x 16 B (+ 32: staggers the 8 planes of a parity over the store banks)
constexpr int GW_ODD = 8 * GW_PLANE;     // 13056 = 51 x 256: where the odd chunks start
constexpr int GW_TILE = 2 * GW_ODD;      // one 8x8 tile's 10x10 halo, 128 channels
constexpr int GW_OPITCH = 272;           // output staging: 128 channels of a pixel + 16 B
constexpr int GW_PF = GW_OCC3 ? 3 : 6;                 // reduction steps of weight fragments in flight per wave (global -> registers, 2 KB each)
constexpr int GW_NSTEP = 36;             // 9 taps x 4 chunks of 32 input channels
constexpr int GW_HV = 2 * 100 * 16;      // halo vectors (8 channels) per workgroup
constexpr int GW_HU = (GW_HV + 255) / 256;

// One swap serves two reductions: v_permlane16_swap exchanges the odd 16-lane rows of a with the even rows of b, so a' + b' holds
// a's row-pair sums in the even rows and b's in the odd rows; v_permlane32_swap does the same with the wave's halves.  Sixteen
// per-lane values reduced over the four rows cost 12 swaps (was 32), and the totals end up spread over the rows.
__device__ __forceinline__ float gw_swap16_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float gw_swap32_add(float a, float b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// A zero the compiler cannot see through: index arithmetic that depends on it is recomputed where it is used instead of being
// hoisted out of the group loop -- hoisted, the per-vector halo offsets (13 per thread) were spilled to scratch in the three-per-CU
// build, and a scratch reload between two halo loads waits (vmcnt retires in order) for every load issued before it.
__device__ __forceinline__ int gw_opaque_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }

__device__ __forceinline__ int gw_reflect(int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * n - 2 - i : i; }

struct GwGroup { int b, oy0[2], ox0[2]; bool tvalid[2]; };

__device__ __forceinline__ GwGroup gw_group(int gid, int gpi, int tpi, int tiles_x) {
  GwGroup q;
  q.b = gid / gpi;
  const int grp = gid - q.b * gpi;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    int tr = grp * 2 + p;
    q.tvalid[p] = tr < tpi;
    tr = q.tvalid[p] ? tr : tpi - 1;
    q.oy0[p] = (tr / tiles_x) * 8; q.ox0[p] = (tr % tiles_x) * 8;
  }
  return q;
}

// Persistent: gridDim.x workgroups (two per CU) walk the tile groups of "their" XCD (workgroup k runs on XCD k % 8, which owns a
// contiguous range of groups, so neighbouring tiles share one L2).  Per group: [halo commit] [36-step K loop] [next group's halo
// and coefficient loads issued] [epilogue].  A one-shot workgroup per group (the first version: 54 us) kept its CU slot until its
// output stores were acknowledged and restarted the weight stream from an empty pipe; here the stores drain under the next
// group's work, the halo round trip hides behind the epilogue, and the weight ring never stops (36 steps per group, PF | 36: the
// prefetch simply wraps to slab 0).
__global__ __launch_bounds__(256, GW_OCC3 ? 3 : 2) void gconv_wide_kernel(const spb_gconv_args_t g, int ngroups, int rotate, int delay) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;                             // [2 tiles][GW_TILE]; in the epilogue: output staging [128 pixels][GW_OPITCH]
  // [4 waves][128][2]; GW_OCC3: behind the output staging inside the (then idle) halo area, fenced by one more barrier per group
  float* red = reinterpret_cast<float*>(GW_OCC3 ? smem + 128 * GW_OPITCH : smem + 2 * GW_TILE);
  float* biasl = reinterpret_cast<float*>(smem + 2 * GW_TILE) + (GW_OCC3 ? 0 : 4 * 128 * 2);   // [128]
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, lq = lane >> 4, wave = t >> 6;
  const int H = g.Hin, W = g.Win;
  const int tiles_x = W >> 3, tpi = tiles_x * (H >> 3), gpi = (tpi + 1) >> 1;
  const int xcd = blockIdx.x & 7, nw8 = gridDim.x >> 3;
  const int gq = ngroups >> 3, gr = ngroups & 7;
  const int cnt = gq + (xcd < gr ? 1 : 0), base = xcd * gq + (xcd < gr ? xcd : gr);
  int l = blockIdx.x >> 3;
  if (l >= cnt) return;
  const bf16_t* Wg = reinterpret_cast<const bf16_t*>(g.W);
  const bf16_t* X = reinterpret_cast<const bf16_t*>(g.X);
  bf16_t* Y = reinterpret_cast<bf16_t*>(g.Y);
  // weight fragments of this wave (output channels 32 w .. 32 w + 31): straight from L2 into the registers the matrix cores
  // read, GW_PF steps ahead; 1 KB contiguous per wave instruction
  const bf16_t* Wl = Wg + ((size_t)wave * 128 + lane) * 8;
  bf16x8_t areg[GW_PF * 2];   // flat, constant indices only (stays in registers)
  // Every workgroup streams the SAME 295 KB of weights; started together they would all ask the same L2 channel for the same
  // lines at the same time.  The 36 reduction steps are a cycle: workgroup k enters it at step 6 * (k' % 6) and never leaves
  // it (the next group's steps follow on), so at any moment the workgroups of an XCD are spread over the whole tensor.
  const int rot = rotate ? GW_PF * ((blockIdx.x >> 3) % (GW_NSTEP / GW_PF)) : 0;
#pragma unroll
  for (int d = 0; d < GW_PF; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      areg[d * 2 + j] = *reinterpret_cast<const bf16x8_t*>(Wl + (size_t)(rot + d) * 4096 + j * 512);
  int sbase = rot;                                   // first step of the next 6 (36 per group: back at `rot` after each group)

  // halo: thread t owns channel chunk cv = t % 16 of pixels (t / 16 + 16 u) of the two 10x10 halos
  const int cv = t & 15;
  Raw8<bf16_t> r[GW_HU];
  float4 cfr[4];
  auto issue_halo = [&](const GwGroup& q) {
    if (g.coef) {
      const float4* cp = reinterpret_cast<const float4*>(g.coef + ((size_t)q.b * 128 + cv * 8) * 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) cfr[j] = cp[j];
    }
    const char* Xb = reinterpret_cast<const char*>(X) + (size_t)q.b * H * W * 256;      // uniform base + 32-bit lane offsets
    const int tl = t + gw_opaque_zero();
#pragma unroll
    for (int u = 0; u < GW_HU; ++u) {
      const int i = tl + 256 * u;
      const int ic = i < GW_HV ? i : GW_HV - 1;
      const int p = ic >= 1600 ? 1 : 0;
      const int hp = (ic - p * 1600) >> 4;
      const int hy = hp / 10, hx = hp - hy * 10;
      const int sy = gw_reflect((p ? q.oy0[1] : q.oy0[0]) - 1 + hy, H), sx = gw_reflect((p ? q.ox0[1] : q.ox0[0]) - 1 + hx, W);
      r[u] = ldraw<bf16_t>(reinterpret_cast<const bf16_t*>(Xb + (unsigned)((sy * W + sx) * 256 + cv * 16)));
    }
  };
  GwGroup cur = gw_group(base + l, gpi, tpi, tiles_x);
  if (!(GWABL & 2)) issue_halo(cur);
  if (delay) {     // experiment: the second workgroup of a CU starts half a period late (its VALU phases under the first one's MFMA loop)
    // decided by wave 0 for the whole workgroup (each wave reading its own HW_ID made every workgroup "late" through the barrier):
    // the wave-slot index of wave 0 on its SIMD differs between the two resident workgroups of a CU
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (t == 0) red[0] = (float)(hwid & 0xf);
    __syncthreads();
    const int slot = (int)red[0];
    const bool late = delay > 0 ? (slot & 1) : ((blockIdx.x >> 3) & 1);
    if (late) for (int i = 0; i < (delay > 0 ? delay : -delay); ++i) __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles = 0.43 us
    __syncthreads();
  }

  const int prow = 4 * (li >> 3), pcol = li & 7;      // fragment f = tile (f / 4), rows (f % 4, f % 4 + 4)
  const char* hb = halo + (lq & 1) * GW_ODD + (lq >> 1) * GW_PLANE + (prow * 10 + pcol) * 16;
  if (t < 128) biasl[t] = g.bias ? g.bias[t] : 0.f;     // visible after the first barrier of the group loop

  int gi = 0;
  for (;; ++gi) {
    GW_STAMP(0);
    // ---- commit the halo of `cur` (loads issued one epilogue ago): x * scale + shift in packed f32 pairs, ReLU as a packed
    // signed-16-bit max on the bf16 bits; the first convolution of a residual block reads a materialised tensor (no transform)
    if (!(GWABL & 2)) {
      const int tl = t + gw_opaque_zero();
      char* hdst = halo + (cv & 1) * GW_ODD + (cv >> 1) * GW_PLANE;
      if (!g.coef && !g.relu) {
#pragma unroll
        for (int u = 0; u < GW_HU; ++u) {
          const int i = tl + 256 * u;
          if (i >= GW_HV) continue;
          const int p = i >= 1600 ? 1 : 0;
          *reinterpret_cast<uint4*>(hdst + p * GW_TILE + ((i - p * 1600) >> 4) * 16) = r[u].u;
        }
      } else {
        spb_f32x2 scp[4], shp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          scp[j] = g.coef ? (spb_f32x2){cfr[j].x, cfr[j].z} : (spb_f32x2){1.f, 1.f};
          shp[j] = g.coef ? (spb_f32x2){cfr[j].y, cfr[j].w} : (spb_f32x2){0.f, 0.f};
        }
        typedef short gw_s16x2 __attribute__((ext_vector_type(2)));
        const short lo = g.relu ? (short)0 : (short)-32768;
        const gw_s16x2 floor2 = {lo, lo};
#pragma unroll
        for (int u = 0; u < GW_HU; ++u) {
          const int i = tl + 256 * u;
          if (i >= GW_HV) continue;
          const int p = i >= 1600 ? 1 : 0;
          const unsigned w4[4] = {r[u].u.x, r[u].u.y, r[u].u.z, r[u].u.w};
          unsigned o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const spb_f32x2 x = {bf2f((bf16_t)(w4[j] & 0xffffu)), bf2f((bf16_t)(w4[j] >> 16))};
            const spb_f32x2 y = x * scp[j] + shp[j];
            const unsigned pk = pack_bf16x2(y[0], y[1]);
            o4[j] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(gw_s16x2, pk), floor2));
          }
          *reinterpret_cast<uint4*>(hdst + p * GW_TILE + ((i - p * 1600) >> 4) * 16) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        }
      }
    }
    lds_barrier();
    GW_STAMP(1);

    // ---- K loop, no barrier and no LDS store in it: wave w owns output channels 32 w .. 32 w + 31 of all 128 pixels (8 pixel
    // fragments from LDS x 2 weight fragments from its registers = 16 MFMAs per 8 ds_read_b128)
    f32x4_t acc[8][2];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
      const float4 bv = *reinterpret_cast<const float4*>(biasl + wave * 32 + lq * 8 + cf * 4);
#pragma unroll
      for (int f = 0; f < 8; ++f) acc[f][cf] = (f32x4_t){bv.x, bv.y, bv.z, bv.w};
    }
    const int nsteps = (GWABL & 1) ? 0 : GW_NSTEP;
    // LDS offset of reduction step s = tap * 4 + cc within a fragment's plane set: chunk pair 2 cc, pixel (ky, kx)
    auto step_off = [](int s) { const int tap = s >> 2, ky = (tap * 11) >> 5; return (s & 3) * (2 * GW_PLANE) + (ky * 10 + tap - 3 * ky) * 16; };
#if GW_BF2
    // pixel fragments one step ahead, in their own registers (left alone the compiler fetched two at a time into the same
    // registers: an LDS round trip per four MFMAs)
    bf16x8_t bfr[2][8];
    {
      const char* hs = hb + step_off(sbase);
#pragma unroll
      for (int f = 0; f < 8; ++f) bfr[0][f] = *reinterpret_cast<const bf16x8_t*>(hs + (f >> 2) * GW_TILE + (f & 3) * 160);
    }
#else
    bf16x8_t bfr[1][8];
#endif
    for (int i0 = 0; i0 < nsteps; i0 += GW_PF) {
#pragma unroll
      for (int d = 0; d < GW_PF; ++d) {
        const int s = sbase + d;
#if GW_BF2
        const char* hs = hb + step_off(s + 1 == GW_NSTEP ? 0 : s + 1);     // after the group's last step: unused
        const int cur = d & 1, nxt = (d + 1) & 1;
#else
        const char* hs = hb + step_off(s);
        const int cur = 0, nxt = 0;
#endif
#pragma unroll
        for (int f = 0; f < 8; ++f) bfr[nxt][f] = *reinterpret_cast<const bf16x8_t*>(hs + (f >> 2) * GW_TILE + (f & 3) * 160);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < 8; ++f) {
#pragma unroll
          for (int cf = 0; cf < 2; ++cf) {
            if (GWABL & 8) acc[f][cf][0] += (float)areg[d * 2 + cf][0] * (float)bfr[cur][f][0];
            else acc[f][cf] = SPB_MFMA16(areg[d * 2 + cf], bfr[cur][f], acc[f][cf]);
          }
        }
        // no branch around the refill: the step index wraps (the weight stream of the next group starts here)
        if (!(GWABL & 4)) {
          const int sf = s + GW_PF < GW_NSTEP ? s + GW_PF : s + GW_PF - GW_NSTEP;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            areg[d * 2 + j] = *reinterpret_cast<const bf16x8_t*>(Wl + (size_t)sf * 4096 + j * 512);
        }
      }
      sbase = sbase + GW_PF < GW_NSTEP ? sbase + GW_PF : 0;
    }
    GW_STAMP(2);
    lds_barrier();   // every wave is done with the halo
    GW_STAMP(3);

    // ---- epilogue 1: accumulators -> LDS as whole pixels: a lane holds channels 32 w + 8 lq .. + 7 of its 8 pixels
    char* ost = halo;
    if (!(GWABL & 64)) {
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        uint4 o;
        o.x = pack_bf16x2(acc[f][0][0], acc[f][0][1]); o.y = pack_bf16x2(acc[f][0][2], acc[f][0][3]);
        o.z = pack_bf16x2(acc[f][1][0], acc[f][1][1]); o.w = pack_bf16x2(acc[f][1][2], acc[f][1][3]);
        *reinterpret_cast<uint4*>(ost + ((f >> 2) * 64 + ((f & 3) + prow) * 8 + pcol) * GW_OPITCH + wave * 64 + lq * 16) = o;
      }
    }
    // ---- the next group's halo: in flight during the rest of the epilogue (clamped, not branched: the last group re-reads its
    // own).  After the accumulators have left their registers: 13 vectors + coefficients per thread.
    const int ln = l + nw8;
    const bool more = ln < cnt;
    const GwGroup nxt = gw_group(base + (more ? ln : l), gpi, tpi, tiles_x);
    if (!(GWABL & 2)) issue_halo(nxt);
    lds_barrier();
    // ---- epilogue 2: pixel (t/16 + 16 u), chunk cv: 16-byte stores, 1 KB contiguous per wave instruction; sums of the stored
    // values in packed f32 pairs.  Addresses are a uniform base + a 32-bit lane offset (no 64-bit vector arithmetic).
    spb_f32x2 s1p[4], s2p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1p[j] = (spb_f32x2){0.f, 0.f}; s2p[j] = (spb_f32x2){0.f, 0.f}; }
    if (!(GWABL & 64)) {
      char* Yb = reinterpret_cast<char*>(Y) + (size_t)cur.b * H * W * g.ldc * 2;        // uniform
      const unsigned rstride = (unsigned)(2 * W * g.ldc * 2);                          // two image rows per u
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (!cur.tvalid[p]) continue;
        unsigned yoff = (unsigned)(((cur.oy0[p] + (t >> 7)) * W + cur.ox0[p] + ((t >> 4) & 7)) * g.ldc * 2 + cv * 16);
        const char* src = ost + (p * 64 + (t >> 4)) * GW_OPITCH + cv * 16;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint4 o = *reinterpret_cast<const uint4*>(src + u * 16 * GW_OPITCH);
          *reinterpret_cast<uint4*>(Yb + yoff) = o;
          yoff += rstride;
          const unsigned w4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float xlo, xhi; spb_unpack2(w4[j], xlo, xhi);
            const spb_f32x2 x = {xlo, xhi};
            s1p[j] += x;
            s2p[j] += x * x;
          }
        }
      }
    }
    if (g.stats) {
      // V[0..7] = sums of channels cv*8 + 0..7, V[8..15] = sums of squares; after the two swap levels row k of Q[i] holds V[4 i + k]
      float V[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) { V[2 * j] = s1p[j][0]; V[2 * j + 1] = s1p[j][1]; V[8 + 2 * j] = s2p[j][0]; V[9 + 2 * j] = s2p[j][1]; }
      float P[8], Q[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) P[i] = gw_swap16_add(V[2 * i], V[2 * i + 1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) Q[i] = gw_swap32_add(P[2 * i], P[2 * i + 1]);
      float* rw = red + (wave * 128 + cv * 8 + lq) * 2;       // lane (row lq): channels cv*8 + lq and cv*8 + 4 + lq
      *reinterpret_cast<float2*>(rw) = make_float2(Q[0], Q[2]);
      *reinterpret_cast<float2*>(rw + 8) = make_float2(Q[1], Q[3]);
    }
    lds_barrier();   // the staging area is read out (the next commit overwrites it); red is complete
    if (g.stats) atomicAdd(g.stats + (size_t)cur.b * 256 + t, red[t] + red[256 + t] + red[512 + t] + red[768 + t]);
    if (GW_OCC3) lds_barrier();     // red sits where the next commit writes
    GW_STAMP(4);
    if (!more) break;
    l = ln;
    cur = nxt;
  }
}

// W [128][9][128] bf16 (the spb_gconv layout) -> packed[step = tap*4 + cc][wave w][cf][lane = lq*16 + li][8]: MFMA row li of
// fragment (w, cf) is output channel 32 w + (li / 4) * 8 + cf * 4 + li % 4 (a lane of the transposed product then holds 8
// consecutive channels over its two fragments), k = cc * 32 + lq * 8 + j
__global__ void gconv_wide_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= GW_NSTEP * 512) return;
  const int s = idx >> 9, gr = idx & 511;
  const int wv = gr >> 7, cf = (gr >> 6) & 1, lq = (gr >> 4) & 3, li = gr & 15;
  const int co = wv * 32 + (li >> 2) * 8 + cf * 4 + (li & 3), tap = s >> 2, cc = s & 3;
  *reinterpret_cast<uint4*>(out + (size_t)idx * 8) =
      *reinterpret_cast<const uint4*>(w + ((size_t)co * 9 + tap) * 128 + cc * 32 + lq * 8);
}

}  // namespace

extern "C" int spb_gconv_wide_pack(const void* w, void* packed, spb_stream_t stream) {
  if (!w || !packed) return SPB_E_ARG;
  hipLaunchKernelGGL(gconv_wide_pack_kernel, dim3((GW_NSTEP * 512 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w, (bf16_t*)packed);
  SPB_CHECK_LAUNCH();
  return 0;
}

static int g_gw_delay = 0;
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_wide_delay(int n) { g_gw_delay = n; return 0; }
#endif
static int g_gw_rot = 1;     // workgroups enter the weight cycle at staggered steps
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_wide_rotate(int on) { g_gw_rot = on; return 0; }
#endif
static int g_gw_wgs = GW_OCC3 ? 768 : 512;   // persistent workgroups: two (three) per CU
#ifdef SPB_TUNING   // tuning build only (libspb_hip_tune.so, include/spb_hip_tuning.h): the product library has no knob
extern "C" int spb_debug_set_gconv_wide_wgs(int n) { g_gw_wgs = n < 8 ? 8 : (n & ~7); return 0; }
#endif

extern "C" int spb_gconv_wide(int dtype, const spb_gconv_args_t* a, spb_stream_t stream) {
  if (!a || !a->X || !a->W || !a->Y) return SPB_E_ARG;
  if (dtype != SPB_BF16) return SPB_E_UNSUPPORTED;
  if (a->B <= 0 || a->Cin != 128 || a->Cout != 128 || a->KH != 3 || a->stride != 1 || a->upsample != 1) return SPB_E_SHAPE;
  if ((a->Hin & 7) || (a->Win & 7) || a->Hin < 8 || a->Win < 8 || a->ldc < 128 || (a->ldc & 7)) return SPB_E_SHAPE;
  const int tpi = (a->Hin >> 3) * (a->Win >> 3), gpi = (tpi + 1) >> 1;
  const int ngroups = a->B * gpi;
  const size_t lds = 2 * GW_TILE + ((GW_OCC3 ? 0 : 4 * 128 * 2) + 128) * sizeof(float);
  const int nwg = ngroups < g_gw_wgs ? ((ngroups + 7) & ~7) : g_gw_wgs;   // a multiple of 8: one share per XCD
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipLaunchKernelGGL(gconv_wide_kernel, dim3((unsigned)nwg), dim3(256), lds, (hipStream_t)stream, *a, ngroups, g_gw_rot, g_gw_delay);
  SPB_CHECK_LAUNCH();
  return 0;
}
