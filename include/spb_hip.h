/* C-ABI of libspb_hip.so -- the MI355X (gfx950) kernels behind the KRN / DANN / SPN / styleaug hot path of
 * tpark94/speedplusbaseline.
 *
 * The reference has no FFI for this path: its hot loop is ordinary torch.nn modules dispatched to cuDNN/cuBLAS
 * (SURVEY.md F1, 8(b)).  Each entry below therefore names the torch call site it replaces (reference file:line).
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates; this library never allocates or frees
 *     user-visible memory; network plans take a caller-provided workspace);
 *   - `dtype` selects the activation storage/compute type: SPB_F32 (exact f32 MFMA, parity mode) or SPB_BF16
 *     (bf16 storage, f32 accumulation).  Parameters, gradients, BN statistics and losses are always f32;
 *   - activations are NHWC ([B,H,W,C], C fastest); a 1x1 convolution is the row-major GEMM [M=B*H*W, C];
 *   - every call only ENQUEUES work on `stream` (asynchronous, graph-capturable) and returns 0 or a negative
 *     SPB_E_* / positive hipError_t code.  Nothing throws across the boundary.
 */
#ifndef SPB_HIP_H
#define SPB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spb_stream_t; /* hipStream_t */

enum { SPB_F32 = 0, SPB_BF16 = 1 };
enum { SPB_ACT_NONE = 0, SPB_ACT_RELU = 1, SPB_ACT_RELU6 = 2, SPB_ACT_LEAKY = 3 };
enum { SPB_E_ARG = -1, SPB_E_SHAPE = -2, SPB_E_STATE = -3, SPB_E_UNSUPPORTED = -4,
       SPB_E_TIMEOUT = -5 /* a stream-fork gate gave up waiting (SPB_FORK_TIMEOUT_S): the object is poisoned, every later call fails */ };

/* A BatchNorm2d (training or eval) between a producer convolution and its consumers.  The producer's epilogue
 * accumulates per-channel batch sums; consumers derive the affine from them in their load prologue.
 * Replaces nn.BatchNorm2d + ReLU6/ReLU/LeakyReLU at park2019.py:48-53,65-66 and torchvision MobileNetV2 blocks. */
typedef struct spb_bnref {
  const float* sums;  /* training: [R][2][C] = sum(z), sum(z^2) replicas;  eval (moments=1): [2][C] = mean, var */
  const float* gamma; /* [C]; NULL => identity (tensor already normalised/activated) */
  const float* beta;  /* [C] */
  const float* bsums; /* backward only: [R][2][C] = sum(g), sum(g*xhat) */
  float inv_n;        /* 1 / (B*H*W) */
  float eps;
  float slope;        /* LeakyReLU negative slope */
  int C;
  int R;              /* number of replicas of sums/bsums (spreads atomic contention) */
  int act;            /* SPB_ACT_* applied after the affine */
  int moments;        /* 1: sums holds mean/var directly */
} spb_bnref_t;

/* ---- pointwise (1x1) convolution = GEMM ------------------------------------------------------------------
 * nn.Conv2d(k=1) forward / input-gradient / weight-gradient: park2019.py:51,64; revgrad.py:76; torchvision
 * MobileNetV2 expand/project convs (park2019.py:107-108).                                                      */
typedef struct spb_gemm_args {
  const void* A;     /* [M,K] input rows (fwd: raw z of the producer or a materialised tensor; bwd: g) */
  const void* A2;    /* bwd prologue only: z with the shape of A (for xhat); NULL otherwise */
  const void* Bw;    /* [N,K] weights, K contiguous (dgrad passes the transposed copy) */
  void* Y;           /* [M,N] */
  const void* res;   /* optional [M,N] tensor added before the output-side mask (residual gradient) */
  const void* Zout;  /* epi_mode 2: raw z [M,N] of the tensor whose gradient Y is */
  const float* bias; /* epi_mode 0: optional [N] */
  float* osums;      /* epi_mode 1: [oR][2][N] sum(y), sum(y^2);  epi_mode 2: sum(g), sum(g*xhat) */
  spb_bnref_t pro;   /* BN/activation applied to A while loading (pro_mode 1) or BN-backward (pro_mode 2) */
  spb_bnref_t epi;   /* epi_mode 2: BN/activation of the output-side tensor */
  int M, K, N;
  int pro_mode;      /* 0: a = A (plain operand; `pro` must still be a valid identity reference);  1: a = act(bn(A));  2: a = bn_backward(g=A, z=A2);  3: a = bn(A) + bn2(A2), see pro2 / Ymat below */
  int epi_mode;      /* 0: y = out_act(acc*out_scale + bias);  1: y = acc, accumulate batch sums;  2: g = (acc+res)*act'(bn(Zout)) */
  int out_act;       /* epi_mode 0 */
  int oR;
  float out_scale;   /* epi_mode 0: y = out_act(acc*out_scale + bias); callers pass 1 for a plain product */
  int lda, ldc;      /* row strides (elements) of A/A2 and of Y/res/Zout; 0 = dense (K and N).  Lets one group of a grouped
                        convolution run on a column slab of the im2col / output matrices (SPN conv2, conv4, conv5) */
  void* stop_event;  /* host side only: optional hipEvent_t completed by THIS launch (attached to its dispatch packet), so
                        another stream can wait for it without an event-record packet in the launch stream */
  /* pro_mode 3, the residual join of an inverted-residual block folded into the next block's expand convolution
   * (torchvision InvertedResidual `x + self.conv(x)`): a = bn(A) + bn2(A2), no activation on either side (pro.act and
   * pro2.act must be SPB_ACT_NONE; pro2.gamma == NULL: A2 is a materialised tensor).  The workgroups of the first column
   * tile also write a to Ymat [M,K] (row stride lda) -- the block output later consumers read. */
  spb_bnref_t pro2;
  void* Ymat;
} spb_gemm_args_t;
int spb_pwconv_gemm(int dtype, const spb_gemm_args_t* args, spb_stream_t stream);
/* Y == NULL with epi_mode 1 (16-bit storage, K <= 32, N in {48, 96, 144, 192}): STATISTICS ONLY -- the per-channel sums of the f32 product
 * are accumulated into osums and the product is not stored.  What is left of an expand convolution whose output the depthwise kernels
 * recompute (spb_dw_args_t::Xe).  Ymat (optional with pro_mode 1, required with pro_mode 3) receives the convolution's operand
 * round16(act(bn(A))) [M,K] (pro_mode 3: the residual join): the tensor those kernels take as Xe with an identity `xe`. */

/* dW[N,K] += sum_m dz[m,n] * a[m,k];  dz = bn_backward(G, Zn) (pro_dz), a = act(bn(X)) (pro_a).  dW is f32. */
/* Weight-gradient partials.  A kernel that splits its reduction (the rows of the batch) over workgroups either adds every
 * workgroup's result into dW with f32 atomics (part == NULL), or -- the form the KRN plan uses -- stores it with plain stores
 * into a scratch slab `part` ([nparts][n] f32, `part_cap` floats available) and reports the reduce job that finishes it:
 * dst[i] += sum_p src[p*stride + i].  With job_out == NULL the entry launches that reduce itself on `stream`; otherwise the
 * caller collects the jobs and runs them in one spb_partial_reduce launch.  (Measured on the KRN step: the ~2 M device-scope
 * float atomics per pointwise weight-gradient launch cost the step ~0.2 ms; the partial sums are also order-deterministic.) */
typedef struct spb_red_job { const float* src; float* dst; long long stride; int n; int nparts; } spb_red_job_t;
int spb_partial_reduce(const spb_red_job_t* jobs, int njobs, spb_stream_t stream);

typedef struct spb_wgrad_args {
  const void* G;  /* [M,N] */
  const void* Zn; /* [M,N] or NULL */
  const void* X;  /* [M,K] */
  float* dW;      /* [N,K] f32, accumulated */
  spb_bnref_t pro_dz;
  spb_bnref_t pro_a;
  int M, K, N;
  int ldg, ldx;   /* row strides (elements) of G/Zn and X; 0 = dense (N and K) */
  float* part;    /* optional scratch for partial sums (see spb_red_job_t); NULL: f32 atomics into dW */
  long long part_cap;      /* floats available at `part` */
  spb_red_job_t* job_out;  /* optional: receives the reduce job (nparts == 0: none needed) instead of launching it */
} spb_wgrad_args_t;
int spb_pwconv_wgrad(int dtype, const spb_wgrad_args_t* args, spb_stream_t stream);

/* Fused backward of a pointwise convolution (input gradient + weight gradient in one pass over g and z) for the
 * wide, shallow layers (M = B*H*W >= ~32k rows, N*K small).  Same math as spb_pwconv_gemm(pro_mode 2, epi_mode 2)
 * followed by spb_pwconv_wgrad.  Returns SPB_E_UNSUPPORTED when no fused instance exists for (dtype, N, K): the caller
 * then runs the two kernels.  Replaces the autograd backward of nn.Conv2d(k=1)+BatchNorm2d(+ReLU6) in torchvision's
 * InvertedResidual (call site park2019.py:107-108). */
typedef struct {
  const void* G;      /* [M,N] g of the conv output (dL/d(bn out) * act') */
  const void* Zn;     /* [M,N] raw conv output z; may be NULL when K <= 32 (it is then recomputed from X, never read: the expand
                         convolutions whose output does not exist in memory, spb_dw_args_t::Xe) */
  const void* Wt;     /* [K,N] transposed weights in the compute dtype */
  const void* X;      /* [M,K] conv input as stored: raw z of its producer, or a materialised tensor */
  const void* Zout;   /* [M,K] raw z of the input-side BN'd tensor (== X when the input is not materialised) */
  const void* res;    /* [M,K] gradient joining the input-side tensor from elsewhere, or NULL */
  void* Y;            /* [M,K] out: g of the input-side tensor */
  float* dW;          /* [N,K] f32, accumulated */
  float* osums;       /* [oR][2][K] f32, accumulated: sum g, sum g*xhat */
  spb_bnref_t pro_dz; /* BN backward of the conv output */
  spb_bnref_t pro_a;  /* BN + activation turning X into the conv input */
  spb_bnref_t epi;    /* BN + activation of the input-side tensor (mask and xhat) */
  int M, K, N, oR;
  float* part;             /* as in spb_wgrad_args_t: scratch for the per-workgroup weight-gradient partials */
  long long part_cap;
  spb_red_job_t* job_out;
} spb_pwbwd_args_t;
int spb_pwconv_bwd_fused(int dtype, const spb_pwbwd_args_t* args, spb_stream_t stream);

/* ---- depthwise 3x3 convolution, pad 1, stride 1|2 ----------------------------------------------------------
 * nn.Conv2d(C,C,3,groups=C): park2019.py:47 and torchvision MobileNetV2 blocks.                               */
typedef struct spb_dw_args {
  const void* X;     /* fwd: [B,H,W,C] input;  dgrad/wgrad: G [B,OH,OW,C] */
  const void* X2;    /* dgrad/wgrad: z [B,OH,OW,C] of the conv output (BN backward prologue) */
  const void* Xin;   /* wgrad: raw input [B,H,W,C] */
  const float* Wd;   /* [C][3][3] f32 (the OIHW parameter itself) */
  void* Y;           /* fwd: [B,OH,OW,C];  dgrad: [B,H,W,C] */
  float* dW;         /* wgrad: [C][3][3] f32 accumulated */
  const void* res;   /* dgrad epi 2 */
  const void* Zout;  /* dgrad epi 2: raw z [B,H,W,C] of the input-side tensor */
  float* osums;
  spb_bnref_t pro;   /* fwd: BN/act of the input;  dgrad/wgrad: BN-backward of the output */
  spb_bnref_t pro_in;/* wgrad: BN/act of the input */
  spb_bnref_t epi;   /* dgrad epi 2 */
  int B, H, W, C, stride;
  int epi_mode;      /* fwd: 1;  dgrad: 0 or 2 */
  int oR;
  float* part;             /* dgrad with dW / wgrad: as in spb_wgrad_args_t */
  long long part_cap;
  spb_red_job_t* job_out;
  /* dgrad, optional: the kernel's first thread stores entry_val to this device word before anything else.  The launch runs behind
   * its stream's earlier launches, so the store tells a kernel spinning on the word (another stream of the SAME device) that all of
   * them have completed and their results are released at device scope -- a stream fork without an event on this stream (the KRN
   * plan hands its weight gradients to a side stream this way).  NULL: nothing is stored. */
  unsigned* entry_flag;
  unsigned entry_val;
  /* Expand recompute (round 6).  Xe != NULL: the depthwise layer's INPUT tensor does not exist in memory.  It is the raw output of the
   * 1x1 expand convolution in front of this layer (torchvision InvertedResidual conv[0] -> conv[1], park2019.py:107-108) and every
   * kernel rebuilds the values it needs on the matrix cores from that convolution's own input:
   *     z_in[p][c] = sum_k We[c][k] * xe[p][k]   (f32 accumulation, never rounded),   xe = round16(act(bn_xe(Xe)))
   * so the 6x-expanded 96 / 144-channel tensors of the 112x112 / 56x56 maps are neither written nor read.  Which BatchNorm / activation
   * then sits ON z_in is named exactly as without Xe: `pro` (forward), `epi` (input gradient: mask and sum g*xhat), `pro_in` (weight
   * gradient); X (forward), Zout (input gradient) and Xin (weight gradient) are ignored.  Its batch sums come from
   * spb_pwconv_gemm(Y = NULL) on the same operands.  16-bit storage, H and W >= 28, Ce in {8, 16, 24, 32}; SPB_E_UNSUPPORTED otherwise. */
  const void* Xe;     /* [B,H,W,Ce] input of the expand convolution: raw z of its producer or a materialised tensor */
  const void* We;     /* [C][Ce] expand weights in the compute dtype, Ce contiguous */
  spb_bnref_t xe;     /* BatchNorm (+ activation) turning Xe into the expand convolution's operand; gamma == NULL: identity */
  int Ce;
} spb_dw_args_t;
int spb_dwconv_fwd(int dtype, const spb_dw_args_t* args, spb_stream_t stream);
/* dgrad with args->dW != NULL also accumulates the weight gradient in the same pass (one read of g, z and the input
 * instead of two kernels); Zout / epi must then name the convolution's input tensor and its BN+activation, also when
 * epi_mode == 0.  This is the form the KRN plan uses. */
int spb_dwconv_dgrad(int dtype, const spb_dw_args_t* args, spb_stream_t stream);
int spb_dwconv_wgrad(int dtype, const spb_dw_args_t* args, spb_stream_t stream);

/* ---- stem: Conv2d(3,32,3,stride 2,pad 1,bias=False) on the NCHW f32 image (torchvision features[0]) --------
 * 16-bit modes (round 6): the forward does NOT round the image or the weights to 16 bits -- both enter the matrix cores as hi + lo pairs and the product is
 * exact to ~1e-5; only the stored output is 16-bit.  (The rounding of the network's input was what an ill-conditioned trained state amplifies: DESIGN.md section 4.) */
int spb_stem_fwd(int dtype, const float* x_nchw, const float* w /*[32][3][3][3]*/, void* y_nhwc, float* osums, int oR,
                 int B, int H, int W, spb_stream_t stream);
int spb_stem_wgrad(int dtype, const float* x_nchw, const void* G, const void* Z, const spb_bnref_t* pro_dz,
                   float* dW, int B, int H, int W, spb_stream_t stream);

/* ---- elementwise BN passes ---------------------------------------------------------------------------------
 * y = act(bn(z)) + act2(bn2(res)), optionally scattered into a wider channel-concatenated tensor and/or through
 * the RouterV2 space-to-depth "reorg" (park2019.py:74-80).                                                     */
typedef struct spb_bnapply_args {
  const void* Z;     /* [B,H,W,C] */
  const void* res;   /* optional [B,H,W,C] */
  void* Y;           /* [B,OH,OW,ldc] */
  spb_bnref_t bn;
  spb_bnref_t bn_res;
  int B, H, W, C;
  int ldc;           /* channels of the destination tensor */
  int coff;          /* first destination channel */
  int reorg;         /* 0: same H,W;  s>0: space-to-depth by s (out[b,h,w,coff+(i*s+j)*C+c] = in[b,h*s+i,w*s+j,c]) */
} spb_bnapply_args_t;
int spb_bn_apply(int dtype, const spb_bnapply_args_t* args, spb_stream_t stream);
/* g = dY(gathered with the same mapping) * act'(bn(z)); accumulates sum(g), sum(g*xhat) into osums. */
typedef struct spb_bnbwd_args {
  const void* dY;    /* [B,OH,OW,ldc] */
  const void* Z;     /* [B,H,W,C] */
  void* G;           /* [B,H,W,C] */
  float* osums;      /* [oR][2][C] */
  spb_bnref_t bn;
  int B, H, W, C, ldc, coff, reorg, oR;
} spb_bnbwd_args_t;
int spb_bn_bwd_prep(int dtype, const spb_bnbwd_args_t* args, spb_stream_t stream);

/* ---- KRN head: Conv2d(1024,2K,7) on the 7x7 map == FC over (h,w,c) + interleaved (x,y) MSE loss -----------
 * park2019.py:121,139-162.                                                                                     */
typedef struct spb_head_args {
  const void* Z;       /* [B, HW*C] raw z of the last ConvDw pointwise conv (NHWC flattened) */
  const void* Wp;      /* [Jp, HW*C] weights permuted to (h,w,c) order in the compute dtype; rows >= J are zero */
  const float* bias;   /* [J] */
  const float* target; /* [B,2,J/2] or NULL */
  float* partial;      /* S*B*Jp floats of workspace: the [B][Jp] 64-bit fixed-point accumulator of the split-K sums, then the ticket
                          word of the last-arriver epilogue -- the workspace must be ZERO before the first call; every call leaves it zero */
  float* pred;         /* [B][J] */
  float* dout;         /* [B][J] = d loss / d pred (unit upstream gradient) */
  float* scalars;      /* [3] = loss, loss_x, loss_y */
  spb_bnref_t pro;     /* BN+ReLU of Z; channel = k % C */
  int B, J, Jp, HW, C, S;
} spb_head_args_t;
int spb_head_fwd(int dtype, const spb_head_args_t* a, spb_stream_t stream);
typedef struct spb_head_bwd_args {
  const void* Z;       /* as above */
  const void* Wp;      /* as above */
  const float* dout;   /* [B][J] */
  void* G;             /* [B, HW*C]: g = dA * relu'(bn(z)) */
  float* osums;        /* [oR][2][C] */
  float* dW;           /* [J][C][HW] f32 (OIHW), accumulated */
  float* dbias;        /* [J] accumulated */
  spb_bnref_t pro;
  float gscale;        /* upstream d(loss) */
  int B, J, Jp, HW, C, oR;
  int roles;           /* 0: everything on `stream`;  1: input gradient + BN sums only;  2: weight + bias gradient only (so the
                          caller can put the latter on a side stream: only the optimizer consumes it) */
  const float* gscale_dev; /* NULL, or a device scalar multiplied onto gscale: the dynamic loss scale of the float16 recipe
                              (GradScaler.scale(loss), trainer.py:86-88) without a host read */
} spb_head_bwd_args_t;
int spb_head_bwd(int dtype, const spb_head_bwd_args_t* a, spb_stream_t stream);

/* ---- parameter / statistics maintenance (one launch for the whole network, table driven) ------------------ */
typedef struct spb_bnupd_entry {
  long long sums_off;  /* into the stats arena (floats) */
  long long bsums_off; /* into the stats arena (floats), -1 if none */
  long long rm_off;    /* running_mean offset in the buffer arena; running_var = rm_off + C */
  long long gamma_off; /* parameter offsets (gamma grad / beta grad written at the same offsets of the grad arena) */
  long long beta_off;
  int C, R, bn_index;
  float inv_n, unbias; /* unbias = n/(n-1) */
} spb_bnupd_entry_t;
/* running_mean/var momentum update + num_batches_tracked++ (nn.BatchNorm2d training side effects) */
int spb_bn_running_update(const spb_bnupd_entry_t* table_dev, int n_bn, const float* stats, float* buffers,
                          long long* nbt, float momentum, spb_stream_t stream);
/* dgamma += sum(g*xhat), dbeta += sum(g) */
int spb_bn_param_grads(const spb_bnupd_entry_t* table_dev, int n_bn, const float* stats, float* grads,
                       spb_stream_t stream);
/* the same, and every BatchNorm's batch-sum slots ([sums | backward sums], 4 R C floats from sums_off: they must be laid out back to back)
 * are zeroed afterwards: the last reader of a training step leaves clean accumulators for the next forward */
int spb_bn_param_grads_zero(const spb_bnupd_entry_t* table_dev, int n_bn, float* stats, float* grads, spb_stream_t stream);
/* eval mode: fill the stats arena slots with running mean/var so consumers see (mean,var) moments */
int spb_bn_load_running(const spb_bnupd_entry_t* table_dev, int n_bn, float* stats, const float* buffers,
                        spb_stream_t stream);

typedef struct spb_prep_entry {
  long long src_off; /* f32 parameter arena offset */
  long long dst_off; /* element offset into the compute-dtype weight arena */
  int rows, cols;    /* source viewed as [rows][cols] */
  int mode;          /* 0: copy;  1: transpose -> [cols][rows];  2: head permute [J][C][HW] -> [Jp][HW][C] */
  int aux;           /* mode 2: HW */
  int aux2;          /* mode 2: Jp */
  int tile0;         /* first tile index of this entry in the launch grid */
} spb_prep_entry_t;
int spb_weight_prep(int dtype, const spb_prep_entry_t* table_dev, int n_entries, int n_tiles, const float* params,
                    void* wcompute, spb_stream_t stream);

/* ---- optimiser: global-norm clip (trainer.py:90,97; dann.py:99) fused with the update (build.py:60-78) ----- */
int spb_grad_sqnorm(const float* grads, long long n, float* sqnorm_out /*[1], zeroed by this call*/, spb_stream_t stream);
/* the same sum of squares as SPB_SQ_PARTS per-workgroup partials (plain stores, no arrival counter: 512 same-address
 * atomics cost the one-scalar form ~13 of its 20 us on the KRN arena); spb_optim_step(sq_partials) consumes them */
#define SPB_SQ_PARTS 256
int spb_grad_sqnorm_partials(const float* grads, long long n, float* partials_out /*[SPB_SQ_PARTS]*/, spb_stream_t stream);
/* optimizer.zero_grad() on the flat f32 gradient arena (trainer.py:81, dann.py:74), and dst += src over two arenas (the two
 * backward passes of a DANN step accumulate into one gradient, dann.py:95); n elements, n % 4 == 0 for the add */
int spb_arena_zero(float* arena, long long n, spb_stream_t stream);
int spb_arena_add(float* dst, const float* src, long long n, spb_stream_t stream);
/* A HIP stream at the device's highest (level < 0), default (0) or lowest (level > 0) dispatch priority.  The SPN step runs
 * its 0.8 ms HBM-bound parameter update on a lowest-priority stream beside the trunk's backward: the dispatcher then fills
 * the launch stream's kernels first and the update takes what they leave (host side: nets/spn.py loss_and_grads). */
int spb_stream_create(int level, spb_stream_t* out);
int spb_stream_destroy(spb_stream_t stream);
/* Stream fork without an event: everything enqueued on `to` after this call runs behind everything enqueued on `from` before it.
 * What it replaces: torch.cuda.Stream.wait_stream / hipEventRecord + hipStreamWaitEvent in the host code that spreads one training
 * step of the reference (trainer.py:72-98,146-185: one stream there) over several streams -- an event record costs the recording
 * stream 6-9 us on this device, this fork 1.6 us: a one-wave kernel on `from` stores a serial number to a device word owned by the
 * object, a one-wave gate kernel on `to` spins on it (both streams on ONE device; one object per `from` stream: serials must be
 * stored in order).  Falls back to an event (created without the system-scope fence) inside a stream capture, when SPB_EVENT_FORKS=1
 * is set, and under rocprofv3 counter collection (ROCPROF_COUNTER_COLLECTION=1: kernels are serialised there, a gate would spin
 * forever; it traps after SPB_FORK_TIMEOUT_S seconds -- default 600, 0 = never -- if the storing launch never runs). */
/* Round 6: the property the device-word forks rest on -- a kernel starts only after every earlier kernel of its stream has completed and
 * released its results at device scope -- is TESTED once per process, at the first fork, on that fork's own two streams (~1 ms and one
 * synchronisation of both: a slow producer, a dependent one-wave kernel that stores the word, a gate + checker on the second stream;
 * csrc/elemwise.hip).  spb_fork_selftest() returns the cached verdict -- if no fork has run yet it runs the test on two streams of its own --: 1 (passed: device-word forks), 0 (failed: events for the rest of the
 * process, one line on stderr), -1 (events forced by SPB_EVENT_FORKS / ROCPROF_COUNTER_COLLECTION).  A gate that gives up after
 * SPB_FORK_TIMEOUT_S seconds no longer traps: it raises a host-visible poison word and the owner's next call returns SPB_E_TIMEOUT. */
int spb_fork_selftest(void);
int spb_hip_runtime_version(void);   /* hipRuntimeGetVersion() of the runtime the library is bound to (bench.py records it) */
typedef struct spb_fork spb_fork_t;
int spb_fork_create(spb_fork_t** out);
void spb_fork_destroy(spb_fork_t* f);
int spb_fork_streams(spb_fork_t* f, spb_stream_t from, spb_stream_t to);
typedef struct spb_optim_args {
  float* params; float* grads; float* m; float* v; /* flat f32 arenas; m/v may be NULL for sgd w/o momentum */
  const float* sqnorm;  /* optional device scalar: clip coefficient = min(1, max_norm/(sqrt(sqnorm)+1e-6)) */
  const float* gmul;    /* optional device scalar multiplied into every gradient (1/world_size, 1/loss_scale) */
  const float* hyper;   /* optional device [3] = lr, 1-beta1^t, 1-beta2^t: overrides the by-value fields so a captured
                           graph can be replayed with per-step values (the host refreshes this buffer before each replay) */
  long long n;
  int kind;             /* 0 sgd, 1 rmsprop, 2 adam, 3 adamw */
  float lr, beta1, beta2, eps, weight_decay, max_norm, clip_value; /* max_norm<=0: no norm clip; clip_value<=0: none */
  float bias_c1, bias_c2; /* 1-beta1^t, 1-beta2^t */
  int first_step;       /* sgd momentum buffer initialisation */
  void* shadow_bf16;    /* optional bf16 arena with the parameter arena's offsets: receives the updated value of every
                           element, so the next forward needs no separate conversion pass (SPN, 152 M parameters) */
  int max_blocks;       /* > 0: at most this many workgroups, each walking the arena with a grid stride -- a background
                           update that leaves the compute units to the kernels of another stream; 0: one block per run */
  const float* skip;        /* optional device scalar: != 0 -> the whole launch does nothing (AMP: non-finite gradients, see spb_amp_step) */
  const float* sq_partials; /* optional, instead of sqnorm: n_sq_partials (<= 256) partial sums of squares from
                               spb_grad_sqnorm_partials; every workgroup adds them up itself (no single-address ticket) */
  int n_sq_partials;
} spb_optim_args_t;
int spb_optim_step(const spb_optim_args_t* a, spb_stream_t stream);

/* Dynamic loss scaling on the device: torch.cuda.amp.GradScaler's arithmetic (reference trainer.py:146-181, train.py:101-104)
 * with no host read of found_inf.  `state` is SPB_AMP_STATE floats the caller initialises to {scale = 65536, 0, ...}:
 *   spb_softce_scaled(..., state + SPB_AMP_SCALE)      the loss gradient carries the scale (the loss value does not)
 *   spb_amp_check(grads, n, state)                      state[FOUND_INF] = 1 if any gradient element is inf / nan
 *   spb_amp_step(state, lr, beta1, beta2, 2, 0.5, 2000) inv_scale of the step in flight, the skip flag, lr and the Adam bias
 *                                                       corrections of the optimizer's OWN step count (it does not advance on a
 *                                                       skipped step), then GradScaler.update(): scale *= backoff on overflow,
 *                                                       *= growth after `interval` clean steps in a row
 *   spb_optim_step(gmul = state + INV_SCALE, hyper = state + LR, skip = state + SKIP)                                        */
#define SPB_AMP_STATE 12
#define SPB_AMP_SCALE 0
#define SPB_AMP_INV_SCALE 1
#define SPB_AMP_TRACKER 2
#define SPB_AMP_FOUND_INF 3
#define SPB_AMP_STEPS 4
#define SPB_AMP_LR 5      /* LR, BC1, BC2 are consecutive: spb_optim_args_t.hyper */
#define SPB_AMP_BC1 6
#define SPB_AMP_BC2 7
#define SPB_AMP_SKIP 8
#define SPB_AMP_TICKET 9   /* spb_amp_decide: workgroup ticket (an unsigned integer in the float slot; zero between launches) */
#define SPB_AMP_SEGS 8
typedef struct spb_amp_segs { const void* ptr[SPB_AMP_SEGS]; long long n[SPB_AMP_SEGS]; int is16[SPB_AMP_SEGS]; int nseg; } spb_amp_segs_t;
int spb_amp_check(const float* grads, long long n, float* state, spb_stream_t stream);
int spb_amp_check16(const void* x, long long n, float* state, spb_stream_t stream);   /* the same on a 16-bit tensor (the library's storage format); n % 8 == 0 */
int spb_amp_step(float* state, float lr, float beta1, float beta2, float growth, float backoff, int interval, spb_stream_t stream);
/* spb_amp_check / spb_amp_check16 over up to SPB_AMP_SEGS segments (f32: n % 4 == 0; 16-bit: n % 8 == 0; 16-byte aligned) and spb_amp_step
 * in ONE launch: the last workgroup to finish takes the decision */
int spb_amp_decide(const spb_amp_segs_t* segs, float* state, float lr, float beta1, float beta2, float growth, float backoff, int interval,
                   spb_stream_t stream);
/* Weight gradient of a fully connected layer (spb_fc_wgrad's operands: GT [N][MP], XT [K][MP], batch M <= 64) fused with that
 * layer's share of the optimizer step: opt->params / m / v / shadow_bf16 point at the layer's [N][K] weight (opt->n == N*K),
 * opt->grads is NULL or receives the raw gradient.  Same arithmetic per element as spb_fc_wgrad followed by spb_optim_step;
 * max_norm must be 0 (a global-norm clip needs all gradients first -- the SPN trainer clips by value, trainer.py:177). */
int spb_fc_wgrad_update(const void* GT, const void* XT, int M, int N, int K, const spb_optim_args_t* opt, spb_stream_t stream);

/* ---- whole-network plans (C++ runtime: layer graph, workspace layout, launch sequencing) -------------------- */
typedef struct spb_krn spb_krn_t;
typedef struct spb_tensor_info {
  char name[96];      /* state-dict key, e.g. "base.3.conv.1.0.weight" */
  long long offset;   /* element offset in the f32 parameter (or buffer) arena */
  long long numel;
  int ndim;
  int shape[4];
} spb_tensor_info_t;

/* KeypointRegressionNet (park2019.py:101-165).  dann=1 adds RevGrad's domain classifier (revgrad.py:58-96);
 * parameter names then carry the "net." / "domain_classifier." prefixes of RevGrad.state_dict().               */
int spb_krn_create(int num_keypoints, int dann, spb_krn_t** out);
void spb_krn_destroy(spb_krn_t* m);
int spb_krn_num_params(const spb_krn_t* m);
int spb_krn_param_info(const spb_krn_t* m, int i, spb_tensor_info_t* out);
int spb_krn_num_buffers(const spb_krn_t* m); /* running_mean / running_var (f32 arena) */
int spb_krn_buffer_info(const spb_krn_t* m, int i, spb_tensor_info_t* out);
int spb_krn_num_bn(const spb_krn_t* m);      /* num_batches_tracked entries (int64 arena, one per BN) */
int spb_krn_bn_name(const spb_krn_t* m, int i, char* out96);
long long spb_krn_param_numel(const spb_krn_t* m);
long long spb_krn_buffer_numel(const spb_krn_t* m);
long long spb_krn_wcompute_bytes(const spb_krn_t* m, int dtype);
long long spb_krn_tables_bytes(const spb_krn_t* m);
/* binds the arenas; tables_dev is a device scratch of spb_krn_tables_bytes() the call fills (synchronously). */
int spb_krn_bind(spb_krn_t* m, float* params, float* grads, float* buffers, long long* nbt, void* wcompute,
                 void* tables_dev, int dtype);

typedef struct spb_krn_ctx spb_krn_ctx_t; /* activations of ONE forward pass at a fixed batch size */
long long spb_krn_ctx_bytes(const spb_krn_t* m, int batch, int dtype);
int spb_krn_ctx_create(spb_krn_t* m, int batch, void* workspace, spb_krn_ctx_t** out);
void spb_krn_ctx_destroy(spb_krn_ctx_t* c);
/* The pointwise weight-gradient GEMMs of spb_krn_backward only feed the optimizer; by default they are enqueued on a
 * context-owned side stream (forked from / joined to `stream` with events) so that they overlap the input-gradient
 * chain.  on = 0 keeps every launch on `stream` (use this when capturing into a hipGraph: measured slower there). */
int spb_krn_ctx_set_side_stream(spb_krn_ctx_t* c, int on);

/* ---- reproducible mode (libspb_hip_det.so: the KRN sources compiled with -DSPB_DET, csrc/common.h) --------------------------
 * The reference leaves training non-deterministic (utils.py:297-298: cudnn.benchmark = True, cudnn.deterministic = False); so does
 * libspb_hip.so (float atomics in the batch-sum and weight-gradient kernels).  The twin library accumulates every such sum EXACTLY
 * (four 64-bit fixed-point windows per float slot, integer atomics: order-independent), so a training run is a pure function of its
 * inputs: bit-identical in every process, whatever ran on the device before.  Used by tests/test_parity_conditioned_gpu.py to
 * condition ONE reproducible state, and available to users as KrnEngine(..., deterministic=True).
 *   spb_det_register(lo, n, shadow): float range [lo, lo+n) accumulates into shadow[4*n] (int64, zero-initialised, caller-owned);
 *   spb_det_flush(lo, stream): fold the windows of that region into the floats and clear them;
 *   spb_det_misses(): float atomics that hit no region since the last call (0 = the run was fully exact);
 *   spb_krn_set_det / spb_krn_ctx_set_det: the plan keeps every launch on the caller's stream and flushes after each launch.
 * libspb_hip.so exports the same symbols; there spb_det_available() is 0 and the others return SPB_E_UNSUPPORTED.            */
int spb_det_available(void);
int spb_det_register(const float* lo, long long n_floats, long long* shadow);
int spb_det_unregister(const float* lo);
int spb_det_unregister_if(const float* lo, const long long* shadow);   /* only if that region still accumulates into `shadow`: an owner releasing
                                                                          ITS registration (the address may have a later owner by then; registering
                                                                          a range drops every older region it overlaps) */
int spb_det_flush(const float* lo, spb_stream_t stream);
long long spb_det_misses(void);
int spb_krn_set_det(spb_krn_t* m, int on);
int spb_krn_ctx_set_det(spb_krn_ctx_t* c, int on);
int spb_krn_ctx_stats(spb_krn_ctx_t* c, float** ptr, long long* n_floats);

/* Introspection (parity tests): BatchNorm'd tensor a of the plan = raw convolution output z [B,H,W,C] (NHWC, compute dtype) at byte
 * offset z_off of the context's workspace, its backward companion g at g_off, normalised by BatchNorm bn_index (spb_krn_bn_name). */
typedef struct { long long z_off, g_off; int H, W, C, bn_index; } spb_act_info_t;
int spb_krn_num_acts(const spb_krn_t* m);
int spb_krn_ctx_act_info(const spb_krn_ctx_t* c, int a, spb_act_info_t* out);
/* Round 6: the expanded tensors of inverted-residual blocks 2-4 (16 -> 96 at 112x112, 24 -> 144 at 56x56) are VIRTUAL in 16-bit mode:
 * only their batch sums exist, every consumer recomputes the values (spb_dw_args_t::Xe).  spb_krn_ctx_virtual(c, a) = 1 for such a
 * tensor; spb_krn_ctx_materialize writes z of all of them into their workspace slots (z_off above) from the last forward pass's
 * state, rounded to the storage type, for tests that want to look at them.  Their g (g_off) is a real tensor. */
int spb_krn_ctx_virtual(const spb_krn_ctx_t* c, int a);
int spb_krn_ctx_materialize(spb_krn_ctx_t* c, spb_stream_t stream);

/* refresh compute-dtype weight copies (W, W^T, permuted head) from the f32 parameters */
int spb_krn_prepare_weights(spb_krn_t* m, spb_stream_t stream);
/* forward.  training=1: batch statistics + running-stat update; training=2: batch statistics, the running-stat update is
 * left to a later spb_krn_update_running (two passes on two streams, see below).  training | 4: spb_krn_prepare_weights first,
 * on the context's side stream beside the stem and the first depthwise layer (they read the f32 parameters), joined before the
 * first 1x1 convolution.  training | 8: the bound gradient arena is zeroed (optimizer.zero_grad()) on the same side-stream
 * fork.  training | 16 (with training=1): the running-stat update is enqueued by the NEXT spb_krn_backward on this context,
 * on its side stream (nothing in a train step reads the running statistics); a forward or spb_krn_update_running that
 * comes first applies it.  target NULL => prediction only.
 * pred [B][2K] f32 (interleaved x,y as the head emits them), scalars [3] = loss, loss_x, loss_y.
 * alpha_valid=1 with a dann plan also runs the domain classifier: domain_logits [B].                           */
int spb_krn_forward(spb_krn_ctx_t* c, const float* x_nchw, const float* target, int training, float* pred,
                    float* scalars, float* domain_logits, spb_stream_t stream);
/* Data-parallel gradient exchange overlapped with backward (one process per GPU, RCCL).  The arena tail
 * [spb_krn_bucket_split(m), n_params) -- inverted-residual blocks 14..17, the ConvDw extras, the head and the RevGrad domain
 * classifier: ~90 % of the elements -- is final after the 7x7 part of the backward pass; spb_krn_backward records an event
 * there and spb_krn_ctx_wait_bucket makes the communication stream wait on it, so that bucket's all-reduce runs beside the
 * backward of blocks 13..1 and the stem.  The head of the arena is reduced after backward. */
long long spb_krn_bucket_split(const spb_krn_t* m);
int spb_krn_ctx_set_bucket(spb_krn_ctx_t* c, int on);   /* off (default): no mid-backward join, no event */
int spb_krn_ctx_wait_bucket(spb_krn_ctx_t* c, spb_stream_t comm_stream);
/* BatchNorm running_mean / running_var / num_batches_tracked update from the batch sums of this context's last training
 * forward (what training=1 does at the end of the forward).  DANN (dann.py:81-92) runs its source and target passes
 * concurrently on two streams; the shared buffers are still updated source first, then target, on one stream. */
int spb_krn_update_running(spb_krn_ctx_t* c, spb_stream_t stream);
/* backward of loss*gscale (+ sum_b domain_logit_grad[b]*logit[b] through the gradient-reversal layer with alpha).
 * ACCUMULATES into `grads` (f32 arena with the parameter layout; NULL = the arena given to spb_krn_bind); the caller
 * zeroes it, as optimizer.zero_grad does.                                                                       */
int spb_krn_backward(spb_krn_ctx_t* c, float* grads, float gscale, int with_pose, const float* domain_logit_grad,
                     float alpha, spb_stream_t stream);
/* binary_cross_entropy_with_logits(logits, full(label), reduction='mean') and its gradient * gscale (dann.py:85-92) */
/* float16 recipe (reference: torch.cuda.amp.autocast + GradScaler around the KRN step, trainer.py:73-94): `scale` is a device scalar
 * (SPB_AMP_SCALE of an AMP state, spb_amp_*) that every later spb_krn_backward on this context multiplies onto the upstream gradient;
 * NULL switches it off.  Only the IEEE-half build (libspb_hip_f16.so) needs it: bfloat16 has float32's exponent range. */
int spb_krn_ctx_set_loss_scale(spb_krn_ctx_t* c, const float* scale);
int spb_bce_logits(const float* logits, float label, int B, float* loss_out, float* dlogit_out, float gscale,
                   spb_stream_t stream);

/* live per-launch timing of the plan's kernels: HIP events recorded on the launch stream around every launch, grouped
 * by kernel family, with the ALGORITHMIC bytes/flops of each launch (every operand read once, every result written
 * once).  bench.py derives roofline.achieved from these.                                                          */
int spb_krn_prof_enable(spb_krn_ctx_t* c, int on);
int spb_krn_prof_num_categories(void);
const char* spb_krn_prof_category_name(int i);
int spb_krn_prof_read(spb_krn_ctx_t* c, int* launches, float* ms, double* bytes, double* flops);
/* per-launch records since the last spb_krn_prof_read, in launch order (category index, ms, algorithmic bytes); returns the
 * number of records, writes at most `max`; does not reset */
int spb_krn_prof_launches(spb_krn_ctx_t* c, int max, int* cat, float* ms, double* bytes);
long long spb_krn_weight_prep_bytes(const spb_krn_t* m);

/* ---- style-transfer decoder (Ghiasi), inference only: src/styleaug/ghiasi.py:6-135, called from
 * StyleAugmentor.forward (styleAugmentor.py:48-68).  An instance-normalised tensor is its RAW conv output (NHWC bf16)
 * plus per-(image, channel) sums; consumers apply a = act(x*scale[b,c] + shift[b,c]) while loading. */
typedef struct {
  const void* X;      /* [B,Hin,Win,Cin] NHWC bf16 */
  const void* W;      /* [Cout][KH*KH][Cin] bf16 (PyTorch weight permuted (0,2,3,1)) */
  const float* bias;  /* [Cout] or NULL */
  const float* coef;  /* [B][Cin][2] scale, shift applied to X on load, or NULL */
  void* Y;            /* [B,Hout,Wout,ldc] NHWC bf16, raw conv output (+bias) */
  float* stats;       /* [B][Cout][2] f32 accumulated: sum and sum of squares of the stored output, or NULL */
  int B, Hin, Win, Cin, Cout, KH, stride, upsample, relu, ldc;
  /* Input transform straight from the PRODUCER's sums (replaces `coef` and the spb_in_coef launch that would fill it) when
   * in_stats != NULL: scale = gamma * rsqrt(max(s2/n - (s1/n)^2, 0) + eps), shift = beta - (s1/n) * scale, computed once per
   * workgroup and channel.  in_stats [B][Cin][2]; in_gamma / in_beta rows [B][in_ld] (NULL: 1 / 0); in_inv_n = 1 / (H*W of X).
   * spb_gconv and spb_gconv_up2 honour it; spb_gconv_wide reads `coef`. */
  const float* in_stats;
  const float* in_gamma;
  const float* in_beta;
  int in_ld;
  float in_inv_n, in_eps;
} spb_gconv_args_t;
/* KxK conv (K = 3 | 9), nn.ReflectionPad2d(K/2), stride 1|2, optional nearest x2 upsampling of the input
 * (torch.nn.Upsample(scale_factor=2)); Hout = Hin*upsample/stride must be a multiple of 8.  SPB_BF16: the matrix-core kernels.
 * SPB_F32: the reference-precision mode (the reference runs the decoder in fp32, trainer.py:68-69): X / Y float32 NHWC, W float32
 * [Cout][KH*KH][Cin], any Cin / Cout / odd KH, Hout*Wout % 64 == 0, the `coef` table only (no in_stats) -- a direct convolution for parity. */
int spb_gconv(int dtype, const spb_gconv_args_t* args, spb_stream_t stream);
/* UpsampleConvInRelu's Upsample(2, nearest) + ReflectionPad2d(1) + Conv2d 3x3 (ghiasi.py:46-59) as four 2x2 convolutions on the
 * LOW-RESOLUTION input, one per output phase (upsample == 2, stride == 1, KH == 3): args->W holds the phase weights
 * [4 = py*2+px][Cout][4 = ty*2+tx][Cin] bf16 with (w0, w1+w2) / (w0+w1, w2) summed along each axis; index clamping on the
 * low-resolution image equals the reflection padding of the upsampled one.  Same result as spb_gconv up to one bf16 rounding
 * of the summed weights, 2.25x fewer matrix-core steps. */
int spb_gconv_up2(int dtype, const spb_gconv_args_t* args, spb_stream_t stream);
/* The residual blocks' 128 -> 128 3x3 convolutions (ghiasi.py:92-104; Cin == Cout == 128, stride 1, no upsampling, Hin and Win
 * multiples of 8, ldc % 8 == 0): same result as spb_gconv, args->W holds the weights as spb_gconv_wide_pack lays them out
 * ([36 reduction steps][8 KB LDS image of the step's 128 x 32 slab]); csrc/ghiasi_wide.hip. */
int spb_gconv_wide(int dtype, const spb_gconv_args_t* args, spb_stream_t stream);
/* w [128][9][128] bf16 (the spb_gconv layout) -> packed, 294 912 bytes */
int spb_gconv_wide_pack(const void* w, void* packed, spb_stream_t stream);
/* first layer: Conv2d(3,32,9) with reflection padding on the fp32 NCHW image -> NHWC bf16 [B,H,W,32] + stats; W % 16 == 0 */
int spb_conv9_rgb(const float* x, const float* w_oihw, const float* bias, void* y, float* stats, int B, int H, int W,
                  spb_stream_t stream);
/* coef[b][c] = (gamma*invstd, beta - mean*gamma*invstd); gamma/beta [B][ld] rows (NULL: 1 / 0); eps as InstanceNorm2d */
int spb_in_coef(const float* stats, const float* gamma, const float* beta, int ld, float* coef, int B, int C, long long hw,
                float eps, spb_stream_t stream);
/* out[b][j] = bias[j] + sum_i style[b][i]*W[j][i]: every nn.Linear(100, C) of the decoder stacked into one [N,100] matrix */
int spb_style_fc(const float* style, const float* W, const float* bias, float* out, int B, int N, spb_stream_t stream);
/* Y = [res +] act(X*scale + shift), NHWC bf16 (the residual stream of ResidualBlock, ghiasi.py:92-104) */
int spb_in_apply(const void* X, const float* coef, const void* res, void* Y, int B, long long hw, int C, int relu,
                 spb_stream_t stream);
/* out (fp32 NCHW, 3 channels) = sigmoid(Z*scale + shift), Z NHWC bf16 with channel stride ldc (ghiasi.py:135) */
int spb_final_sigmoid(const void* Z, const float* coef, float* out, int B, long long hw, int ldc, spb_stream_t stream);
/* float32 instances of the two for the reference-precision mode */
int spb_in_apply_f32(const float* X, const float* coef, const float* res, float* Y, int B, long long hw, int C, int relu, spb_stream_t stream);
int spb_final_sigmoid_f32(const float* Z, const float* coef, float* out, int B, long long hw, int ldc, spb_stream_t stream);
/* the same two with the coefficients taken from the producer's sums (arguments as spb_in_coef; no coefficient launch in between) */
int spb_in_apply_stats(const void* X, const float* stats, const float* gamma, const float* beta, int ld, float eps, const void* res,
                       void* Y, int B, long long hw, int C, int relu, spb_stream_t stream);
int spb_final_sigmoid_stats(const void* Z, const float* stats, const float* gamma, const float* beta, int ld, float eps, float* out,
                            int B, long long hw, int ldc, spb_stream_t stream);

/* ---- Spacecraft Pose Network building blocks (src/nets/spn.py:37-143; loss assembly src/core/trainer.py:160-165).
 * Convolutions and fully connected layers run through spb_pwconv_gemm / spb_pwconv_wgrad on im2col'd operands; all
 * tensors NHWC, dtype SPB_BF16 or SPB_F32. */
/* dst[(b,oy,ox)][gi*Kg + (ky*KW+kx)*cig + cl] = src[b, oy*stride-pad+ky, ox*stride-pad+kx, gi*cig + cl] (0 outside), cig = C/groups,
 * Kg = Kpad/groups: one contiguous column slab per convolution group, so a grouped nn.Conv2d (spn.py:60,66,68) is one dense
 * GEMM per group on a slab (spb_gemm_args_t.lda / ldc).  C/groups % 8 == 0, Kpad >= KH*KW*C */
int spb_im2col(int dtype, const void* src, void* dst, int B, int H, int W, int C, int KH, int KW, int stride, int pad, int Kpad,
               int groups, spb_stream_t stream);
/* first layer: fp32 NCHW image (3 channels), valid padding; k = (ci*KH + ky)*KW + kx (chw_order of spb_spn_pack_conv) */
int spb_im2col_rgb(int dtype, const float* x, void* dst, int B, int H, int W, int KH, int KW, int stride, int Kpad,
                   spb_stream_t stream);
/* adjoint of spb_im2col for stride 1: dx[b,iy,ix,c] = sum over taps of dcol */
int spb_col2im(int dtype, const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int pad, int Kpad,
               int groups, spb_stream_t stream);
/* nn.MaxPool2d(3, stride=2): argmax (0..8, first maximum in scan order) is needed by the backward gather */
int spb_maxpool3s2_fwd(int dtype, const void* x, void* y, unsigned char* argmax, int B, int H, int W, int C, spb_stream_t stream);
int spb_maxpool3s2_bwd(int dtype, const void* dy, const unsigned char* argmax, void* dx, int B, int H, int W, int C,
                       spb_stream_t stream);
/* the same with the ReLU backward of the pooled tensor folded in: dx = maxpool_bwd(dy) * (y > 0), y = the ReLU output that was
 * pooled ([B][H][W][C]; spn.py:61,66,70 pool right after a ReLU) */
int spb_maxpool3s2_relu_bwd(int dtype, const void* dy, const unsigned char* argmax, const void* y, void* dx, int B, int H, int W, int C,
                            spb_stream_t stream);
/* nn.LocalResponseNorm(2, alpha, beta, k) over the channel axis of [npix, C] */
int spb_lrn2_fwd(int dtype, const void* x, void* y, long long npix, int C, float alpha, float beta, float k, spb_stream_t stream);
int spb_lrn2_bwd(int dtype, const void* x, const void* g, void* dx, long long npix, int C, float alpha, float beta, float k,
                 spb_stream_t stream);
/* g = (dy [+ add]) * (y > 0) * scale: ReLU backward (y = activation output; with inverted dropout y is post-dropout) */
int spb_relu_bwd(int dtype, const void* dy, const void* y, const void* add, void* g, long long n, float scale, spb_stream_t stream);
/* nn.Dropout(p), in place: y *= keep/(1-p); keep from a counter hash of (seed, index) and written to mask, or read from it */
int spb_dropout(int dtype, void* y, unsigned char* mask, long long n, float p, unsigned long long seed, int use_given_mask,
                spb_stream_t stream);
/* softmax_cross_entropy_with_logits(logits, target, 'mean') (spn.py:37-48): out[0] += weight*loss, out[slot] += loss;
 * dlogits (may be NULL) = d(weight*loss)/dlogits */
int spb_softce(int dtype, const void* logits, const float* target, void* dlogits, float* out, int slot, int B, int C, float weight,
               spb_stream_t stream);
/* the same with dlogits multiplied by the device scalar *gscale (the AMP loss scale, spb_amp_* below); out is unscaled */
int spb_softce_scaled(int dtype, const void* logits, const float* target, void* dlogits, float* out, int slot, int B, int C,
                      float weight, const float* gscale, spb_stream_t stream);
/* the same with reduction='none' (spn.py:43-44): rows[b] = -sum_c target[b][c] * log_softmax(logits[b])[c] */
int spb_softce_rows(int dtype, const void* logits, const float* target, float* rows, int B, int C, spb_stream_t stream);
/* out[n] += sum_m g[m][n] (bias gradients) */
int spb_colsum(int dtype, const void* g, float* out, long long M, int N, spb_stream_t stream);

/* nn.Conv2d weights [Cout][Cin/groups][KH][KW] f32 (spn.py:56-70) -> Wp [Cout][Kg]: row co holds its group's filter in
 * (ky,kx,c_local) order, zero padded to Kg (i.e. `groups` stacked [Cout/groups][Kg] GEMM operands matching spb_im2col's
 * slabs), and WpT [groups][Kg][Cout/groups], the per-group transposes the input gradient needs (may be NULL).  dtype = output. */
/* ---- SPN trunk convolutions as implicit GEMMs (csrc/spn_conv.hip; reference spn.py:60-101 nn.Conv2d + ReLU) -------------
 * X: bf16 NHWC [B][H][W][Cx]; Wp: bf16 [groups*Ng][Kp], row n = (ky, kx, c) of output channel n over its group's Cg input
 * channels (spb_spn_pack_conv's layout), zero padded to Kp; Y: bf16 [B*OH*OW][groups*Ng] = act(conv + bias).
 * mask (optional, same shape as Y): elements of Y where mask <= 0 are stored as 0 -- the ReLU backward of the layer below
 * when the call computes an input gradient: X = output gradient, Wp = spb_spn_pack_conv_dgrad's mirrored weights,
 * Cg <-> Ng swapped, pad = K-1-pad (stride-1 layers).  Cg, Cx, Kp must be multiples of 8 (16-byte operand vectors). */
typedef struct spb_spn_conv_args {
  const void* X; const void* Wp; const float* bias; const void* mask; void* Y;
  int B, H, W, Cx, KH, KW, stride, pad, groups, Cg, Ng, Kp, relu;
} spb_spn_conv_args_t;
int spb_spn_conv(const spb_spn_conv_args_t* args, spb_stream_t stream);
/* weight gradient of the same convolution (args as for the forward call; Wp / bias / mask / Y / relu unused):
 * dWp f32 [groups*Ng][Kp] += sum over output pixels of G[m][n] * X[pixel(m, tap)][c]; G: bf16 [B*OH*OW][groups*Ng].
 * Partial sums of the pixel ranges meet in dWp with float atomics: zero it first. */
int spb_spn_conv_wgrad(const spb_spn_conv_args_t* args, const void* G, float* dWp, spb_stream_t stream);
/* conv1: float32 NCHW image [B][3][H][W] -> bf16 NHWC relu(conv + bias), 11 x 11, stride 4, no padding, 96 output channels
 * (other shapes: SPB_E_UNSUPPORTED); Wb [96][Kb = 544] in the band layout k' = (c*11 + ky)*16 + kx (spb_spn_pack_jobs mode 2:
 * kernel rows padded to 16 columns with zeros).  No column matrix: a band of image rows is staged in LDS. */
int spb_spn_stem(const float* x, const void* Wb, const float* bias, void* Y, int B, int H, int W, int KH, int KW, int stride, int N, int Kb,
                 int relu, spb_stream_t stream);
/* All the repacks a step needs after the optimizer changed the convolution weights, in one launch.  mode 0: spb_spn_pack_conv
 * (out [Cout][Kp], optional outT [groups][Kp][Cout/groups], chw = the RGB stem's (c, ky, kx) column order); mode 1:
 * spb_spn_pack_conv_dgrad (out [Cin][Kp]); mode 2: the RGB stem's band layout (out [Cout][Kp], k' = (c*KH + ky)*16 + kx, KW <= 16).
 * dtype selects bf16 / f32 outputs for every job. */
#define SPB_SPN_MAX_PACK_JOBS 12
typedef struct spb_spn_pack_job {
  const float* W; void* out; void* outT;
  int Cout, Cin, groups, KH, KW, Kp, mode, chw;
} spb_spn_pack_job_t;
int spb_spn_pack_jobs(int dtype, const spb_spn_pack_job_t* jobs, int njobs, spb_stream_t stream);
/* the same kernel on an explicit column matrix: dW f32 [N][lddw] += G^T (bf16 [M][N]) * col (bf16 [M][ldcol], K columns used) */
int spb_spn_col_wgrad(const void* G, const void* col, float* dW, int M, int N, int K, int ldcol, int lddw, spb_stream_t stream);
/* W: f32 [Cout][Cin/groups][KH][KW] -> WpD: bf16 [Cin][KpD], row g*Cg+ci = (mirrored tap, n) over the group's Cout/groups */
int spb_spn_pack_conv_dgrad(const float* W, void* WpD, int Cout, int Cin, int groups, int KH, int KW, int KpD, spb_stream_t stream);
int spb_spn_pack_conv(int dtype, const float* W, void* Wp, void* WpT, int Cout, int Cin, int groups, int KH, int KW, int Kg,
                      int chw_order, spb_stream_t stream);   /* chw_order != 0: rows keep nn.Conv2d's (c,ky,kx) order (conv1 / spb_im2col_rgb) */
/* inverse for gradients: dWp f32 [Cout][Kg] -> dW f32 [Cout][Cin/groups][KH][KW] */
int spb_spn_unpack_conv_grad(const float* dWp, float* dW, int Cout, int Cin, int groups, int KH, int KW, int Kg, int chw_order,
                             spb_stream_t stream);

/* ---- SPN fully connected layers at training batch sizes (M <= 64 rows; nn.Linear fc6..fc11, spn.py:71-99), bf16 only.
 * accT is a feature-major f32 accumulator [features][MP], MP = 32 (M <= 32) or 64, zero between uses; the epilogue
 * consumes and re-zeroes it.  Return SPB_E_UNSUPPORTED for shapes outside the streamed kernels' reach (the caller then
 * uses spb_pwconv_gemm / spb_pwconv_wgrad).
 *   spb_fc_fwd    accT[n][m] += sum_k W[n][k] X[m][k]          X [M][K], W [N][K];  K % 64 == 0
 *   spb_fc_dgrad  accT[k][m] += sum_n W[n][k] G[m][n]          G [M][N];            K % 128 == 0, N % 8 == 0
 *   spb_fc_wgrad  dW[n][k]    = sum_m GT[n][m] XT[k][m]        GT [N][MP], XT [K][MP] (written by the epilogue); K % 4 == 0 */
int spb_fc_fwd(const void* X, const void* W, float* accT, int M, int N, int K, spb_stream_t stream);
int spb_fc_dgrad(const void* G, const void* W, float* accT, int M, int N, int K, spb_stream_t stream);
int spb_fc_wgrad(const void* GT, const void* XT, float* dW, int M, int N, int K, spb_stream_t stream);
typedef struct spb_fc_epi_args {
  float* accT;              /* [F][MP] f32 accumulator (consumed and zeroed) or NULL */
  const void* src;          /* accT == NULL: bf16 [M][F] values to use instead (e.g. dlogits from spb_softce) */
  const float* bias;        /* mode 0: [F] or NULL */
  const void* H;            /* mode 1: bf16 [M][F] forward activation (after ReLU / dropout); gradient passes where H > 0; NULL: everywhere */
  void* Y;                  /* bf16 [M][F] or NULL */
  void* YT;                 /* bf16 [F][MP] (rows >= M zero) or NULL */
  unsigned char* mask;      /* mode 0, p > 0: keep mask [M][F] (written, or read when mask_given) */
  float* db;                /* mode 1: [F] column sums of the result (bias gradient) or NULL */
  int M, F;
  int mode;                 /* 0 forward: bias, ReLU (relu != 0), nn.Dropout(p);  1 backward: * scale where H > 0 */
  int relu;
  float p, scale;
  unsigned long long seed;
  int mask_given;
} spb_fc_epi_args_t;
int spb_fc_epilogue(const spb_fc_epi_args_t* a, spb_stream_t stream);
/* pool5 output NHWC bf16 [B][HW][C] -> Fm [B][C*HW] in the NCHW order of x.view(-1, 9216) (spn.py:131) and FT [C*HW][MP] */
int spb_spn_flatten(const void* P, void* Fm, void* FT, int B, int HW, int C, spb_stream_t stream);
/* accT [C*HW][MP] (both heads' fc6 / fc9 input gradients summed) -> NHWC bf16 gradient of pool5's output; accT zeroed */
int spb_spn_unflatten_grad(float* accT, void* Gp, int B, int HW, int C, spb_stream_t stream);

/* ---- GPU input pipeline: resize of the host-cropped region of interest + ToTensor + augmentations for a
 * whole batch.  Replaces, per sample, T.resized_crop (torchvision 0.9.0 on a PIL image = crop + Image.resize(BILINEAR)),
 * T.to_tensor, Rotate / Flip / BrightnessContrast / GaussianNoise of src/datasets/transforms.py:38-196 as composed by
 * build_transforms (transforms.py:217-244) and called from Park2019KRNDataset.__getitem__ (Park2019KRNDataset.py:81-109).
 * Bit-exact against Pillow's Resample.c arithmetic and the reference's float32 tensor ops for given random draws; the draws
 * themselves (crop jitter, coins, angle, a, b, noise) are the host's (speedplusbaseline_amd/transforms.py). */
typedef struct spb_preproc_args {
  const void* src;      /* packed uint8 crops: image b starts at byte table[b].off, rows of w pixels, C interleaved channels */
  const int* table;     /* device [B][8] int32: off_lo, off_hi, h, w, rot (quarter turns counter-clockwise 0..3),
                           flip (0 none, 1 horizontal, 2 vertical), flags (1 brightness/contrast, 2 noise), 0 */
  const float* ftable;  /* device [B][2]: a, b of clamp(a*x + b, 0, 1) */
  const float* noise;   /* device [B][3][S][S] standard-normal draws, read where flag 2 is set; NULL if no image has it */
  float* out;           /* [B][3][S][S] float32 in [0, 1] */
  int* bounds;          /* workspace [B][2][S][2] int32 */
  int* coeffs;          /* workspace [B][2][S][spb_preproc_max_taps()] int32 */
  void* tmp;            /* workspace [B][max_h][S][C] bytes (horizontal pass) */
  int B, S, C;          /* C = 1 (grey frame, replicated to 3 bands like convert('RGB')) or 3 */
  int max_h;            /* largest crop height in the batch */
  int flags_any;        /* OR of the per-image flags */
  float noise_std;      /* GaussianNoise.std = 25/255 (transforms.py:100) */
} spb_preproc_args_t;
int spb_preproc_max_taps(void);   /* crops larger than (taps-1)/2 x S per side need more filter taps than the kernels hold */
int spb_preproc_batch(const spb_preproc_args_t* a, spb_stream_t stream);

/* debug / test helpers */
int spb_debug_trread(const unsigned short* in4096, unsigned short* out256, spb_stream_t stream);
const char* spb_version(void);

#ifdef __cplusplus
}
#endif
#endif
