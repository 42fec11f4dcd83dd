/*
 * iaf_hip.h -- C ABI of the MI355X (gfx950) IAF posterior engine.
 *
 * The reference (openai/iaf) has no FFI: its hot path sits behind Python callables.  This ABI is
 * what a ctypes wrapper carrying those callables' signatures binds (iaf_amd/_capi.py; the
 * reference-side stub is shown in INTEGRATION.md).  Each entry point cites the reference
 * interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every tensor pointer is a DEVICE pointer to contiguous fp32 (torch.Tensor.data_ptr()),
 *     activations NCHW exactly as the reference's TF path (tf_utils/layers.py:46,64);
 *   - all pointers are BORROWED for the duration of the call; outputs and the workspace are
 *     caller-allocated; the engine owns only its packed-weight cache (SURVEY 8b "Ownership");
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous on it and re-entrant per stack (no global mutable state);
 *   - return value: 0 = ok, <0 = IAF_ERR_* argument errors (the Python wrapper raises the same
 *     exception kinds as the reference's asserts), >0 = a hipError_t. Never aborts.
 */
#ifndef IAF_HIP_H
#define IAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IAF_OK 0
#define IAF_ERR_NULL (-1)          /* null pointer argument */
#define IAF_ERR_SHAPE (-2)         /* non-positive / inconsistent dimension */
#define IAF_ERR_NOT_MULTIPLE (-3)  /* n_h not a multiple of n_z or vice versa (layers.py:116 assert) */
#define IAF_ERR_NOT_PREPARED (-4)  /* forward called before iaf_stack_prepare */
#define IAF_ERR_WORKSPACE (-5)     /* workspace too small / misaligned */
#define IAF_ERR_UNSUPPORTED (-6)   /* shape outside what the gfx950 kernels cover (channels % 16) */
#define IAF_ERR_CAPTURE_SLOTS (-8)  /* a prep / weight-norm-backward batch object (iaf_prep_batch_*, iaf_wn_bwd_batch_*,
                                    * iaf_conv3x3_prep_batch_*) keeps one frozen device table of descriptors per DISTINCT set of
                                    * tensor pointers it has been captured with into hipGraphs (a captured launch cannot be fed
                                    * a table that changes under it); 16 per object, for the object's life.  Re-capturing the
                                    * same tensors shares a table.  Beyond that: create another batch object. */
#define IAF_ERR_RANGE (-9)         /* IAF_PRECISION_F16X2: an operand beyond fp16's largest finite number (65504) went into the two-plane fp16
                                    * kernels in an EARLIER launch of this stack (see iaf_stack_range_errors): that launch's outputs carry
                                    * inf / NaN; the stack has gone back to the bf16x3 kernels (whose planes have fp32's exponent range) --
                                    * prepare again if the next call says IAF_ERR_NOT_PREPARED, then repeat the call */
#define IAF_ERR_EXCHANGE (-7)      /* a bounded wait of the halo exchange gave up in an EARLIER launch of this stack (see
                                    * iaf_stack_exchange_errors): that launch's outputs carry NaN; the stack has switched to the
                                    * kernels that recompute their halo rows, and the call can simply be repeated */

#define IAF_VARIANT_TF 0           /* tf_utils/layers.py statement (parity target) */
#define IAF_VARIANT_THEANO 1       /* graphy/nodes/ar.py statement: flipped kernel (taps look left/above), border-indicator
                                      input channel, exp(3*s) scale, +1e-8 in the norm.  Weights: V[i] = <name>_w OIHW
                                      [n_out][n_in+1][3][3], g[i] = <name>_s, b[i] = <name>_b (ar.py:288-296) */
#define IAF_VARIANT_THEANO_FLIPMASK 2   /* the same with flipmask=True (graphy/nodes/ar.py:263-264): mask reversed along all four axes, so
                                        * the conv depends on pixels right/below and on the HIGHER-numbered channels */

/* ABI / build identification; also proves the library loaded. */
int iaf_abi_version(void);
const char* iaf_error_string(int code);
/* number of HIP devices visible; <0 on error */
int iaf_device_count(void);

/* ------------------------------------------------------------------------------------------
 * AR stack = the variables of one ar_multiconv2d (tf_utils/layers.py:158-166): depth_ar hidden
 * masked convs n_z->n_h, n_h->n_h ... and two output convs n_h->n_z ("layer_%d", "layer_out_%d").
 * depth_ar is len(n_h) (2 in tf_train.py:69; `depth_ar*[n_h2]` in models.py:92).
 * ------------------------------------------------------------------------------------------ */
typedef struct iaf_stack iaf_stack_t;

int iaf_stack_create(iaf_stack_t** out, int n_z, int n_h, int depth_ar, int variant);
int iaf_stack_destroy(iaf_stack_t* s);

/* Replaces get_conv_ar_mask + the weight-norm lines of conv2d (layers.py:134-141, 56-60):
 *   v = mask*V ; w = exp(g)[o] * v / sqrt(max(sum_{h,w,i} v^2, 1e-12))
 * and repacks w into MFMA fragment order (dead taps dropped).  V/g/b: arrays of depth_ar+2
 * device pointers in the order layer_0..layer_{d-1}, layer_out_0, layer_out_1; V is HWIO
 * [3,3,n_in,n_out] fp32 exactly as the reference variable.  Call again whenever weights change. */
int iaf_stack_prepare(iaf_stack_t* s, const float* const* V, const float* const* g, const float* const* b,
                      void* stream);

/* The same for MANY stacks in ONE launch: the weights of every IAF layer are known when a step starts, and one
 * 480-workgroup launch costs about as much as one 24-workgroup launch.  V/g/b are the concatenation, stack by
 * stack, of the (depth_ar+2) pointers iaf_stack_prepare takes.  The batch object owns pinned + device descriptor
 * tables; the run is stream-ordered and graph-capturable. */
typedef struct iaf_prep_batch iaf_prep_batch_t;
int iaf_prep_batch_create(iaf_prep_batch_t** out, iaf_stack_t* const* stacks, int n);
int iaf_prep_batch_run(iaf_prep_batch_t* b, const float* const* V, const float* const* g, const float* const* bias,
                       void* stream);
int iaf_prep_batch_destroy(iaf_prep_batch_t* b);

/* bytes of caller-provided scratch needed by the forward calls for a [B,*,H,W] problem */
size_t iaf_stack_workspace_bytes(const iaf_stack_t* s, int B, int H, int W);

/* Replaces ar_multiconv2d(name, x, context, n_h, n_out) (layers.py:158-166; call site
 * tf_train.py:69): z [B,n_z,H,W], context [B,n_h,H,W] -> m_raw, s_raw [B,n_z,H,W]. */
int iaf_ar_multiconv2d_forward(iaf_stack_t* s, const float* z, const float* context, float* m_raw, float* s_raw,
                               int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream);

/* The IAF step, tf_train.py:69-72 (models.py:281-285):
 *   m,s = 0.1*ar_multiconv2d(z, context);  z_new = (z-m)/exp(s);  logsd = s   (logqs += s)
 * z_new and logsd are [B,n_z,H,W]; z_new may alias z only if the caller no longer needs z. */
int iaf_step_forward(iaf_stack_t* s, const float* z, const float* context, float* z_new, float* logsd,
                     int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream);

/* Inverse of the IAF step: given the flow output z (what iaf_step_forward wrote to z_new) and the context, recover
 * z0 with (z0 - m(z0))/exp(s(z0)) == z, and logsd = s(z0).  The reference never inverts the flow (sample mode bypasses
 * it, tf_train.py:60-66) -- this is the density-evaluation direction SURVEY D3 / 8f-4 lists; it is checked by round trip
 * against iaf_step_forward.  Method: Jacobi sweeps z0 <- z*exp(s(z0)) + m(z0), each one full-rate forward launch (the one-launch
 * step kernel in MODE_INVERSE where a geometry is compiled).  Exactness: m, s at (pixel p, channel c) depend on z0 at the pixels
 * right of / below p and on the lower channels at p -- a DAG of depth H*W*n_z; a sweep moves every element one level down it, so
 * H*W*n_z sweeps are exact for ANY weights, and with the reference's 0.1 on m and s the map contracts to fp32 resolution in ~8
 * (tools/inverse_bench.py).  An anti-diagonal wavefront scan would be H*W*n_z dependent single-channel steps: latency, not work.
 *   max_sweeps  upper bound (and the exact count when tol == 0);
 *   tol > 0     stop when max|z0_new - z0_old| <= tol, tested ON THE DEVICE every check_every sweeps and after the last one: all
 *               max_sweeps launches are queued at once, a launch behind the converged one returns immediately (one-launch step
 *               kernels; the layer-by-layer kernels run on -- a sweep at the fixed point changes nothing), and a last small launch
 *               leaves the result in z0.  No host synchronisation inside the loop (round 5: one per test);
 *   sweeps_done / residual (optional): sweeps run, last tested max update (-1 if never tested).  Asking for them with tol > 0 costs
 *               ONE synchronisation at the end of the call (inside a stream capture: *sweeps_done = -1, nothing is read).
 * iaf_step_inverse_device: the same, never synchronises, capturable with any tol; d_sweeps_residual (optional, DEVICE, two words):
 * sweeps run as an int, the last residual as a float.
 * z0 must not alias z.  Workspace: iaf_stack_workspace_bytes. */
int iaf_step_inverse_device(iaf_stack_t* s, const float* z, const float* context, float* z0, float* logsd, int B, int H, int W,
                            void* workspace, size_t workspace_bytes, int max_sweeps, float tol, int check_every, void* stream,
                            unsigned* d_sweeps_residual);
int iaf_step_inverse(iaf_stack_t* s, const float* z, const float* context, float* z0, float* logsd, int B, int H, int W,
                     void* workspace, size_t workspace_bytes, int max_sweeps, float tol, int check_every, void* stream,
                     int* sweeps_done, float* residual);

/* ------------------------------------------------------------------------------------------
 * Training.  The reference never writes a backward pass: TF autodiff derives it from the graph
 * (opt.compute_gradients, tf_train.py:138; Theano: T.grad, graphy/misc/optim.py:102).  These entry points compute the
 * same gradients for the IAF step, for every variant.  With IAF_VARIANT_THEANO[_FLIPMASK] the (V, g, b) / (dV, dg, db)
 * slots carry (w, s, b): w and dw are OIHW [n_out][n_in+1][3][3] including the border-indicator channel (ar.py:288-296).
 * ------------------------------------------------------------------------------------------ */
/* on != 0: allocate the transposed weight packs; the next iaf_stack_prepare / iaf_prep_batch_run fills them */
int iaf_stack_set_training(iaf_stack_t* s, int on);
size_t iaf_stack_train_workspace_bytes(const iaf_stack_t* s, int B, int H, int W);
/* iaf_step_forward that keeps every hidden activation in `workspace` for iaf_step_backward */
int iaf_step_forward_train(iaf_stack_t* s, const float* z, const float* context, float* z_new, float* logsd,
                           int B, int H, int W, void* workspace, size_t workspace_bytes, void* stream);
/* Given dL/dz_new and dL/dlogsd (both [B,n_z,H,W]) and the SAME workspace the forward_train call filled:
 *   dz [B,n_z,H,W], dcontext [B,n_h,H,W], and for every conv (order as in iaf_stack_prepare) dV (HWIO, zero where
 *   the MADE mask is: the mask multiplies V in the graph, layers.py:57), dg, db -- all overwritten.
 * V/g are the reference variables again (weight-norm backward needs them). */
int iaf_step_backward(iaf_stack_t* s, const float* z, const float* context, const float* z_new, const float* logsd,
                      const float* dz_new, const float* dlogsd, float* dz, float* dcontext, const float* const* V,
                      const float* const* g, float* const* dV, float* const* dg, float* const* db, int B, int H, int W,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Deferred weight-norm backward.  A stack's own mask + weight-norm pass is a 24-workgroup launch (~21 us, bound by one CU's
 * address unit); a model has tens of stacks.  iaf_stack_set_defer_weightnorm(s, 1) makes iaf_step_backward /
 * iaf_posterior_block_backward stop after the weight-gradient reduction (dV/dg/db are NOT written; the reduced gradient
 * stays in that call's workspace, which must stay untouched); iaf_wn_bwd_batch_run then finishes ALL stacks of the batch in
 * one launch.  V/g/dV/dg/db: one pointer per conv over all stacks, in iaf_prep_batch_run order.
 * IAF_ERR_NOT_PREPARED if a stack of the batch has no deferred backward pending. */
typedef struct iaf_wn_bwd_batch iaf_wn_bwd_batch_t;
int iaf_stack_set_defer_weightnorm(iaf_stack_t* s, int on);
int iaf_wn_bwd_batch_create(iaf_wn_bwd_batch_t** out, iaf_stack_t* const* stacks, int n);
int iaf_wn_bwd_batch_run(iaf_wn_bwd_batch_t* b, const float* const* V, const float* const* g, float* const* dV,
                         float* const* dg, float* const* db, void* stream);
int iaf_wn_bwd_batch_destroy(iaf_wn_bwd_batch_t* b);

/* Training form of the posterior block (below): same outputs, keeps what the backward needs in `workspace`
 * (iaf_stack_train_workspace_bytes). */
int iaf_posterior_block_forward_train(iaf_stack_t* s, const float* qz_mean, const float* qz_logsd, const float* rz_mean,
                                      const float* rz_logsd, const float* pz_mean, const float* pz_logsd,
                                      const float* up_context, const float* down_context, const float* eps, float kl_min,
                                      float* z_out, float* kl_obj, float* kl_cost, int B, int H, int W, void* workspace,
                                      size_t workspace_bytes, void* stream);
/* Backward of tf_train.py:56-85 given dL/dz [B,n_z,H,W] (may be NULL = 0) and dL/dkl_obj [B]:
 *   dmean     = dL/d qz_mean  = dL/d rz_mean           dlogsd_q = dL/d qz_logsd = dL/d rz_logsd
 *   dpz_mean, dpz_logsd, dcontext (= dL/d up_context = dL/d down_context), and dV/dg/db as in iaf_step_backward.
 * Free bits: the gradient of max(mean_b sum_hw kl, kl_min) is gated per channel (tf_train.py:79-82); the gate is left
 * in the workspace by iaf_posterior_block_forward_train, so kl_min must be the value that call used. */
int iaf_posterior_block_backward(iaf_stack_t* s, const float* qz_mean, const float* qz_logsd, const float* rz_mean,
                                 const float* rz_logsd, const float* pz_mean, const float* pz_logsd, const float* eps,
                                 float kl_min, const float* z, const float* dz, const float* dkl_obj, float* dmean,
                                 float* dlogsd_q, float* dpz_mean, float* dpz_logsd, float* dcontext,
                                 const float* const* V, const float* const* g, float* const* dV, float* const* dg,
                                 float* const* db, int B, int H, int W, void* workspace, size_t workspace_bytes,
                                 void* stream);

/* The gradient exchange of a data-parallel training step: all-reduce(sum) over the ranks' flat fp32 gradient buffers
 * (tf_utils/common.py:83-86, average_grads: per-variable sum over the towers of tf_train.py:124-147, then 1/N -- the 1/N
 * rides in iaf_adamax_ema_step's grad_scale).  One process per GPU; RCCL (ncclAllReduce) over xGMI, bound at run time
 * (dlopen: the copy the process already carries, e.g. PyTorch's, else /opt/rocm/lib/librccl.so.1; iaf_comm_library() says
 * which).  Protocol: rank 0 calls iaf_comm_unique_id and hands the IAF_COMM_ID_BYTES bytes to every rank by any channel it
 * has; every rank calls iaf_comm_create (a collective: it returns once all `world` ranks have joined) with its rank and its
 * HIP device ordinal; iaf_allreduce_sum_f32 reduces buf[0..n) in place across the communicator, asynchronously on `stream`
 * (device pointer; the call is a collective -- every rank issues the same sequence of calls with the same n); buckets of
 * tens of MB keep the per-link-bound xGMI rings busy (iaf_amd/parallel.py cuts the flat buffer in completion order).
 * Errors: IAF_ERR_* for arguments, IAF_ERR_UNSUPPORTED if no RCCL can be loaded, 10000 + ncclResult_t for failures inside
 * RCCL (iaf_error_string knows them). */
#define IAF_COMM_ID_BYTES 128
typedef struct iaf_comm iaf_comm_t;
int iaf_comm_unique_id(void* id_out);
int iaf_comm_create(iaf_comm_t** out, const void* id, int rank, int world, int device);
int iaf_comm_size(const iaf_comm_t* c, int* rank, int* world);
int iaf_allreduce_sum_f32(iaf_comm_t* c, float* buf, size_t n, void* stream);
int iaf_comm_destroy(iaf_comm_t* c);
const char* iaf_comm_library(void);

/* Optimiser step on flat fp32 buffers of n elements: Adamax (tf_utils/adamax.py:40-56: slot "v" is the first moment,
 * slot "m" the infinity norm), on grad*grad_scale (grad_scale = 1/N folds the division of average_grads,
 * tf_utils/common.py:86, after an all-reduce(sum)), then the EMA of the new parameters (tf_train.py:157-158;
 * ema may be NULL).  In place. */
int iaf_adamax_ema_step(float* var, const float* grad, float* slot_m, float* slot_v, float* ema, size_t n, float lr,
                        float beta1, float beta2, float eps, float ema_decay, float grad_scale, void* stream);

/* Full posterior block, tf_train.py:56-85 (mode "train"): everything between down_conv1 and
 * the concat, i.e. posterior sample, logqs, IAF step, log-det accumulation, prior logps, KL and
 * free bits.  All [B,n_z,H,W] inputs NCHW; up_context/down_context [B,n_h,H,W]; eps is the
 * N(0,1) noise (an INPUT: parity is on identical eps).  Outputs: z [B,n_z,H,W], kl_obj [B],
 * kl_cost [B]; kl_elem (may be NULL) receives logqs-logps [B,n_z,H,W]. */
int iaf_posterior_block_forward(iaf_stack_t* s, const float* qz_mean, const float* qz_logsd,
                                const float* rz_mean, const float* rz_logsd, const float* pz_mean,
                                const float* pz_logsd, const float* up_context, const float* down_context,
                                const float* eps, float kl_min, float* z_out, float* kl_obj, float* kl_cost,
                                float* kl_elem, int B, int H, int W, void* workspace, size_t workspace_bytes,
                                void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise pieces of tf_utils/distributions.py, exposed for the wrappers
 * ------------------------------------------------------------------------------------------ */
/* DiagonalGaussian.sample (distributions.py:20-21): out = mean + exp(0.5*logvar)*noise */
int iaf_gaussian_sample(const float* mean, const float* logvar, const float* noise, float* out, size_t n,
                        void* stream);
/* ... with a log standard deviation where the reference writes `2 * logsd` in place (tf_train.py:56-57, rand.py:81-86) */
int iaf_gaussian_sample_logsd(const float* mean, const float* logsd, const float* noise, float* out, size_t n, void* stream);
int iaf_gaussian_logps_logsd(const float* mean, const float* logsd, const float* sample, float* out, size_t n, void* stream);
/* gaussian_diag_logps (distributions.py:10) */
int iaf_gaussian_logps(const float* mean, const float* logvar, const float* sample, float* out, size_t n,
                       void* stream);

/* compute_lowerbound (distributions.py:55-62) with k>=1: log_pxz, sum_kl [n*k] -> out [n] */
int iaf_compute_lowerbound(const float* log_pxz, const float* sum_kl, float* out, int n, int k, void* stream);
/* Streaming form for k = 10^4 (BASELINE config 5): state = (run_max[n], run_sum[n]); feed chunks of
 * k_chunk weights per image, then finalize.  Equal to compute_lowerbound on the concatenation. */
int iaf_lowerbound_stream_init(float* run_max, float* run_sum, int n, void* stream);
int iaf_lowerbound_stream_update(float* run_max, float* run_sum, const float* log_pxz, const float* sum_kl,
                                 int n, int k_chunk, void* stream);
/* Free bits on given KL elements (tf_train.py:77-85): kl_cost[b] = sum_{c,h,w} kl; kl_obj[b] = sum_c max(mean_b sum_{h,w} kl,
 * kl_min) (kl_min > 0) or kl_cost[b].  The Theano objective adds that per-layer value once, as a scalar (models.py:458-461).
 * scratch: B*C floats. */
int iaf_kl_free_bits(const float* kl_elem, float* kl_obj, float* kl_cost, int B, int C, int HW, float kl_min, float* scratch,
                     void* stream);
/* kl[i] = logq0[i] + logdet[i] - logp[i] (models.py:175, 298, 328), for posteriors whose three terms come from separate
 * launches ('up_iaf2_nl': IAF step in the bottom-up pass, prior density in the top-down pass) */
/* iaf_kl_free_bits + the per-channel gate [C] of the free bits: 1 where mean_b sum_hw kl > kl_min (where max() passes the
 * gradient, tf_train.py:79-80 / models.py:460-461), else 0 */
int iaf_kl_free_bits_gate(const float* kl_elem, float* kl_obj, float* kl_cost, float* gate, int B, int C, int HW, float kl_min,
                          float* scratch, void* stream);
/* The elementwise halves of the backward of the Theano layer's 'up_iaf2_nl' posterior (models.py:168-176, 201-210, 295-298,
 * 454-466) on either side of iaf_step_backward.  kl = logq0 + logdet - logp(z); G = d obj / d kl = gate[c] * gscale (free bits;
 * gate from iaf_kl_free_bits_gate) or dko[b] (gate = NULL).  d_h, d_up: [B, n_h + n_z, HW] in concat([h_det, z]) order, d_up
 * may be NULL (zeros).
 *   pre:  dz_tot = d_h[z part] + G (z - pz_mean) exp(-2 pz_logsd) [+ d_up[z part]],  G (expanded, [B, n_z, HW]),
 *         d_down_conv1 [B, n_h + 2 n_z, HW] = [d_h[h_det part] | -G dlt e2 | G (1 - dlt^2 e2)]
 *   post: d_up_conv1 [B, 2 n_h + 2 n_z, HW] = [d_up[h_det part] | dz0 | dz0 (z0 - qz_mean) - G | dctx] */
int iaf_up_iaf2_backward_pre(const float* z, const float* pz_mean, const float* pz_logsd, const float* d_h, const float* d_up,
                             const float* gate, float gscale, const float* dko, float* dz_tot, float* G, float* d_down_conv1,
                             int B, int n_h, int n_z, int HW, void* stream);
int iaf_up_iaf2_backward_post(const float* dz0, const float* z0, const float* qz_mean, const float* G, const float* dctx,
                              const float* d_up, float* d_up_conv1, int B, int n_h, int n_z, int HW, void* stream);
int iaf_kl_combine(const float* logq0, const float* logdet, const float* logp, float* kl, size_t n, void* stream);
/* out[j] = sum_i mat[i*n + j], i < m: the running `kl_cost += cur_cost` over a model's layers (tf_train.py:198-200) when
 * every layer wrote its [n] KL costs into one row of a [m, n] matrix */
int iaf_colsum(const float* mat, float* out, int m, int n, void* stream);
int iaf_lowerbound_stream_finalize(const float* run_max, const float* run_sum, float* out, int n, int k_total,
                                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Tuning / introspection (not part of the reference surface)
 * ------------------------------------------------------------------------------------------ */
/* Override the launch shape of GEMM layer `layer` (0..depth_ar): co-tiles per wave, pixel tiles
 * per workgroup, waves along co, split-K factor.  Returns IAF_ERR_UNSUPPORTED if no such kernel. */
int iaf_stack_set_tuning(iaf_stack_t* s, int layer, int nt, int pxt, int wco, int ks);

/* Arithmetic of the forward masked convs (the reference runs cuDNN fp32 convs, tf_utils/layers.py:64):
 *   IAF_PRECISION_BF16X3 (default)  every fp32 operand split into three bf16 parts, the six leading part-products
 *       accumulated in fp32 on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16): fp32-grade results (error vs an fp64
 *       evaluation not larger than the fp32 chain's) at 2.7x the matrix-core rate of
 *   IAF_PRECISION_F32               the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), bit-equal to an fmaf chain.
 * Layers the bf16x3 kernels do not cover (c_in not a multiple of 32, co-tile counts without a compiled shape, the data
 * gradients, the plain 9-tap convs) or do not speed up (fewer than ~4096 pixels per launch, unless autotuned or a shape
 * is pinned) run the fp32 kernel in either mode; iaf_stack_get_precision reports what GEMM layer `layer` will run at a
 * given problem size.  Outputs of the two modes differ by fp32 round-off only; each is deterministic. */
/*   IAF_PRECISION_F16X2 (round 6)   as BF16X3, except that the one-launch step kernels compiled for it (TF statement, n_z = 32,
 *       n_h = 160, depth_ar = 2, images 16 and 8 pixels wide: the BASELINE run) split every operand into TWO fp16 planes, x = hi +
 *       lo' 2^-11 with hi = fp16(x), lo' = fp16((x - hi) 2^11), and accumulate THREE part-products per K step on
 *       v_mfma_f32_16x16x32_f16 -- hi hi' in one fp32 accumulator, hi lo' + lo' hi' in a second one that is scaled by 2^-11 where
 *       the sums meet: half the matrix-core instructions and two thirds of the pack and LDS bytes of bf16x3, 22 significand bits
 *       per operand, error against an fp64 evaluation within the bound tests/test_hip_dynamic_range.py holds bf16x3 to.  What fp16
 *       does not have is fp32's exponent range: an operand (weight or activation) beyond 65504 makes the launch's outputs inf / NaN
 *       and raises a word in mapped host memory; the stack's NEXT call returns IAF_ERR_RANGE once and the stack runs the bf16x3
 *       kernels from then on (iaf_stack_range_errors reads the word; iaf_stack_set_precision(s, F16X2) re-arms).  Every other
 *       launch of such a stack (other geometries, layer-by-layer kernels, backward) is the BF16X3 one. */
#define IAF_PRECISION_F32 0
#define IAF_PRECISION_BF16X3 1
#define IAF_PRECISION_F16X2 2
int iaf_stack_set_precision(iaf_stack_t* s, int precision);
/* *errors = the range word of an IAF_PRECISION_F16X2 stack (0: none; bit 0: an activation, bit 1: a weight); synchronises the device */
int iaf_stack_range_errors(const iaf_stack_t* s, unsigned* errors);
/* 1 if the one-launch step of the stack at this size would run a two-plane fp16 kernel now, else 0 */
int iaf_stack_step_is_f16(const iaf_stack_t* s, int B, int H, int W);
int iaf_stack_get_precision(const iaf_stack_t* s, int layer, int B, int H, int W);
/* launch shape of the bf16x3 kernel for GEMM layer `layer`: co tiles per wave, pixel tiles per wave, waves along
 * pixels, K-slice waves, co groups sharing one staged activation tile (nt = 0 restores the automatic choice) */
int iaf_stack_set_tuning_bf3(iaf_stack_t* s, int layer, int nt, int ppw, int pxt, int ks, int wco);
/* Kernel-family / launch-shape search for one problem size -- the counterpart of the cuDNN algorithm search behind the
 * reference's tf.nn.conv2d (tf_utils/layers.py:64): times every GEMM layer as the exact-fp32 kernel and as every compiled
 * bf16x3 shape (`reps` back-to-back launches each, on the caller's buffers) and remembers the winner for (B*H*W, W).
 * chosen[l] (optional, depth_ar+1 entries) = 0 for the fp32 kernel or nt*10000 + ppw*1000 + pxt*100 + ks*10 + wco; us[l] its time.
 * Synchronises the stream: call it before capturing a graph.  Results of later launches are unaffected beyond fp32
 * round-off (both families meet the same parity bar). */
/* The first masked conv of a stack (c_in = n_z = 32: five K steps) can run INSIDE the second one's bf16x3 kernel: its
 * output goes straight into that kernel's LDS tile (recomputed on the halo) -- one launch and one HBM round trip less per
 * IAF step.  mode 0 = never, 1 = whenever the kernels allow it, 2 (default) = only where iaf_stack_autotune measured it
 * clearly faster for the problem size (on MI355X at the BASELINE sizes the fused prologue is a latency chain that costs
 * more than the launch it saves, docs/LAB_NOTEBOOK_r01-r03.md 4.8, so the default in practice runs separate launches).  TF statement, bf16x3 precision, n_z = 32 only; results equal the unfused
 * path's bit for bit (the same products in the same order).  In iaf_stack_autotune's report a fused pair shows as
 * chosen[0] = -1 and chosen[1] = the fused kernel's shape. */
int iaf_stack_set_fuse_first(iaf_stack_t* s, int mode);
/* The whole IAF step (tf_train.py:69-72, or the posterior block's tf_train.py:56-75) as ONE launch: a workgroup owns R
 * full-width rows of one image and computes every masked conv of the stack for them, hidden activations in LDS, the
 * halo rows recomputed (the masked convs look only right and below, so no other workgroup is involved).  mode 1 (default):
 * wherever a compiled geometry covers the problem -- any of the three statements of the operator (the Theano one runs on the
 * image rotated by 180 degrees, where its taps are the TF ones), bf16x3 precision, (n_h, n_z, depth_ar) = (160, 32, 2)
 * (64, 32, 1) or (64 / 128 / 192, 64, 4) where its LDS regions fit, images 16, 8 or 4 pixels wide; everything else, and any stack with a pinned per-layer launch shape
 * (iaf_stack_set_tuning*, fuse_first = 1), takes the layer-by-layer path.  mode 0: never; mode 2: wherever a geometry
 * covers it, whatever the size rule or a measurement says.  Same arithmetic as the
 * layer-by-layer bf16x3 kernels in a different summation order: results agree to fp32 round-off, not bit for bit.
 * Without a measurement the step runs fused at 16-pixel rows, and at 8-pixel rows while B*H < 1024 (every workgroup
 * streams the whole weight set out of L2: at large batches the layer-by-layer kernels amortise it better);
 * iaf_stack_autotune times both paths for its problem size and records the winner (chosen[] = -2 for every layer, us[last] =
 * the step).  iaf_stack_step_is_fused: rows per workgroup the step would run with at this size, 0 if it would not run fused. */
int iaf_stack_set_fuse_step(iaf_stack_t* s, int mode);
/* Which packed copies of the weights the prep launches (iaf_stack_prepare, iaf_prep_batch_run) keep up to date: the default
 * is both -- the fp32 fragment pack of the exact-fp32 kernels AND the bf16x3 pack (10 B written per live weight).  A stack
 * whose every launch runs on the bf16 matrix cores (the one-launch step, or bf16x3 layer-by-layer kernels: inference at the
 * BASELINE sizes) can drop the fp32 pack: packs = IAF_PACK_BF16X3 (6 B per weight, -40 % of the prep launch's stores).  A
 * launch that would need the missing pack then fails with IAF_ERR_NOT_PREPARED instead of reading stale weights; training
 * stacks and stacks with a layer outside the bf16x3 kernels (c_in not a multiple of 32) refuse with IAF_ERR_UNSUPPORTED.
 * The reference re-derives w = exp(g) * mask*V / ||mask*V|| every step (layers.py:56-60); which layout it is left in is
 * this engine's business. */
/* IAF_PRECISION_F16X2 stacks also write IAF_PACK_F16X2, the two-plane fp16 pack (4 B per weight); one whose every launch is an F16
 * step kernel can keep ONLY that: packs = IAF_PACK_F16X2. */
#define IAF_PACK_F32 1
#define IAF_PACK_BF16X3 2
#define IAF_PACK_F16X2 4
int iaf_stack_set_packs(iaf_stack_t* s, int packs);
int iaf_stack_step_is_fused(const iaf_stack_t* s, int B, int H, int W);
/* The one-launch step at 16-pixel rows (the BASELINE geometry and the deep stacks of config 3, all three statements of the
 * operator) does not recompute the rows its row blocks share: the block below hands its first hidden rows to the block above
 * through device memory, inside the launch.
 *   Order.  HIP promises no dispatch order, so the kernel makes its own: a workgroup takes a ticket from one of eight work lists
 *   (its XCD's first: eight heads are cheap to pull, one is not) whose items are dealt out bottom row block first, and so only
 *   ever waits for the holder of a LOWER ticket -- a workgroup that is already running.  Results do not depend on dispatch
 *   order, timing or workgroup -> XCD placement.
 *   Hand-over.  The data is the flag: between launches every 8-byte piece of the row buffer holds a "not there yet" pattern (a
 *   bf16 NaN payload no arithmetic produces); the producer stores its row with agent-scope (write-through) 8-byte stores, the
 *   consumer reads the row with agent-scope loads until no piece shows the pattern, and puts the pattern back.
 *   State.  Rows and list heads live in a set of buffers the stack owns PER STREAM (allocated on a stream's first such launch
 *   outside a stream capture): calls on different streams are independent.  A stream capture allocates nothing: it uses its
 *   stream's set if a warm-up launch on that stream created one, else it TAKES OVER the stack's newest set that is large enough
 *   AND that no graph captured on another stream already names (such a set stays with its graph: two graphs on two streams never
 *   share rows or counters; a capture that finds none runs the recomputing kernel)
 *   (the set changes owner: the stream it was warmed up on allocates a new one on its next eager launch, so replays of the graph
 *   and those launches share nothing; only launches of that stream still in flight from before the capture use the old set -- a
 *   graph is not replayed before its capture has ended, synchronise that stream before the first replay), else the recomputing
 *   kernel.  Two graphs captured on one stream share that stream's set: replay them in stream order, not concurrently.
 *   Failure.  Every wait is bounded (seconds).  A wait that gives up fills the rows it waited for with NaN -- the launch's
 *   outputs then carry NaN where they depend on them (numerical failure = NaN for the caller's loop, tf_train.py:283-285) --
 *   and raises the stack's error word; launches already queued on those buffers import NaN without looking (the buffers can no
 *   longer be trusted); the NEXT call on the stack returns IAF_ERR_EXCHANGE once (no synchronisation needed: the word sits in
 *   mapped host memory) and the stack goes on with the recomputing kernels until iaf_stack_set_halo_exchange(s, 1) re-arms
 *   the exchange.  A captured hipGraph holds the exchange-form launch itself: REPLAYS of it keep importing NaN (no host code runs that
 *   could switch kernels) until the stack is re-armed AND the graph is captured again -- a training loop that replays graphs should read
 *   iaf_stack_exchange_errors (or, from Python, CVAE1.exchange_errors()) when its loss turns NaN.
 *   (The two-plane fp16 kernels of IAF_PRECISION_F16X2 keep sets of their own, with the pattern 0xfdfffdff: a pair of SIGNALLING fp16 NaNs.)
 *   The pattern: every dword of the row buffers holds 0xffbfffbf between launches, a pair of SIGNALLING bf16 NaNs.  The rows carry hidden
 *   activations, results of arithmetic, and arithmetic returns quiet NaNs only: a caller's NaN -- whatever its payload -- travels as data.
 * iaf_stack_exchange_errors: *errors = the error word (0 = never gave up; it synchronises the device so that every launch so
 * far is accounted for).  IAF_FUSE_XCH=0 in the environment keeps the recomputing kernel. */
int iaf_stack_exchange_errors(const iaf_stack_t* s, unsigned* errors);
/* on = 0: this stack's one-launch step recomputes its halo rows (geometries that exist only in the exchange form run layer by
 * layer then); on = 1 (default): exchange where it applies.  Either way the exchange state starts afresh (buffers re-armed,
 * error words cleared; synchronises the device). */
int iaf_stack_set_halo_exchange(iaf_stack_t* s, int on);
/* Test knobs of the exchange (OR of): 1 = work lists chosen by a hash of the workgroup index instead of its XCD (lists run dry,
 * workgroups take from other lists), 2 = workgroups delay their ticket by pseudo-random amounts (tickets out of dispatch
 * order), 16 = the posterior block's free-bits reductions by the separate finish launch instead of the step launch's
 * last workgroup (to compare the two), 8 = fault injection: the bottom row block of image 0 never hands over its first hidden row and waits are short, so
 * the block above it gives up (pair form: one half is never handed over).  32 = the PAIR form of the step at 8-pixel rows
 * (iaf_stack_step_pairs; opt-in: on MI355X at B = 32 it is slower than the one-row kernel, profiles/r05/experiments/pair_form.txt).
 * 0 = production.  IAF_XCH_DEBUG=<bits 1|2|16> in the environment: every stack is created with
 * these bits set (the whole GPU suite under a scrambled hand-over order: profiles/r04/pytest_gpu_scrambled.txt). */
int iaf_stack_set_halo_exchange_debug(iaf_stack_t* s, unsigned knobs);
/* 1 if the step at this size runs as one launch in the halo-exchange form, else 0 */
int iaf_stack_step_exchanges(const iaf_stack_t* s, int B, int H, int W);
/* 1 if the step at this size runs as one launch in the PAIR form (opt-in: knob 32 of iaf_stack_set_halo_exchange_debug or
 * IAF_FUSE_PAIR=1; 8-pixel rows: two workgroups per (image, two rows), each computing
 * half of the last hidden layer's channels and of the output pair from half of their weights, the halves of the last hidden layer
 * swapped through the stack's exchange buffers -- same ordering, hand-over, failure behaviour (IAF_ERR_EXCHANGE) and switch
 * (iaf_stack_set_halo_exchange) as the halo exchange; iaf_stack_step_is_fused reports 2 rows), else 0.  Replaces
 * /root/reference/tf_utils/layers.py:158-166 + tf_train.py:69-72 at [B, n_z, 8, 8] like every other form of the step. */
int iaf_stack_step_pairs(const iaf_stack_t* s, int B, int H, int W);
int iaf_stack_autotune(iaf_stack_t* s, const float* z, const float* context, float* z_new, float* logsd, int B, int H, int W,
                       void* workspace, size_t workspace_bytes, int reps, void* stream, int* chosen, float* us);
/* Per-kernel timing with HIP events on the launch stream: every launch of GEMM layer `layer` is
 * bracketed by an engine-owned event pair (up to max_samples; layer < 0 disables).  Not usable
 * during stream capture.  iaf_stack_profile_read synchronises the recorded events, writes the
 * elapsed milliseconds of each sample to ms_out[0..*n_out) and resets the sample counter. */
int iaf_stack_profile_enable(iaf_stack_t* s, int layer, int max_samples);
int iaf_stack_profile_read(iaf_stack_t* s, float* ms_out, int capacity, int* n_out);
/* Roofline timing: runs one full iaf_step (so every layer has valid inputs), then launches GEMM layer `layer`
 * `reps` times back to back between ONE pair of HIP events on `stream`; *avg_ms = elapsed / reps (includes the
 * ~1 us inter-launch gap, excludes event/dispatch latency).  Synchronises the stream.  layer = -1 times the FUSED launch
 * (layers 0+1 in one kernel, iaf_stack_set_fuse_first), layer = -2 the whole step as one launch (iaf_stack_set_fuse_step);
 * IAF_ERR_UNSUPPORTED if the stack would not run that way here.  iaf_stack_profile_enable(layer = -2) brackets the
 * one-launch step the same way. */
int iaf_step_time_layer(iaf_stack_t* s, int layer, const float* z, const float* context, float* z_new, float* logsd,
                        int B, int H, int W, void* workspace, size_t workspace_bytes, int reps, void* stream,
                        float* avg_ms);
/* dev tool: every workgroup of GEMM layer `layer` writes 8 s_memtime stamps (kernel start, tile loads
 * issued, tile staged, steady loop done, K loop done, end) to buf[wg*8..]; buf = NULL disables. */
int iaf_stack_set_debug(iaf_stack_t* s, int layer, void* buf);
/* algorithmic work of one iaf_step_forward call: live (mask-aware) FLOPs and fused bytes
 * (4*(3*n_z+n_h) B/px + weight bytes) */
int iaf_step_work(const iaf_stack_t* s, int B, int H, int W, double* live_flops, double* dense_flops,
                  double* bytes);
/* the same for one GEMM layer (0..depth_ar): live/dense FLOPs of one launch and the bytes that
 * launch must move at least once (activations in + out, packed weights, bias, context) */
int iaf_layer_work(const iaf_stack_t* s, int layer, int B, int H, int W, double* live_flops, double* dense_flops,
                   double* bytes);

/* ------------------------------------------------------------------------------------------
 * Plain weight-normed 3x3 convs either side of the IAF step: IAFLayer's up_conv1 / up_conv3 /
 * down_conv1 / down_conv2, i.e. tf_utils/layers.py:31-64 (conv2d, mask=None, stride (1,1), pad SAME, non-init
 * branch) as called from tf_train.py:36-44, 52-54, 87-94 with the surrounding elementwise work fused:
 *   in  = concat_channels(x, x2)            (x2 optional; tf_train.py:87 h = concat(1, [z, h_det]))
 *   in  = elu(in) if elu_input               (tf_train.py:35,40,52,88)
 *   y   = conv2d(in, exp(g) * l2_normalize(V, [0,1,2])) + b                       (layers.py:60-64)
 *   n_outs == 1 and residual:  outs[0] = residual + 0.1 * y                       (tf_train.py:44,94)
 *   otherwise:  outs[k] = channels [sum(out_channels[:k]), +out_channels[k]) of y (the split of tf_train.py:37,54)
 * All tensors are contiguous NCHW fp32 device memory.  V is HWIO [3,3,n_in,n_out]; g, b are [n_out].
 * Channel counts that are multiples of 16 run on the MFMA kernel (concat / split points must then be multiples of 4,
 * else IAF_ERR_UNSUPPORTED); any other counts run on a direct-conv fallback.  The downsampling variants (stride 2,
 * deconv2d, resize_nearest_neighbor) are not covered.
 * ------------------------------------------------------------------------------------------ */
typedef struct iaf_conv3x3 iaf_conv3x3_t;
#define IAF_CONV3X3_MAX_OUTS 6

int iaf_conv3x3_create(iaf_conv3x3_t** out, int n_in, int n_out);
/* one MADE-masked conv on its own: ar_conv2d(name, x, num_filters, zerodiagonal=...) (tf_utils/layers.py:144-154),
 * same object type and calls as the plain conv.  IAF_ERR_NOT_MULTIPLE unless n_in | n_out or n_out | n_in (:116). */
int iaf_conv3x3_create_masked(iaf_conv3x3_t** out, int n_in, int n_out, int zerodiagonal);
/* One masked conv of the THEANO path on its own: N.ar.conv2d(name, n_in, n_out, (3,3), zerodiagonal, flipmask, w=w)
 * (graphy/nodes/ar.py:200-375, used by posteriors 'up_iaf1' / 'down_iaf2' ..., models.py:53-55, 73-79).  prepare() then takes
 * V = <name>_w OIHW [n_out][n_in+1][3][3] (border-indicator channel last), g = <name>_s (applied as exp(3 s)), b = <name>_b.
 * Channels must be multiples of 16.  Forward only. */
int iaf_conv3x3_create_masked_theano(iaf_conv3x3_t** out, int n_in, int n_out, int zerodiagonal, int flipmask);
int iaf_conv3x3_destroy(iaf_conv3x3_t* c);
/* weight normalisation + packing (one launch); call again whenever V/g/b change */
int iaf_conv3x3_prepare(iaf_conv3x3_t* c, const float* V, const float* g, const float* b, void* stream);
int iaf_conv3x3_forward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                        const float* residual, float* const* outs, const int* out_channels, int n_outs, int B, int H,
                        int W, void* stream);
/* weight prep of n plain convs (not the masked kind, MFMA-path channel counts) in ONE launch; V/g/b: n pointers each,
 * in the order of `convs`.  The counterpart of iaf_prep_batch_* for the convs around the IAF step. */
typedef struct iaf_conv3x3_prep_batch iaf_conv3x3_prep_batch_t;
int iaf_conv3x3_prep_batch_create(iaf_conv3x3_prep_batch_t** out, iaf_conv3x3_t* const* convs, int n);
int iaf_conv3x3_prep_batch_run(iaf_conv3x3_prep_batch_t* b, const float* const* V, const float* const* g,
                               const float* const* bias, void* stream);
int iaf_conv3x3_prep_batch_destroy(iaf_conv3x3_prep_batch_t* b);
/* ---- training of the plain convs: what opt.compute_gradients (tf_train.py:138) derives for layers.py:52-64 ----
 * iaf_conv3x3_set_training(1) allocates the transposed weight pack; the NEXT prepare fills it.  MFMA-path plain convs only.
 * iaf_conv3x3_backward: forward was  y = conv(act(concat(x, x2)), exp(g) V/||V||) + b  with y handed out as n_dys split
 * tensors.  Inputs: the forward inputs again, the gradients dys[k] ([B, dy_channels[k], H, W]) of the split outputs and
 * a common factor dy_scale (0.1 when the forward folded `input + 0.1*y`).  Outputs: dxs[k] = gradient w.r.t. the forward
 * input(s) split like the forward concat ([x] or [x | x2]), = [dx_residual +] act'(.) * (W^T dY)  (n_dxs = 0 skips it;
 * dx_residual only with n_dxs == 1: the `input +` branch of tf_train.py:44,94);  dV [3,3,n_in,n_out], dg, db [n_out]. */
int iaf_conv3x3_set_training(iaf_conv3x3_t* c, int on);
size_t iaf_conv3x3_train_workspace_bytes(const iaf_conv3x3_t* c, int B, int H, int W);
int iaf_conv3x3_backward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                         const float* const* dys, const int* dy_channels, int n_dys, float dy_scale, float* const* dxs,
                         const int* dx_channels, int n_dxs, const float* dx_residual, const float* V, const float* g,
                         float* dV, float* dg, float* db, int B, int H, int W, void* workspace, size_t workspace_bytes,
                         void* stream);
/* launch-shape search for the data gradient of iaf_conv3x3_backward (same arguments, + reps): times whole backward calls
 * per candidate shape of the transposed problem and pins the fastest; outputs end up as a normal backward call leaves
 * them.  Synchronises the stream; not capturable. */
int iaf_conv3x3_autotune_backward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                  const float* const* dys, const int* dy_channels, int n_dys, float dy_scale,
                                  float* const* dxs, const int* dx_channels, int n_dxs, const float* dx_residual,
                                  const float* V, const float* g, float* dV, float* dg, float* db, int B, int H, int W,
                                  void* workspace, size_t workspace_bytes, int reps, void* stream, int* best_shape,
                                  float* best_us);
/* the same deferral for the plain convs (needs set_training first): iaf_conv3x3_backward keeps the reduced dW / db
 * partials inside the object and leaves dV/dg/db to one iaf_conv3x3_wn_bwd_batch_run over all convs of the model */
typedef struct iaf_conv3x3_wn_bwd_batch iaf_conv3x3_wn_bwd_batch_t;
int iaf_conv3x3_set_defer_weightnorm(iaf_conv3x3_t* c, int on);
int iaf_conv3x3_wn_bwd_batch_create(iaf_conv3x3_wn_bwd_batch_t** out, iaf_conv3x3_t* const* convs, int n);
int iaf_conv3x3_wn_bwd_batch_run(iaf_conv3x3_wn_bwd_batch_t* b, const float* const* V, const float* const* g,
                                 float* const* dV, float* const* dg, float* const* db, void* stream);
int iaf_conv3x3_wn_bwd_batch_destroy(iaf_conv3x3_wn_bwd_batch_t* b);
/* launch shape override (nt = 0 restores the automatic choice); see iaf_stack_set_tuning */
int iaf_conv3x3_set_tuning(iaf_conv3x3_t* c, int nt, int pxt, int wco, int ks);
/* dev tool (tools/conv_stamps.py): buf = device array [workgroups][8] of u64, `bytes` long, that THIS conv's bf16x3 launches fill with
 * cycle stamps (start, weight ring primed, -, tile staged, K loop done, partial sums exchanged, stores issued) -- a launch whose grid
 * does not fit `bytes` leaves it alone; NULL switches it off */
int iaf_conv3x3_set_debug(iaf_conv3x3_t* c, void* buf, size_t bytes);
/* arithmetic of the conv, as iaf_stack_set_precision: IAF_PRECISION_BF16X3 (plain convs with c_in % 32 == 0 from 4096 pixels on, or as
 * iaf_conv3x3_autotune measured, run the split-product kernels on three bf16 planes) or IAF_PRECISION_F32 (the exact-fp32 MFMA kernel
 * always; masked single convs run it regardless).  iaf_conv3x3_runs_bf16x3: 1 if a forward call at this size would run a split-product
 * kernel (bf16 or fp16 planes). */
int iaf_conv3x3_set_precision(iaf_conv3x3_t* c, int precision);
/* IAF_PRECISION_F16X2 for a plain conv (THE DEFAULT where c_in % 32 == 0; from 2048 pixels on): its stride-1 forward launches split the
 * operands into two fp16 planes as the one-launch step does (iaf_stack_set_precision); the strided forms and the weight gradient stay on
 * bf16 planes.  Range protocol of the forward: an operand beyond 65504 -> inf / NaN outputs, the NEXT iaf_conv3x3_forward returns
 * IAF_ERR_RANGE once and the conv runs bf16x3 from then on; *errors = its range word (synchronises).
 * The DATA gradient of such a conv (iaf_conv3x3_backward) runs on two fp16 planes as well, with no range of its own: every workgroup
 * scales the dY tile it stages by the power of two that puts the tile's largest magnitude into [2^13, 2^14) and its sums by the inverse
 * -- gradients of any fp32 magnitude, errors relative to the tile's largest element (IAF_DGRAD_F16=0 in the environment: bf16 planes). */
int iaf_conv3x3_range_errors(const iaf_conv3x3_t* c, unsigned* errors);
/* Which packs the prep launches of a PLAIN conv keep up to date (IAF_PACK_* as iaf_stack_set_packs; default: all three, 14 bytes written
 * per weight).  A conv of a model runs at one size, i.e. reads one pack: the two-plane fp16 one (iaf_conv3x3_runs_f16x2), else the
 * bf16x3 one (iaf_conv3x3_runs_bf16x3; also what iaf_conv3x3_forward_stride2 reads), else the fp32 one.  A forward launch whose pack is
 * not kept returns IAF_ERR_NOT_PREPARED; after a range failure of an fp16-only conv every pack is kept again from the next prepare on
 * (IAF_ERR_RANGE once, then IAF_ERR_NOT_PREPARED until that prepare).  Training convs, masked convs, deconvs (iaf_conv3x3_prepare_deconv)
 * and generic channel counts keep every pack: IAF_ERR_UNSUPPORTED for anything else. */
int iaf_conv3x3_set_packs(iaf_conv3x3_t* c, int packs);
int iaf_conv3x3_runs_bf16x3(iaf_conv3x3_t* c, int B, int H, int W);
/* 1 if that launch would be the two-plane fp16 one */
int iaf_conv3x3_runs_f16x2(iaf_conv3x3_t* c, int B, int H, int W);
/* times every compiled launch shape with `reps` back-to-back forwards on the caller's buffers (same arguments as
 * iaf_conv3x3_forward; outputs end up holding the forward result), pins the fastest as if by set_tuning, and reports it
 * (best_shape[4] = nt,pxt,wco,ks; best_us per call; both optional).  With IAF_PRECISION_BF16X3 (the default) a plain conv
 * with c_in % 32 == 0 is also timed on the bf16 matrix cores (bf16x3 split products, fp32-grade: iaf_conv_bf3.hpp with all 9
 * taps) in every compiled shape, and the overall winner is pinned: best_shape = (-nt, ppw, wco, ks) then (iaf_conv3x3_set_tuning
 * accepts that form back).  Synchronises the
 * stream; not capturable. */
int iaf_conv3x3_autotune(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                         const float* residual, float* const* outs, const int* out_channels, int n_outs, int B, int H,
                         int W, int reps, void* stream, int* best_shape, float* best_us);
/* FLOPs (2*9*n_in*n_out per pixel) and minimum bytes (activations in+out, V/g/b once) of one forward call */
int iaf_conv3x3_work(const iaf_conv3x3_t* c, int B, int H, int W, double* flops, double* bytes);

/* ------------------------------------------------------------------------------------------
 * Data-dependent initialisation, the init=True branch of conv2d (tf_utils/layers.py:38-51).  The caller runs the conv
 * with g = 0, b = 0 (so the prepared weights are l2_normalize(mask*V), :44-45) to get x_init [B,C,H,W]; this call then
 * computes, per channel over (N,H,W):  scale = init_scale / sqrt(var + 1e-10),  g = log(scale)/3,  b = -mean*scale,
 * y = scale * (x_init - mean) [+ add]  (:47-51; y may be NULL or alias x_init; `add`, optional, is the context that
 * ar_multiconv2d adds to its first layer, layers.py:163-164).
 * ------------------------------------------------------------------------------------------ */
int iaf_datainit_normalize(const float* x_init, const float* add, float* y, float* g, float* b, int B, int C, int HW,
                           float init_scale, void* stream);

/* discretized_logistic(mean, logscale, binsize, sample) (tf_utils/distributions.py:28-32; tf_train.py:210):
 * out[b] = sum over the n_per_row elements of row b.  logscale: one device scalar (logscale_is_scalar, the reference's
 * dec_log_stdv) or a tensor shaped like mean. */
int iaf_discretized_logistic(const float* mean, const float* logscale, int logscale_is_scalar, const float* sample,
                             float* out, int B, size_t n_per_row, float binsize, void* stream);

/* ------------------------------------------------------------------------------------------
 * The DOWNSAMPLING IAFLayer (tf_train.py:33,42-43,89-91): stride-2 up_conv1, resize_nearest_neighbor, deconv2d
 * ------------------------------------------------------------------------------------------ */
/* 2x resampling of an NCHW tensor; H, W = size of the SMALLER of the two tensors.
 *   DOWN_EVEN  dst[i,j] = src[2i,2j]      == resize_nearest_neighbor(x, 0.5)   (tf_utils/layers.py:169-175, tf_train.py:43)
 *   DOWN_ODD   dst[i,j] = src[2i+1,2j+1]  the outputs conv2d(stride=[2,2], SAME) keeps of the stride-1 conv (tf_train.py:33,36)
 *   UP_NEAREST dst[y,x] = src[y/2,x/2]    == resize_nearest_neighbor(x, 2)     (tf_train.py:90)
 *   UP_ZERO_ODD dst[2i+1,2j+1] = src[i,j], 0 elsewhere: the zero-inserted input of conv2d_transpose (layers.py:67-80) */
#define IAF_RESAMPLE_DOWN_EVEN 0
#define IAF_RESAMPLE_DOWN_ODD 1
#define IAF_RESAMPLE_UP_NEAREST 2
#define IAF_RESAMPLE_UP_ZERO_ODD 3
#define IAF_RESAMPLE_UP_ZERO_EVEN 4   /* dst[2i,2j] = src[i,j], 0 elsewhere: adjoint of DOWN_EVEN (backward of resize_nearest_neighbor(x, 0.5)) */
#define IAF_RESAMPLE_DOWN_SUM4 5      /* dst[i,j] = sum of the 2x2 block: adjoint of UP_NEAREST (backward of resize_nearest_neighbor(x, 2)) */
int iaf_resample2(const float* src, float* dst, int B, int C, int H, int W, int mode, void* stream);
/* deconv2d(name, x, num_filters, stride=(2,2)) (tf_utils/layers.py:83-112): V is [3,3,n_out,n_in]; the reference's weight
 * norm runs over (kh,kw,n_OUT) per INPUT channel (layers.py:104) and is kept.  Prepares an iaf_conv3x3 (n_in, n_out) so
 * that iaf_conv3x3_forward on the UP_ZERO_ODD-resampled input, at the output resolution, equals the reference's
 * conv2d_transpose(SAME, stride 2) + b.  iaf_conv3x3_backward of a conv prepared this way differentiates THIS weight
 * norm (dV in V's [3,3,n_out,n_in] layout) and, when training is on, the prepare also writes the data gradient's pack. */
int iaf_conv3x3_prepare_deconv(iaf_conv3x3_t* c, const float* V, const float* g, const float* b, void* stream);
/* The two strided convs at their minimal work (nine taps per pixel of the SMALLER grid; bf16x3 kernels only).
 * iaf_conv3x3_forward_stride2: y = conv2d(name, [elu](x), n_out, stride=[2,2]) (tf_train.py:33,36; tf_utils/layers.py:31-64), split into
 *   n_outs tensors like iaf_conv3x3_forward.  x [B,n_in,2H,2W]; H, W = the OUTPUT size; outs[k] [B,out_channels[k],H,W].
 * iaf_conv3x3_forward_deconv: out = [residual +] 0.1 * deconv2d(name, [elu](concat(x[:, :c_split], x2)), n_out) (tf_train.py:87-94;
 *   tf_utils/layers.py:83-112) of a conv prepared by iaf_conv3x3_prepare_deconv.  x, x2 [B,.,H,W] (H, W = the INPUT size), out
 *   [B,n_out,2H,2W], residual [B,n_out,H,W] (added to all four pixels of its 2x2 block: resize_nearest_neighbor(input, 2),
 *   tf_train.py:90) or NULL (then out = the deconv itself, unscaled).
 * Both return IAF_ERR_UNSUPPORTED for shapes their tiles do not cover (fp32 precision pinned, c_in not a multiple of 32, an LDS tile
 * over 160 KiB): the caller then uses iaf_conv3x3_forward + iaf_resample2 (DOWN_ODD / UP_ZERO_ODD), which computes the same numbers. */
int iaf_conv3x3_forward_stride2(iaf_conv3x3_t* c, const float* x, int elu_input, float* const* outs, const int* out_channels,
                                int n_outs, int B, int H, int W, void* stream);
int iaf_conv3x3_forward_deconv(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                               const float* residual, float* out, int B, int H, int W, void* stream);
/* eps_out with (qz_mean+rz_mean) + exp(qz_logsd+rz_logsd)*eps_out == z: mode "init" of IAFLayer.down runs the posterior
 * block on a PRIOR sample (tf_train.py:60-61, 67-85) */
int iaf_noise_from_sample(const float* z, const float* qz_mean, const float* qz_logsd, const float* rz_mean,
                          const float* rz_logsd, float* eps_out, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * The two ends of the model around the IAFLayer stack: CVAE1._forward (tf_train.py:150-218)
 * ------------------------------------------------------------------------------------------ */
/* out[(b k + s)][i] = clip((x[b][i] + 0.5) / 256, 0, 1) - 0.5, s < k: tf.to_float + clip (tf_train.py:153-154) and repeat(x, k)
 * (:159; tf_utils/distributions.py:40-52).  x uint8 [B][n_per_image], out float [B k][n_per_image]. */
int iaf_image_to_float(const unsigned char* x, float* out, int B, size_t n_per_image, int k, void* stream);
/* weight norm of a kh x kw filter into w (same layout as V):
 *   deconv = 0  conv2d (tf_utils/layers.py:56-60):    V [kh,kw,n_in,n_out],  w = exp(g[o]) V / ||V||_(kh,kw,n_in)
 *   deconv = 1  deconv2d (tf_utils/layers.py:104-106): V [kh,kw,n_out,n_in], w = exp(g[o]) V / ||V||_(kh,kw,n_out)  (per INPUT channel) */
int iaf_convk_weightnorm(const float* V, const float* g, float* w, int kh, int kw, int n_in, int n_out, int deconv, void* stream);
/* y = tf.nn.conv2d([elu](x), w, [1,1,s,s], SAME, NCHW) + b (tf_utils/layers.py:63-64) for any filter size / stride -- a direct
 * conv for the model's first layer conv2d("x_enc", x, h_size, [5,5], [2,2]) (tf_train.py:183; 3 input channels: not MFMA work).
 * x [B,n_in,H,W], y [B,n_out,ceil(H/s),ceil(W/s)]. */
int iaf_convk_forward(const float* x, const float* w, const float* b, float* y, int B, int n_in, int H, int W, int n_out, int kh, int kw,
                      int stride, int elu_input, void* stream);
/* y = clip(conv2d_transpose([elu](x), w, SAME, stride s) + b, clip_lo, clip_hi) (tf_utils/layers.py:67-80,108-111; the model's last
 * layer deconv2d("x_dec", elu(h), 3, [5,5]) + clip_by_value, tf_train.py:206-208).  x [B,n_in,H,W], y [B,n_out,H s,W s]; no clipping
 * when clip_lo >= clip_hi. */
int iaf_deconvk_forward(const float* x, const float* w, const float* b, float* y, int B, int n_in, int H, int W, int n_out, int kh,
                        int kw, int stride, int elu_input, float clip_lo, float clip_hi, void* stream);
/* out[b,c,:] = v[c]: tf.tile(tf.reshape(h_top, [1,-1,1,1]), [data_size,1,S,S]) (tf_train.py:190-192) */
int iaf_tile_channels(const float* v, float* out, int B, int C, int HW, void* stream);
/* out[0] = sum_i (a[i] + sb b[i]) (b may be NULL), fixed summation order: obj = reduce_sum(kl_obj - log_pxz) (tf_train.py:211),
 * loss = reduce_sum(compute_lowerbound(...)) (:218) */
int iaf_sum_axpy(const float* a, const float* b, float sb, float* out, int n, void* stream);
/* Backward of the two ends -- what TF autodiff derives for tf_train.py:183, 189-192, 206-211 under opt.compute_gradients(obj) (:128):
 * iaf_discretized_logistic_backward: d_mean = up * d log_pxz / d mean with clip_by_value's gradient folded in (zero where mean sits on
 *   clip_lo / clip_hi; no clipping when clip_lo >= clip_hi), d_logscale_rows[b] = up * sum over row b of d log p / d logscale
 *   (tf_utils/distributions.py:28-32; up = -1 for obj = sum(kl_obj - log_pxz)).
 * iaf_convk_wgrad: dW[a][c][ci][o] = sum_{b,oy,ox} X[b,ci,oy s + a - pad_t, ox s + c - pad_l] DY[b,o,oy,ox] (SAME padding), X [B,n_small,H,W]
 *   on the large grid, DY [B,n_big,ceil(H/s),ceil(W/s)], optional ELU on either operand: x_enc's filter gradient (X = image, DY = d h)
 *   and x_dec's (X = d x_out, DY = elu(h): V [kh,kw,3,h] has the same layout).
 * iaf_convk_weightnorm_backward: dW -> dV, dg through iaf_convk_weightnorm's reparametrisation (deconv: scratch of n_in * n_out floats).
 * iaf_channel_sum: out[c] = sum_{b,p} x[b,c,p] (bias gradients; d h_top = the adjoint of the tile).
 * iaf_mul_elu_grad: out = g * elu'(h). */
int iaf_discretized_logistic_backward(const float* mean, const float* logscale, const float* sample, float up, float clip_lo,
                                      float clip_hi, float* d_mean, float* d_logscale_rows, int B, size_t n_per_row, float binsize,
                                      void* stream);
int iaf_convk_wgrad(const float* x, const float* dy, float* dW, int B, int n_small, int H, int W, int n_big, int kh, int kw, int stride,
                    int elu_x, int elu_dy, void* stream);
int iaf_convk_weightnorm_backward(const float* V, const float* g, const float* dW, float* dV, float* dg, float* scratch, int kh, int kw,
                                  int n_in, int n_out, int deconv, void* stream);
int iaf_channel_sum(const float* x, float* out, int B, int C, int HW, void* stream);
int iaf_mul_elu_grad(const float* g, const float* h, float* out, size_t n, void* stream);
/* elementwise pieces of the model's data-dependent init pass (CVAE1 in mode "init", tf_train.py:175 arg_scope(init=True)):
 * out = sa a + sb b (residuals `input + 0.1 h`, tf_train.py:44,94; context = up_context + down_context, :58);
 * out = (z - scale m) / exp(scale s) (the IAF update from ar_multiconv2d's raw outputs, :70-71);  out = clip(x, lo, hi) (:208) */
int iaf_axpby(const float* a, float sa, const float* b, float sb, float* out, size_t n, void* stream);
int iaf_affine_transform(const float* z, const float* m, const float* s, float scale, float* out, size_t n, void* stream);
int iaf_clip(const float* x, float lo, float hi, float* out, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IAF_HIP_H */
