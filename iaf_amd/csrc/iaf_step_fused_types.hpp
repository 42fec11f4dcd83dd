// iaf_step_fused_types.hpp -- launch descriptor of the one-launch IAF step (iaf_step_fused.hpp), shared with the host code.
#pragma once

struct StepP {
    const float* z;            // [B][n_z][H][W]; NULL: the posterior sample (qm + rm) + exp(ql + rl) * eps
    const float* ctx;          // [B][n_h][H][W] context added after the first conv (layers.py:163-164)
    const float* ctx2;         // optional second context (up_context + down_context, tf_train.py:58)
    const void* wp3[5];        // bf16x3 packs of the D <= 4 hidden convs and the output pair
    const float* bias[5];
    const float* zin;          // MODE_IAF / MODE_INVERSE: the z of the affine transform
    float* hsave[4];           // training: hidden activations of the OWNED rows, pixel-major [P][n_h] (what the backward reads), or NULL
    float* out0;
    float* out1;
    float* kl_elem;
    const float* qm; const float* ql; const float* rm; const float* rl; const float* pm; const float* pl; const float* eps;
    int B, H, HW, mode, nrb;   // nrb = ceil(H / R) row blocks per image
    // The Theano statement of the operator (graphy/nodes/ar.py + conv.py): its taps look left / above -- the TF geometry on
    // the image rotated by 180 degrees, so that variant of the kernel sends every global access through (H-1-row, W-1-col) and
    // nothing else changes; border[l] = [4][packed c_out of layer l] weights of the border-indicator channel (taps 1..4), added
    // where a tap leaves the image (Theano variants only)
    const float* border[5];
    unsigned long long* dbg;   // dev tool: per-workgroup cycle stamps [grid][16]
    // Posterior mode: per-channel sums of the workgroup's KL elements, [B * nrb][n_z] -- the first step of the block's
    // reductions (tf_train.py:77: sum over H, W) leaves the launch as 1/(R*W) of the bytes of the KL tensor, and kl_elem may
    // then be NULL (no [B, n_z, H, W] KL tensor is written or re-read).  The row blocks are summed in row order, the batch
    // mean / free-bits max / channel sum applied (tf_train.py:79-85) by iaf_kl_finish_kernel: ONE small launch behind this
    // one instead of the two (row sums over the KL tensor + finish) of round 2.  NULL: no partial sums are written.
    float* kl_part;
    // The block's free-bits reductions inside this launch (kernels with helper waves, posterior mode; NULL fin_ctl: the caller runs
    // iaf_kl_finish_kernel behind the launch): the LAST workgroup to arrive -- counted in fin_ctl, 64-bit words, one 128-byte line
    // each: [32 g] arrivals of the workgroups with blockIdx % 8 = g, [256] groups complete; zero between launches -- sums kl_part
    // over the row blocks, takes the batch mean / max(., kl_min) / channel sum (tf_train.py:79-85) and writes fin_obj, fin_cost [B],
    // in the summation order of iaf_kl_finish_kernel (bit-identical results).
    float* fin_obj;
    float* fin_cost;
    float* fin_gate;               // optional [n_z]: 1 where the free-bits max() passes the gradient (training forward), else 0
    unsigned long long* fin_ctl;
    float fin_kl_min;
    // XCH kernels (halo rows exchanged between the row blocks of an image instead of recomputed; iaf_step_fused.hpp "XCH"):
    char* xh;                      // rows [layer][B * nrb][xrow bytes]; every dword = IAF_XSENT between launches (the data is the flag)
                                   // (PAIR kernels: [B * nrb][half][prow bytes], the halves of the last hidden region the partners swap)
    unsigned long long* xctl;      // [32 y] head of work list y (tickets taken), [32 y + 16] its arrivals, [IAF_XCTL_DONE] lists complete,
                                   // [IAF_XCTL_STICKY] sticky error -- a 128-byte line each; zero between launches
    unsigned* xerr;                // host-visible error word (mapped pinned memory), or NULL
    unsigned* rng_err;             // F16 kernels: host-visible word (mapped pinned memory) raised when an operand beyond fp16's largest finite
                                   // number went into the planes (the launch's outputs then carry inf / NaN); or NULL
    const unsigned* skip;          // MODE_INVERSE: a device word; non-zero = the chain of sweeps this launch belongs to has converged: return at once
    unsigned xknob;                // test knobs: 1 lists ignore the placement, 2 tickets out of dispatch order,
                                   // 8 fault injection (image 0's bottom block never hands over its first row; short waits)
};

// StepP::xctl, 64-bit words, every counter in a 128-byte line of its own: [32 y] head of work list y (tickets taken), [32 y + 16] its
// arrivals, y < IAF_XCTL_LISTS; [IAF_XCTL_DONE] lists complete; [IAF_XCTL_STICKY] sticky error; StepP::fin_ctl = xctl + IAF_XCTL_FIN
#define IAF_XCTL_LISTS 8          /* one per XCD; 32 (four per XCD, picked by CU id) was tried in round 5: where an XCD's CUs are spread unevenly
                                     over its four lists the surplus workgroups steal, one memory-side round trip per dry list -- up to 30 k
                                     cycles of prologue on some boxes (profiles/r05/experiments/ticket_lists.txt) */
#define IAF_XCTL_DONE 1024
#define IAF_XCTL_STICKY 1040
#define IAF_XCTL_FIN 1536
#define IAF_XCTL_WORDS 2048
// "Not there yet" in the hand-over buffers (StepP::xh): every dword holds this pattern between launches.  A pair of SIGNALLING bf16 NaNs
// (exponent all ones, quiet bit clear): what travels through those buffers are hidden activations -- results of arithmetic (bias + ELU,
// then the bf16 split), and arithmetic only ever returns QUIET NaNs (IEEE mode, which HIP kernels run in), whatever payload a caller's NaN
// carried in.  No data dword can equal it, so the consumers' "has every unit arrived" test needs no fix-up on the producers' side (round 5
// first used 0xffffffff -- which a NaN of all ones in the inputs does reach -- plus a fix-up in the exporting helper waves: 0.7 us per
// 16x16 launch, VALU work in waves that share their SIMD's issue with a compute wave).
#define IAF_XSENT 0xffbfffbfu
// ... of the F16 kernels, whose rows hold fp16 planes: a pair of SIGNALLING fp16 NaNs (exponent all ones, quiet bit 9 clear) -- 0xffbf read as
// fp16 is a QUIET NaN, which a conversion of a caller's NaN can return
#define IAF_XSENT_F16 0xfdfffdffu

typedef void (*step_fn_t)(StepP);
// kernel + dynamic LDS bytes for (n_h / 16, n_z / 16, depth_ar, image width, output rows per workgroup), or NULL
// var: 0 TF statement, 1 Theano, 2 Theano with flipmask
extern "C" step_fn_t iaf_pick_step_fused_a(int nht, int nzt, int depth, int W, int R, int var, size_t* lds);   // depth_ar <= 2 geometries
extern "C" step_fn_t iaf_pick_step_fused_b(int nht, int nzt, int depth, int W, int R, int var, size_t* lds);   // depth_ar = 4 geometries
// ... in the halo-exchange form (StepP::xh; *xrow = bytes of one exported row); var as above
extern "C" step_fn_t iaf_pick_step_fused_xch(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow);
extern "C" step_fn_t iaf_pick_step_fused_xch_b(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow);   // depth_ar = 4
extern "C" step_fn_t iaf_pick_step_fused_h(int nht, int nzt, int depth, int W, int R, int var, size_t* lds);   // recomputing form with helper waves
extern "C" step_fn_t iaf_pick_step_fused_c(int nht, int nzt, int depth, int W, int R, int var, size_t* lds);   // depth_ar = 3 geometries
extern "C" step_fn_t iaf_pick_step_fused_xch_c(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow);
// the pair form (two workgroups per (image, row block), *prow = bytes one of them hands the other); R = rows per PAIR
extern "C" step_fn_t iaf_pick_step_fused_pair(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* prow);
extern "C" step_fn_t iaf_pick_step_fused_d(int nht, int nzt, int depth, int W, int R, int var, size_t* lds);   // n_z = 32, depth_ar = 2, n_h = 64 / 128
extern "C" step_fn_t iaf_pick_step_fused_xch_d(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow);
// the two-plane fp16 kernels ("f16x2"): form = 0 recomputing with helper waves, 1 halo exchange; *xrow as above (0 for form 0)
extern "C" step_fn_t iaf_pick_step_fused_f16_a(int nht, int nzt, int depth, int W, int R, int var, int form, size_t* lds, size_t* xrow);
extern "C" step_fn_t iaf_pick_step_fused_f16_b(int nht, int nzt, int depth, int W, int R, int var, int form, size_t* lds, size_t* xrow);
static inline step_fn_t iaf_pick_step_fused_f16(int nht, int nzt, int depth, int W, int R, int var, int form, size_t* lds, size_t* xrow) {
    step_fn_t f = iaf_pick_step_fused_f16_a(nht, nzt, depth, W, R, var, form, lds, xrow);
    return f ? f : iaf_pick_step_fused_f16_b(nht, nzt, depth, W, R, var, form, lds, xrow);
}
static inline step_fn_t iaf_pick_step_fused(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    step_fn_t f = iaf_pick_step_fused_a(nht, nzt, depth, W, R, var, lds);
    if (!f) f = iaf_pick_step_fused_b(nht, nzt, depth, W, R, var, lds);
    if (!f) f = iaf_pick_step_fused_c(nht, nzt, depth, W, R, var, lds);
    return f ? f : iaf_pick_step_fused_d(nht, nzt, depth, W, R, var, lds);
}
