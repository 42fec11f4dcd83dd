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
    unsigned long long* dbg;   // dev tool: per-workgroup cycle stamps [grid][8]
#ifdef IAF_EXP_FUSED_KL
    // Experiment (written, not yet run on a GPU): the posterior block's KL reductions inside this launch.
    // Every workgroup leaves the per-channel sums of its rows' KL elements in kl_part [B * nrb][n_z], takes a ticket from
    // kl_cnt (zeroed by a memset node ahead of every launch); the workgroup that draws the last ticket sums the partials in a
    // fixed order and applies the free-bits rule (tf_train.py:77-85) -- what iaf_kl_rowsum_kernel + iaf_kl_finish_kernel do
    // in two more launches.  kl_elem may then be NULL (no [B, n_z, H, W] KL tensor is written).
    float* kl_part;
    unsigned* kl_cnt;
    float* kl_obj;             // [B]
    float* kl_cost;            // [B]
    float* kl_gate;            // [n_z] or NULL (training: where the free-bits max() passes the gradient)
    float kl_min;
#endif
};

typedef void (*step_fn_t)(StepP);
// kernel + dynamic LDS bytes for (n_h / 16, n_z / 16, depth_ar, image width, output rows per workgroup), or NULL
// var: 0 TF statement, 1 Theano, 2 Theano with flipmask
extern "C" step_fn_t iaf_pick_step_fused_a(int nht, int nzt, int depth, int W, int R, int var, size_t* lds);   // depth_ar <= 2 geometries
extern "C" step_fn_t iaf_pick_step_fused_b(int nht, int nzt, int depth, int W, int R, int var, size_t* lds);   // depth_ar = 4 geometries
static inline step_fn_t iaf_pick_step_fused(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    step_fn_t f = iaf_pick_step_fused_a(nht, nzt, depth, W, R, var, lds);
    return f ? f : iaf_pick_step_fused_b(nht, nzt, depth, W, R, var, lds);
}
