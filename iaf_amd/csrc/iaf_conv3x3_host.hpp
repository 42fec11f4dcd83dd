// iaf_conv3x3_host.hpp -- host side + C ABI of the plain / single masked 3x3 convs (iaf_conv3x3_*), data-dependent init and the likelihood.
// Part of the single translation unit iaf_engine.hip (included there, in order; not a standalone header).
#pragma once

// ---------------------------------------------------------------------------------------------
// plain weight-normed 3x3 convs around the IAF step: up_conv1 / up_conv3 / down_conv1 / down_conv2
// (tf_train.py:36-44, 52-54, 87-94; operator tf_utils/layers.py:31-64 with mask=None, stride 1, pad SAME).
// Same implicit-GEMM kernel as the masked stack with all 9 taps live (template NTP = 9) and the EPI_PLAIN epilogue:
// ELU / channel concat fused into the input staging, channel split / residual fused into the store.
// ---------------------------------------------------------------------------------------------
struct iaf_conv3x3 {
    int n_in, n_out;
    int mask_mode;     // 0 plain conv2d, 1 ar_conv2d(zerodiagonal=False), 2 ar_conv2d(zerodiagonal=True)
    int variant = IAF_VARIANT_TF;   // masked convs only: the Theano statements of ar.conv2d (graphy/nodes/ar.py:200-375)
    bool generic, prepared;
    GemmLayer L;
    PrepLayer* h_desc = nullptr;   // pinned staging of the prep descriptor
    PrepLayer* d_desc = nullptr;
    bool training = false;
    bool deconv = false;           // weights prepared by iaf_conv3x3_prepare_deconv (deconv2d's norm + rotated filter)
    // forward on the bf16 matrix cores (bf16x3 split products, iaf_conv_bf3.hpp with 9 taps): plain convs with c_in % 32 == 0
    // IAF_PRECISION_F16X2 (round 6): as BF16X3, the stride-1 forward launches on TWO fp16 planes (iaf_conv_bf3.hpp F16) from the pack L.wp2;
    // an operand beyond 65504 raises rng_err (mapped host memory): the next forward returns IAF_ERR_RANGE once and the conv runs bf16x3
    // from then on (f16_off) until iaf_conv3x3_set_precision(F16X2) re-arms
    int precision = IAF_PRECISION_BF16X3;
    unsigned* rng_err_host = nullptr;
    unsigned* rng_err_dev = nullptr;
    bool f16_off = false;
    int packs = IAF_PACK_F32 | IAF_PACK_BF16X3 | IAF_PACK_F16X2;   // iaf_conv3x3_set_packs: which packs the prep launches keep up to date
    int bf3_choice = 1;            // 1 size rule, 2 the shape in L.b_* (pinned by iaf_conv3x3_autotune), 3 fp32 kernel (measured faster)
    GemmLayer T;                   // transposed problem dX = W^T dY (valid when training)
    // deferred weight-norm backward (iaf_conv3x3_wn_bwd_batch_run): the reduced dW / db partials live here, not in the
    // (shared) workspace
    bool defer_wn = false;
    float* own_dW = nullptr; float* own_dbp = nullptr;
    int pend_nslab = 0; bool pending = false;
};

extern "C" int iaf_conv3x3_destroy(iaf_conv3x3_t* c) {
    if (!c) return IAF_ERR_NULL;
    if (c->L.wp) (void)hipFree(c->L.wp);
    if (c->L.wpt3) (void)hipFree(c->L.wpt3);
    if (c->L.bias) (void)hipFree(c->L.bias);
    if (c->L.wpt) (void)hipFree(c->L.wpt);
    if (c->L.wp3) (void)hipFree(c->L.wp3);
    if (c->L.wp2) (void)hipFree(c->L.wp2);
    if (c->L.wpt2) (void)hipFree(c->L.wpt2);
    if (c->rng_err_host) (void)hipHostFree(c->rng_err_host);
    if (c->L.border) (void)hipFree(c->L.border);
    if (c->own_dW) (void)hipFree(c->own_dW);
    if (c->own_dbp) (void)hipFree(c->own_dbp);
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    if (c->d_desc) (void)hipFree(c->d_desc);
    delete c;
    return IAF_OK;
}

static inline bool conv_split(const iaf_conv3x3* c) { return c->precision != IAF_PRECISION_F32; }
// the forward launches of this conv run the two-plane fp16 kernels now
static inline bool conv_f16_active(const iaf_conv3x3* c) {
    return c->precision == IAF_PRECISION_F16X2 && !c->f16_off && c->L.wp2 && !c->generic && !c->mask_mode;
}
// which packs a prep launch writes for a plain conv (iaf_conv3x3_set_packs; training keeps all of them)
static inline void conv_prep_packs(const iaf_conv3x3* c, PrepLayer& P) {
    const GemmLayer& L = c->L;
    P.wp = (c->packs & IAF_PACK_F32) ? L.wp : nullptr;
    P.wp3 = (c->packs & IAF_PACK_BF16X3) ? L.wp3 : nullptr;
    P.wp2 = ((c->packs & IAF_PACK_F16X2) && conv_f16_active(c)) ? L.wp2 : nullptr;
    P.rng_err = P.wp2 ? c->rng_err_dev : nullptr;
    P.wpt = c->training ? L.wpt : nullptr;
}
static int conv3x3_create(iaf_conv3x3_t** out, int n_in, int n_out, int mask_mode);
extern "C" int iaf_conv3x3_create(iaf_conv3x3_t** out, int n_in, int n_out) { return conv3x3_create(out, n_in, n_out, 0); }
extern "C" int iaf_conv3x3_create_masked(iaf_conv3x3_t** out, int n_in, int n_out, int zerodiagonal) {
    if (n_in > 0 && n_out > 0 && !(n_in % n_out == 0 || n_out % n_in == 0)) return IAF_ERR_NOT_MULTIPLE;   // layers.py:116
    return conv3x3_create(out, n_in, n_out, zerodiagonal ? 2 : 1);
}

// N.ar.conv2d(name, n_in, n_out, (3,3), zerodiagonal, flipmask, w=w) of the Theano path (graphy/nodes/ar.py:200-375; e.g.
// posteriors 'up_iaf1' / 'down_iaf2', models.py:53-55,73-79): weights w OIHW [n_out][n_in+1][3][3], s, b.  Forward only.
extern "C" int iaf_conv3x3_create_masked_theano(iaf_conv3x3_t** out, int n_in, int n_out, int zerodiagonal, int flipmask) {
    if (n_in > 0 && n_out > 0 && !(n_in % n_out == 0 || n_out % n_in == 0)) return IAF_ERR_NOT_MULTIPLE;   // ar.py:250,257
    int rc = conv3x3_create(out, n_in, n_out, zerodiagonal ? 2 : 1);
    if (rc) return rc;
    iaf_conv3x3* c = *out;
    if (c->generic) { iaf_conv3x3_destroy(c); *out = nullptr; return IAF_ERR_UNSUPPORTED; }
    c->variant = flipmask ? IAF_VARIANT_THEANO_FLIPMASK : IAF_VARIANT_THEANO;
    if ((rc = (int)hipMalloc(&c->L.border, (size_t)4 * c->L.ncot * 16 * sizeof(float))) != 0) { iaf_conv3x3_destroy(c); *out = nullptr; return rc; }
    return IAF_OK;
}

static int conv3x3_create(iaf_conv3x3_t** out, int n_in, int n_out, int mask_mode) {
    if (!out) return IAF_ERR_NULL;
    *out = nullptr;
    if (n_in <= 0 || n_out <= 0) return IAF_ERR_SHAPE;
    iaf_conv3x3* c = new (std::nothrow) iaf_conv3x3();
    if (!c) return (int)hipErrorOutOfMemory;
    c->n_in = n_in; c->n_out = n_out; c->prepared = false; c->mask_mode = mask_mode;
    c->generic = (n_in % 16 != 0 || n_out % 16 != 0 || n_in > 16 * PREP_MAXI);
    GemmLayer& L = c->L;
    L.cin = n_in; L.cout = n_out; L.npair = 1; L.zerodiag = (mask_mode == 2) ? 1 : 0; L.full3x3 = (mask_mode == 0);
    L.nchunk = (n_in + 15) / 16; L.ncot = (n_out + 15) / 16;
    default_tuning(L, false);
    L.live_macs_per_px = L.dense_macs_per_px = 9.0 * n_in * n_out;
    if (mask_mode) count_macs(L, n_in, n_out, L.zerodiag, 1);
    const size_t wfloats = c->generic ? (size_t)MAXTAPS * n_in * n_out : (size_t)L.nchunk * MAXTAPS * L.ncot * 256;
    int rc;
    if ((rc = (int)hipMalloc(&L.wp, wfloats * sizeof(float))) != 0 ||
        (rc = (int)hipMalloc(&L.bias, ((size_t)L.ncot * 16 + (size_t)n_in) * sizeof(float))) != 0 ||   // + deconv norms
        (rc = (int)hipHostMalloc((void**)&c->h_desc, sizeof(PrepLayer))) != 0 ||
        (rc = (int)hipMalloc((void**)&c->d_desc, sizeof(PrepLayer))) != 0) {
        iaf_conv3x3_destroy(c);
        return rc;
    }
    if (!c->generic && mask_mode == 0 && n_in % 32 == 0 &&
        (rc = (int)hipMalloc(&L.wp3, (size_t)(n_in / 32) * MAXTAPS * L.ncot * 3 * 64 * 16)) != 0) {      // bf16x3 pack
        iaf_conv3x3_destroy(c);
        return rc;
    }
    // default arithmetic of a plain conv with a split pack: the two-plane fp16 forward (IAF_DEFAULT_PRECISION=bf16x3: round 5's)
    if (L.wp3) {
        static const bool f16_default = !(getenv("IAF_DEFAULT_PRECISION") && !strcmp(getenv("IAF_DEFAULT_PRECISION"), "bf16x3"));
        if (f16_default) {
            rc = iaf_conv3x3_set_precision(c, IAF_PRECISION_F16X2);
            if (rc != IAF_OK && rc != IAF_ERR_UNSUPPORTED) { iaf_conv3x3_destroy(c); return rc; }
        }
    }
    *out = c;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_prepare(iaf_conv3x3_t* c, const float* V, const float* g, const float* b, void* stream) {
    if (!c || !V || !g || !b) return IAF_ERR_NULL;
    const GemmLayer& L = c->L;
    c->deconv = false;
    if (c->generic) {
        GenPrepArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.nlayers = 1;
        GenPrepLayer& P = ga.L[0];
        P.V[0] = V; P.g[0] = g; P.b[0] = b; P.w = L.wp; P.bias = L.bias;
        P.cin = L.cin; P.cout_each = L.cout; P.npair = 1; P.zerodiag = L.zerodiag; P.ch_begin = 0; P.ntaps = MAXTAPS;
        P.mask9 = c->mask_mode ? 1 : 0;
        hipLaunchKernelGGL(iaf_generic_prep_kernel, dim3(L.cout), dim3(256), 0, (hipStream_t)stream, ga);
    } else if (c->mask_mode) {    // the masked prep of the stack, one layer
        PrepArgs a;
        memset(&a, 0, sizeof(a));
        a.nlayers = 1;
        PrepLayer& P = a.L[0];
        P.V[0] = V; P.g[0] = g; P.b[0] = b; P.wp = L.wp; P.bias = L.bias; P.variant = c->variant; P.border = L.border;
        P.cin = L.cin; P.cout_each = L.cout; P.ncot = L.ncot; P.nchunk = L.nchunk; P.zerodiag = L.zerodiag; P.npair = 1;
        hipLaunchKernelGGL(iaf_prep_kernel, dim3(L.ncot), dim3(256), 0, (hipStream_t)stream, a, 0u);
    } else {
        PrepLayer& P = *c->h_desc;
        memset(&P, 0, sizeof(P));
        P.V[0] = V; P.g[0] = g; P.b[0] = b; P.bias = L.bias; P.variant = PREP_PLAIN9;
        conv_prep_packs(c, P);
        P.cin = L.cin; P.cout_each = L.cout; P.ncot = L.ncot; P.nchunk = L.nchunk; P.npair = 1; P.tile_begin = 0;
        HIP_TRY(hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(PrepLayer), hipMemcpyHostToDevice, (hipStream_t)stream));
        hipLaunchKernelGGL(iaf_prep_plain_kernel, dim3(L.ncot), dim3(256), 0, (hipStream_t)stream, c->d_desc, (const int*)nullptr);
        if (c->training) {
            PackT3Batch tb((hipStream_t)stream);
            int rc = tb.add(L, MAXTAPS);
            if (!rc) rc = tb.flush();
            if (rc) return rc;
        }
    }
    HIP_TRY(hipGetLastError());
    c->prepared = true;
    return IAF_OK;
}

// deconv2d("down_deconv2") of a downsampling IAFLayer (tf_train.py:91; tf_utils/layers.py:67-112): V is [3,3,n_out,n_in].
// The object is an ordinary iaf_conv3x3 (n_in, n_out) afterwards: run it with iaf_conv3x3_forward on the zero-inserted
// input (iaf_resample2 mode IAF_RESAMPLE_UP_ZERO_ODD) at the OUTPUT resolution.  Forward only.
extern "C" int iaf_conv3x3_prepare_deconv(iaf_conv3x3_t* c, const float* V, const float* g, const float* b, void* stream) {
    if (!c || !V || !g || !b) return IAF_ERR_NULL;
    if (c->mask_mode) return IAF_ERR_UNSUPPORTED;
    if (c->packs != (IAF_PACK_F32 | IAF_PACK_BF16X3 | IAF_PACK_F16X2)) return IAF_ERR_UNSUPPORTED;    // (the deconv packs derive from the fp32 one)
    GemmLayer& L = c->L;
    hipStream_t st = (hipStream_t)stream;
    float* inv_norm = L.bias + (size_t)L.ncot * 16;    // n_in floats behind the packed bias (conv3x3_create)
    hipLaunchKernelGGL(iaf_deconv_norm_kernel, dim3(L.cin), dim3(256), 0, st, V, inv_norm, L.cin, L.cout);
    const size_t total = (size_t)9 * L.cin * L.cout;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(iaf_deconv_pack_kernel, dim3(blocks), dim3(256), 0, st, V, g, b, (const float*)inv_norm, L.wp, L.bias,
                       L.cin, L.cout, L.ncot, c->generic ? 1 : 0, (c->training && !c->generic) ? L.wpt : nullptr);
    HIP_TRY(hipGetLastError());
    if (L.wp3 && !c->generic && L.nchunk % 2 == 0) {   // the same pack as three bf16 planes (iaf_conv3x3_forward_deconv): wp has the layout
        PackT3Args a;                                  // iaf_pack_t3_kernel reads ([K chunk][tap][N tile][lane][4])
        memset(&a, 0, sizeof(a));
        a.L[0].src = L.wp; a.L[0].dst = L.wp3; a.L[0].ntp = MAXTAPS; a.L[0].nct = L.ncot; a.L[0].begin = 0;
        a.n = 1; a.total = (L.nchunk / 2) * MAXTAPS * L.ncot * 64;
        hipLaunchKernelGGL(iaf_pack_t3_kernel, dim3((a.total + 255) / 256), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
    }
    if (c->training && !c->generic) {           // ... and its bf16x3 form for the data gradient
        PackT3Batch tb(st);
        int rc = tb.add(L, MAXTAPS);
        if (!rc) rc = tb.flush();
        if (rc) return rc;
    }
    c->prepared = true;
    c->deconv = true;
    return IAF_OK;
}

// 2x resampling of an NCHW tensor (modes: IAF_RESAMPLE_* in include/iaf_hip.h); H, W = size of the SMALLER tensor
extern "C" int iaf_resample2(const float* src, float* dst, int B, int C, int H, int W, int mode, void* stream) {
    if (!src || !dst) return IAF_ERR_NULL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 5) return IAF_ERR_SHAPE;
    const bool down = (mode == IAF_RESAMPLE_DOWN_EVEN || mode == IAF_RESAMPLE_DOWN_ODD || mode == IAF_RESAMPLE_DOWN_SUM4);
    const size_t n_dst = (size_t)B * C * H * W * (down ? 1 : 4);
    hipLaunchKernelGGL(iaf_resample2_kernel, ew_grid(n_dst), dim3(256), 0, (hipStream_t)stream, src, dst, n_dst, H, W, mode);
    return (int)hipGetLastError();
}

// eps' with (qm+rm) + exp(ql+rl) * eps' == z: the posterior noise that reproduces a given sample z (mode "init" of
// IAFLayer.down runs the posterior block on a PRIOR sample, tf_train.py:60-61,67-85)
__global__ __launch_bounds__(256) void iaf_noise_from_sample_kernel(const float* __restrict__ z, const float* __restrict__ qm,
                                                                   const float* __restrict__ ql, const float* __restrict__ rm,
                                                                   const float* __restrict__ rl, float* __restrict__ eps, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        eps[i] = (z[i] - (qm[i] + rm[i])) * __expf(-(ql[i] + rl[i]));
}
extern "C" int iaf_noise_from_sample(const float* z, const float* qz_mean, const float* qz_logsd, const float* rz_mean,
                                     const float* rz_logsd, float* eps_out, size_t n, void* stream) {
    if (!z || !qz_mean || !qz_logsd || !rz_mean || !rz_logsd || !eps_out) return IAF_ERR_NULL;
    if (n == 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_noise_from_sample_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, z, qz_mean, qz_logsd, rz_mean,
                       rz_logsd, eps_out, n);
    return (int)hipGetLastError();
}

// kl = logq0 + logdet - logp: models.py:175,298,328 when the three terms come from separate launches (posterior
// 'up_iaf2_nl': the IAF step runs in the up pass, the prior is only known in the down pass)
__global__ __launch_bounds__(256) void iaf_kl_combine_kernel(const float* __restrict__ logq0, const float* __restrict__ logdet,
                                                            const float* __restrict__ logp, float* __restrict__ kl, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) kl[i] = (logq0[i] + logdet[i]) - logp[i];
}
extern "C" int iaf_kl_combine(const float* logq0, const float* logdet, const float* logp, float* kl, size_t n, void* stream) {
    if (!logq0 || !logdet || !logp || !kl) return IAF_ERR_NULL;
    if (n == 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_kl_combine_kernel, ew_grid(n), dim3(256), 0, (hipStream_t)stream, logq0, logdet, logp, kl, n);
    return (int)hipGetLastError();
}

// out[j] = sum_i mat[i][j]: the per-row KL costs of all layers of a model -> sum_kl_costs (tf_train.py:198-200:
// `kl_cost += cur_cost` over the layer loop), the second argument of compute_lowerbound
__global__ __launch_bounds__(256) void iaf_colsum_kernel(const float* __restrict__ mat, float* __restrict__ out, int m, int n) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int i = 0; i < m; ++i) s += mat[(size_t)i * n + j];
    out[j] = s;
}
extern "C" int iaf_colsum(const float* mat, float* out, int m, int n, void* stream) {
    if (!mat || !out) return IAF_ERR_NULL;
    if (m <= 0 || n <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_colsum_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mat, out, m, n);
    return (int)hipGetLastError();
}

// weight prep of many plain convs in one launch (the four convs of every IAFLayer of a model): descriptors in device
// memory, refreshed per run like iaf_prep_batch_run
struct iaf_conv3x3_prep_batch {
    int n, ntiles;
    iaf_conv3x3** convs;
    PrepLayer* h_layers;   // current descriptor table (host)
    DescTable tab;         // its way to the device (see DescTable in iaf_engine.hip)
    int* d_tile2layer;
};

extern "C" int iaf_conv3x3_prep_batch_destroy(iaf_conv3x3_prep_batch_t* b) {
    if (!b) return IAF_ERR_NULL;
    free(b->h_layers);
    desc_destroy(&b->tab);
    if (b->d_tile2layer) (void)hipFree(b->d_tile2layer);
    free(b->convs);
    delete b;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_prep_batch_create(iaf_conv3x3_prep_batch_t** out, iaf_conv3x3_t* const* convs, int n) {
    if (!out || !convs) return IAF_ERR_NULL;
    *out = nullptr;
    if (n <= 0) return IAF_ERR_SHAPE;
    iaf_conv3x3_prep_batch* b = new (std::nothrow) iaf_conv3x3_prep_batch();
    if (!b) return (int)hipErrorOutOfMemory;
    memset(b, 0, sizeof(*b));
    b->n = n;
    b->convs = (iaf_conv3x3**)calloc(n, sizeof(iaf_conv3x3*));
    int nt = 0;
    for (int i = 0; i < n; ++i) {
        if (!convs[i]) { iaf_conv3x3_prep_batch_destroy(b); return IAF_ERR_NULL; }
        if (convs[i]->generic || convs[i]->mask_mode) { iaf_conv3x3_prep_batch_destroy(b); return IAF_ERR_UNSUPPORTED; }
        b->convs[i] = convs[i];
        nt += convs[i]->L.ncot;
    }
    b->ntiles = nt;
    int* t2l = (int*)malloc(sizeof(int) * nt);
    int rc;
    b->h_layers = (PrepLayer*)calloc(n, sizeof(PrepLayer));
    if (!b->h_layers) { free(t2l); iaf_conv3x3_prep_batch_destroy(b); return (int)hipErrorOutOfMemory; }
    if ((rc = desc_init(&b->tab, sizeof(PrepLayer) * n)) != 0 ||
        (rc = (int)hipMalloc((void**)&b->d_tile2layer, sizeof(int) * nt)) != 0) {
        free(t2l); iaf_conv3x3_prep_batch_destroy(b); return rc;
    }
    int tile = 0;
    for (int i = 0; i < n; ++i) {
        const GemmLayer& L = convs[i]->L;
        PrepLayer& P = b->h_layers[i];
        P.bias = L.bias; P.variant = PREP_PLAIN9;
        conv_prep_packs(convs[i], P);
        P.cin = L.cin; P.cout_each = L.cout; P.ncot = L.ncot; P.nchunk = L.nchunk; P.npair = 1; P.tile_begin = tile;
        for (int t = 0; t < L.ncot; ++t) t2l[tile++] = i;
    }
    rc = (int)hipMemcpy(b->d_tile2layer, t2l, sizeof(int) * nt, hipMemcpyHostToDevice);
    free(t2l);
    if (rc) { iaf_conv3x3_prep_batch_destroy(b); return rc; }
    *out = b;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_prep_batch_run(iaf_conv3x3_prep_batch_t* b, const float* const* V, const float* const* g,
                                          const float* const* bias, void* stream) {
    if (!b || !V || !g || !bias) return IAF_ERR_NULL;
    bool changed = false;
    for (int i = 0; i < b->n; ++i) {
        if (!V[i] || !g[i] || !bias[i]) return IAF_ERR_NULL;
        PrepLayer& P = b->h_layers[i];
        PrepLayer N = P;                             // iaf_conv3x3_set_precision / _set_packs / _set_training / a range failure since the last run
        conv_prep_packs(b->convs[i], N);
        changed |= (P.V[0] != V[i]) | (P.g[0] != g[i]) | (P.b[0] != bias[i]) | (P.wpt != N.wpt) | (P.wp2 != N.wp2) | (P.rng_err != N.rng_err) |
                   (P.wp != N.wp) | (P.wp3 != N.wp3);
        P = N;
        P.V[0] = V[i]; P.g[0] = g[i]; P.b[0] = bias[i];
    }
    hipStream_t st = (hipStream_t)stream;
    const void* d_layers = nullptr;     // see iaf_prep_batch_run
    { int rc = desc_upload(&b->tab, b->h_layers, changed, st, &d_layers); if (rc) return rc; }
    hipLaunchKernelGGL(iaf_prep_plain_kernel, dim3(b->ntiles), dim3(256), 0, st, (const PrepLayer*)d_layers, b->d_tile2layer);
    HIP_TRY(hipGetLastError());
    {
        PackT3Batch tb(st);
        for (int i = 0; i < b->n; ++i)
            if (b->convs[i]->training) { int rc = tb.add(b->convs[i]->L, MAXTAPS); if (rc) return rc; }
        int rc = tb.flush();
        if (rc) return rc;
    }
    for (int i = 0; i < b->n; ++i) { b->convs[i]->prepared = true; b->convs[i]->deconv = false; }   // (packs of a conv2d now, as iaf_conv3x3_prepare leaves them)
    return IAF_OK;
}

// co tiles per workgroup (NT x WCO) of a 9-tap split-product launch need not divide the layer's tile count: the last workgroup's surplus
// tiles are computed on clamped fragments and dropped (iaf_conv_bf3.hpp, toff) -- accepted up to a quarter of the layer's tiles
static inline bool bf3_ragged_ok(int ncot, int per_wg) {
    if (per_wg <= 0) return false;
    const int covered = (ncot + per_wg - 1) / per_wg * per_wg;
    return (covered - ncot) * 4 <= ncot;
}
extern "C" int iaf_conv3x3_set_tuning(iaf_conv3x3_t* c, int nt, int pxt, int wco, int ks) {
    if (!c) return IAF_ERR_NULL;
    GemmLayer& L = c->L;
    if (nt == 0) { L.user_tuned = false; c->bf3_choice = 1; return IAF_OK; }     // back to the automatic choice
    if (nt < 0) {           // a bf16x3 launch shape, as iaf_conv3x3_autotune reports it: (-nt, ppw, wco, ks), pxt = 1
        if (c->generic || c->mask_mode || !L.wp3 || !bf3_ragged_ok(L.ncot, -nt * wco) || !pick_bf3_plain(-nt, pxt, 1, ks, wco)) return IAF_ERR_UNSUPPORTED;
        L.b_nt = -nt; L.b_ppw = pxt; L.b_pxt = 1; L.b_ks = ks; L.b_wco = wco;
        c->bf3_choice = 2;
        return IAF_OK;
    }
    if (c->generic || !pick_kernel(nt, pxt, wco, ks, IN_NCHW, c->mask_mode ? EPI_PLAIN5 : EPI_PLAIN)) return IAF_ERR_UNSUPPORTED;
    if (L.ncot % (nt * wco) != 0 || L.nchunk < ks) return IAF_ERR_UNSUPPORTED;
    L.nt = nt; L.pxt = pxt; L.wco = wco; L.ks = ks; L.user_tuned = true;
    c->bf3_choice = 3;      // a pinned fp32 shape means the fp32 kernel
    return IAF_OK;
}

// launch of the conv kernel for a plain / single masked conv descriptor (forward, or its transposed problem with the
// taps mirrored for the data gradient)
// the plain conv on the bf16 matrix cores: which shape, if any, for this size (choice: see iaf_conv3x3.bf3_choice)
static bool conv3x3_bf3_shape(GemmLayer& L, int choice, long long P, int W, bool f16 = false) {
    if (!L.wp3 || choice == 3) return false;
    if (choice != 2) {
        // size rule: worth it from ~4096 pixels on (as for the masked stack); on the fp16 planes (half the MFMAs) from 2048
        if (P < (f16 ? 2048 : 4096)) return false;
        // (nt, wco) at ppw 2, ks 4 -- what iaf_conv3x3_autotune finds at the BASELINE sizes (round 6): every workgroup of a 32-pixel block
        // stages the block's tile again, so the FEWEST workgroups per block that still give the chip a workgroup per CU, at most an
        // eighth of the co tiles wasted on a ragged last group; else whatever gives the most workgroups
        static const int cand[8][2] = {{4, 3}, {5, 2}, {4, 2}, {2, 3}, {5, 1}, {4, 1}, {2, 2}, {2, 1}};
        const long long nblk = (P + 31) / 32;
        int best = -1;
        long long best_wgs = 0;
        for (int i = 0; i < 8; ++i) {
            const int nt = cand[i][0], wco = cand[i][1], per = nt * wco;
            const int covered = (L.ncot + per - 1) / per * per;
            if ((covered - L.ncot) * 8 > L.ncot || !pick_bf3_plain(nt, 2, 1, 4, wco)) continue;
            if (bf3_plain_lds_bytes(L.cin, W, nt, 2, 1, 4, wco, f16 ? 2 : 3) > 160 * 1024) continue;
            const long long wgs = nblk * (covered / per);
            if (wgs >= 256) { best = i; break; }
            if (wgs > best_wgs) { best_wgs = wgs; best = i; }
        }
        if (best < 0) return false;
        L.b_nt = cand[best][0]; L.b_ppw = 2; L.b_pxt = 1; L.b_ks = 4; L.b_wco = cand[best][1];
    }
    if (!bf3_ragged_ok(L.ncot, L.b_nt * L.b_wco) || !pick_bf3_plain(L.b_nt, L.b_ppw, L.b_pxt, L.b_ks, L.b_wco)) return false;
    return bf3_plain_lds_bytes(L.cin, W, L.b_nt, L.b_ppw, L.b_pxt, L.b_ks, L.b_wco, f16 ? 2 : 3) <= 160 * 1024;
}

// dev tool (tools/conv_stamps.py): per-workgroup cycle stamps of THIS conv's bf16x3 plain-conv launches, [grid][8] u64 in a caller
// buffer of `bytes` bytes (NULL: off).  Per object and size-checked at every launch (it was one process-global pointer
// that every conv's launches on every stream wrote through without a bound); declared here, defined behind the object's definition.

// dgrad16: the data gradient (bwd3) of a conv whose arithmetic is IAF_PRECISION_F16X2 -- two fp16 planes with a tile-local scale
static int conv3x3_launch(GemmLayer& L, ConvP& p, int epi_sel, int inmode, bool masked, bool mirror, hipStream_t st,
                          int variant = IAF_VARIANT_TF, int bf3_choice = 3, unsigned* f16_rng = nullptr, bool dgrad16 = false) {
    // the plain conv on the bf16 matrix cores: forward (NCHW input, EPI_PLAIN), or its data gradient (L = the transposed
    // problem with wp3 = the transposed bf16x3 pack: dY pixel-major, mirrored taps, EPI_DGRAD)
    const bool fwd3 = !masked && !mirror && epi_sel == EPI_PLAIN && inmode == IN_NCHW;
    const bool bwd3 = !masked && mirror && epi_sel == EPI_DGRAD9 && inmode == IN_PIXMAJOR && variant == IAF_VARIANT_TF;
    const bool f16l = (fwd3 && f16_rng != nullptr) || (bwd3 && dgrad16 && L.wp2 != nullptr);
    if ((fwd3 || bwd3) && conv3x3_bf3_shape(L, bf3_choice, p.P, p.W, f16l) &&
        pick_bf3_plain(L.b_nt, L.b_ppw, L.b_pxt, L.b_ks, L.b_wco, bwd3 ? EPI_DGRAD : EPI_PLAIN)) {
        conv_fn_t fn = pick_bf3_plain(L.b_nt, L.b_ppw, L.b_pxt, L.b_ks, L.b_wco, bwd3 ? EPI_DGRAD : EPI_PLAIN);
        const int tm = 16 * L.b_ppw * L.b_pxt, W = p.W, sg = bwd3 ? -1 : 1;
        p.border = nullptr; p.wp = (const float*)L.wp3; p.bias = L.bias; p.lim = nullptr;
        // f16_rng (the caller's conv runs IAF_PRECISION_F16X2): the forward on two fp16 planes, the same launch shape and LDS bound
        bool npl2 = false;
        if (conv_fn_t fn16 = (fwd3 && f16_rng && L.wp2) ? pick_bf3_plain_f16(L.b_nt, L.b_ppw, L.b_pxt, L.b_ks, L.b_wco) : nullptr) {
            fn = fn16; p.wp = (const float*)L.wp2; p.rng_err = f16_rng; npl2 = true;
        } else if (conv_fn_t fd16 = (bwd3 && dgrad16 && L.wp2) ? pick_bf3_plain_f16d(L.b_nt, L.b_ppw, L.b_pxt, L.b_ks, L.b_wco) : nullptr) {
            fn = fd16; p.wp = (const float*)L.wp2; p.rng_err = nullptr; npl2 = true;
        } else if (f16l) {
            return IAF_ERR_UNSUPPORTED;         // (the shape was sized for two planes and no two-plane kernel exists for it: not reached by the compiled shape lists)
        }
        for (int t = 0; t < MAXTAPS; ++t) { p.tap_dh[t] = sg * (t / 3 - 1); p.tap_dw[t] = sg * (t % 3 - 1); }   // cross-correlation, SAME (mirrored: dX)
        p.halo_before = W + 1;
        p.nslot = tm + 2 * (W + 1);
        p.cin = L.cin; p.cout = L.cout; p.nchunk = L.nchunk; p.ncot = L.ncot; p.cp = L.cin + 8;
        const size_t lds = bf3_plain_lds_bytes(L.cin, W, L.b_nt, L.b_ppw, L.b_pxt, L.b_ks, L.b_wco, npl2 ? 2 : 3);
        int rc = raise_lds_cap((const void*)fn, lds);
        if (rc) return rc;
        dim3 grid((p.P + tm - 1) / tm, (L.ncot + L.b_nt * L.b_wco - 1) / (L.b_nt * L.b_wco));
        p.gx = (int)grid.x;
        p.lds_bytes = (int)lds;
        p.dbg = (L.dbg && (size_t)grid.x * grid.y * 8 * sizeof(unsigned long long) <= L.dbg_bytes) ? L.dbg : nullptr;
        hipLaunchKernelGGL(fn, grid, dim3(64 * L.b_pxt * L.b_ks * L.b_wco), lds, st, p);
        return (int)hipGetLastError();
    }
    if (!L.user_tuned) {
        GemmLayer t = L;
        t.nt = L.t_nt; t.pxt = L.t_ppw; t.wco = L.t_pxt; t.ks = L.t_ks;
        if (L.tuned_P == p.P && L.tuned_W == p.W && L.t_nt > 0 && conv_lds_bytes(t, p.W) <= 160 * 1024) {
            L.nt = t.nt; L.pxt = t.pxt; L.wco = t.wco; L.ks = t.ks;     // measured for this size (iaf_conv3x3_autotune_backward)
        } else {
            auto_shape(L, false, p.P, p.W);
        }
    }
    conv_fn_t fn = pick_kernel(L.nt, L.pxt, L.wco, L.ks, inmode, epi_sel);
    if (!fn) return IAF_ERR_UNSUPPORTED;
    // Theano's true convolution looks left/above (taps negated); flipmask turns it back (see launch_gemm)
    const int tm = 16 * L.pxt, W = p.W, sgn = ((variant == IAF_VARIANT_THEANO) != mirror) ? -1 : 1;
    p.border = (masked && !mirror) ? L.border : nullptr;
    p.wp = L.wp; p.bias = L.bias; p.lim = nullptr;
    if (masked) {     // the 5 live taps of the MADE-masked filter: look right / below only
        static const int tf_dh[NTAPS] = {0, 0, 1, 1, 1}, tf_dw[NTAPS] = {0, 1, -1, 0, 1};
        for (int t = 0; t < NTAPS; ++t) { p.tap_dh[t] = sgn * tf_dh[t]; p.tap_dw[t] = sgn * tf_dw[t]; }
        p.halo_before = (sgn < 0) ? W + 1 : 0;
        p.nslot = tm + W + 1;
    } else {
        for (int t = 0; t < MAXTAPS; ++t) { p.tap_dh[t] = sgn * (t / 3 - 1); p.tap_dw[t] = sgn * (t % 3 - 1); }   // cross-correlation, SAME
        p.halo_before = W + 1;
        p.nslot = tm + 2 * (W + 1);
    }
    p.cin = L.cin; p.cout = L.cout; p.nchunk = L.nchunk; p.ncot = L.ncot;
    p.cp = L.cin + 8;
    const size_t lds = conv_lds_bytes(L, W);
    if (lds > 160 * 1024) return IAF_ERR_UNSUPPORTED;
    int rc = raise_lds_cap((const void*)fn, lds);
    if (rc) return rc;
    dim3 grid((p.P + tm - 1) / tm, L.ncot / (L.nt * L.wco));
    p.gx = (int)grid.x;
    p.lds_bytes = (int)lds;
    hipLaunchKernelGGL(fn, grid, dim3(64 * L.pxt * L.wco * L.ks), lds, st, p);
    return (int)hipGetLastError();
}

extern "C" int iaf_conv3x3_forward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                   const float* residual, float* const* outs, const int* out_channels, int n_outs, int B,
                                   int H, int W, void* stream) {
    if (!c || !x || !outs || !out_channels) return IAF_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || n_outs < 1 || n_outs > MAXSPLIT) return IAF_ERR_SHAPE;
    if ((long long)B * H * W > (1LL << 30) / 64) return IAF_ERR_SHAPE;
    if (!c->prepared) return IAF_ERR_NOT_PREPARED;
    if (x2 && (c_split <= 0 || c_split >= c->n_in)) return IAF_ERR_SHAPE;
    if (residual && n_outs != 1) return IAF_ERR_SHAPE;
    int tot = 0, ends[MAXSPLIT];
    for (int k = 0; k < n_outs; ++k) {
        if (!outs[k]) return IAF_ERR_NULL;
        if (out_channels[k] <= 0) return IAF_ERR_SHAPE;
        tot += out_channels[k];
        ends[k] = tot;
    }
    if (tot != c->n_out) return IAF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    GemmLayer& L = c->L;
    if (c->generic) {
        GenPlainP p;
        memset(&p, 0, sizeof(p));
        p.x = x; p.x2 = x2; p.w = L.wp; p.bias = L.bias; p.res = residual;
        p.B = B; p.H = H; p.W = W; p.cin = L.cin; p.cout = L.cout; p.c_split = c_split; p.in_elu = elu_input ? 1 : 0;
        p.nsplit = n_outs;
        for (int k = 0; k < n_outs; ++k) { p.split_end[k] = ends[k]; p.split_ptr[k] = outs[k]; }
        hipLaunchKernelGGL(iaf_generic_conv3x3_kernel, ew_grid((size_t)B * L.cout * H * W), dim3(256), 0, st, p);
        return (int)hipGetLastError();
    }
    // MFMA path: a lane owns 4 consecutive channels, so the concat point and the split points must be multiples of 4
    if (x2 && (c_split & 3)) return IAF_ERR_UNSUPPORTED;
    for (int k = 0; k < n_outs; ++k)
        if (ends[k] & 3) return IAF_ERR_UNSUPPORTED;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.x = x; p.x2 = x2; p.c_split = c_split; p.in_elu = elu_input ? 1 : 0; p.res = residual;
    p.nsplit = n_outs;
    for (int k = 0; k < MAXSPLIT; ++k) {
        p.split_end[k] = ends[k < n_outs ? k : n_outs - 1];
        p.split_ptr[k] = outs[k < n_outs ? k : n_outs - 1];
    }
    // an F16 launch of this conv (or the prep of its pack) met an operand beyond fp16's range: said once; bf16x3 from here on
    if (c->precision == IAF_PRECISION_F16X2 && !c->f16_off && c->rng_err_host && *(volatile unsigned*)c->rng_err_host) {
        c->f16_off = true;
        if (!(c->packs & IAF_PACK_BF16X3)) {           // (iaf_conv3x3_set_packs: the bf16x3 pack was not kept up to date -- every pack from the next prepare on)
            c->packs = IAF_PACK_F32 | IAF_PACK_BF16X3 | IAF_PACK_F16X2;
            c->prepared = false;
        }
        return IAF_ERR_RANGE;
    }
    if (!c->mask_mode && c->packs != (IAF_PACK_F32 | IAF_PACK_BF16X3 | IAF_PACK_F16X2)) {
        // iaf_conv3x3_set_packs: the pack this launch reads must be one the prep launches write
        GemmLayer t = L;
        const bool split = conv_split(c) && !c->deconv && conv3x3_bf3_shape(t, c->bf3_choice, p.P, W, conv_f16_active(c));
        const bool f16 = split && conv_f16_active(c) && pick_bf3_plain_f16(t.b_nt, t.b_ppw, t.b_pxt, t.b_ks, t.b_wco);
        const int need = f16 ? IAF_PACK_F16X2 : split ? IAF_PACK_BF16X3 : IAF_PACK_F32;
        if (!(c->packs & need)) return IAF_ERR_NOT_PREPARED;
    }
    return conv3x3_launch(L, p, c->mask_mode ? EPI_PLAIN5 : EPI_PLAIN, IN_NCHW, c->mask_mode != 0, false, st, c->variant,
                          (conv_split(c) && !c->deconv) ? c->bf3_choice : 3, conv_f16_active(c) ? c->rng_err_dev : nullptr);
}

// ---- the downsampling IAFLayer's two strided convs at their minimal work (iaf_conv_bf3.hpp, template parameter S2) ------------
extern "C" conv_fn_t iaf_pick_bf3s_2_1_4_1(int nt, int s2);
extern "C" conv_fn_t iaf_pick_bf3s_4_1_4_1(int nt, int s2);
extern "C" conv_fn_t iaf_pick_bf3s_2_1_4_2(int nt, int s2);
static conv_fn_t pick_bf3_s2(int nt, int ppw, int ks, int wco, int s2) {
    if (ppw == 2 && ks == 4 && wco == 1) return iaf_pick_bf3s_2_1_4_1(nt, s2);
    if (ppw == 4 && ks == 4 && wco == 1) return iaf_pick_bf3s_4_1_4_1(nt, s2);
    if (ppw == 2 && ks == 4 && wco == 2) return iaf_pick_bf3s_2_1_4_2(nt, s2);
    return nullptr;
}
// launch shape (nt, ppw, ks, wco; pxt = 1) of a strided conv: the first compiled one that divides the co tiles and fits LDS.
// IAF_S2_SHAPE / IAF_DECONV_SHAPE="nt,ppw,ks,wco": dev override.
static conv_fn_t s2_shape(const GemmLayer& L, int s2, int W, int* sh, size_t* lds_out) {
    static const int cand[2][6][4] = {
        {{4, 2, 4, 2}, {5, 2, 4, 2}, {2, 2, 4, 2}, {4, 2, 4, 1}, {5, 2, 4, 1}, {2, 2, 4, 1}},       // stride 2: 32 output pixels (4 x 32 staged)
        {{5, 4, 4, 1}, {4, 4, 4, 1}, {2, 4, 4, 1}, {5, 2, 4, 2}, {4, 2, 4, 2}, {2, 2, 4, 2}}};      // deconv: 64 input pixels
    int env[4] = {0, 0, 0, 0};
    const char* e = getenv(s2 == 1 ? "IAF_S2_SHAPE" : "IAF_DECONV_SHAPE");
    const bool has_env = e && sscanf(e, "%d,%d,%d,%d", &env[0], &env[1], &env[2], &env[3]) == 4;
    for (int i = has_env ? -1 : 0; i < 6; ++i) {
        const int* q = i < 0 ? env : cand[s2 - 1][i];
        const int nt = q[0], ppw = q[1], ks = q[2], wco = q[3];
        if (nt <= 0 || wco <= 0 || L.ncot % (nt * wco) != 0) continue;
        conv_fn_t fn = pick_bf3_s2(nt, ppw, ks, wco, s2);
        if (!fn) continue;
        const int tm = 16 * ppw;
        const size_t slots = s2 == 1 ? (size_t)4 * tm + 2 * W + 2 + 1 : (size_t)tm + W + 1 + 1;
        const size_t tile = slots * (3 * (L.cin / 8) + 2) * 16;
        const size_t red = (size_t)wco * ks * ppw * nt * 1024;
        const size_t lds = tile > red ? tile : red;
        if (lds > 160 * 1024) continue;
        sh[0] = nt; sh[1] = ppw; sh[2] = ks; sh[3] = wco;
        *lds_out = lds;
        return fn;
    }
    return nullptr;
}

// y = conv2d(name, [elu](x), n_out, stride=[2,2]) (tf_train.py:33,36; layers.py:31-64: SAME, so window (i,j) covers input rows 2i..2i+2)
// split into n_outs tensors like iaf_conv3x3_forward.  x [B,n_in,2H,2W]; H, W = the OUTPUT size.  Nine taps per c_in on H x W pixels --
// a quarter of the stride-1 conv + subsampling it replaces.  IAF_ERR_UNSUPPORTED when the four staged phase tiles do not fit LDS
// (then: iaf_conv3x3_forward at [2H,2W] + iaf_resample2 DOWN_ODD, the same numbers).
extern "C" int iaf_conv3x3_forward_stride2(iaf_conv3x3_t* c, const float* x, int elu_input, float* const* outs, const int* out_channels,
                                           int n_outs, int B, int H, int W, void* stream) {
    if (!c || !x || !outs || !out_channels) return IAF_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || n_outs < 1 || n_outs > MAXSPLIT) return IAF_ERR_SHAPE;
    if ((long long)B * H * W > (1LL << 30) / 64) return IAF_ERR_SHAPE;
    if (!c->prepared) return IAF_ERR_NOT_PREPARED;
    int tot = 0, ends[MAXSPLIT];
    for (int k = 0; k < n_outs; ++k) {
        if (!outs[k]) return IAF_ERR_NULL;
        if (out_channels[k] <= 0) return IAF_ERR_SHAPE;
        tot += out_channels[k];
        ends[k] = tot;
    }
    if (tot != c->n_out) return IAF_ERR_SHAPE;
    GemmLayer& L = c->L;
    if (c->generic || c->mask_mode || c->deconv || !L.wp3 || !conv_split(c)) return IAF_ERR_UNSUPPORTED;
    if (!(c->packs & IAF_PACK_BF16X3)) return IAF_ERR_NOT_PREPARED;     // (iaf_conv3x3_set_packs: the strided form reads the bf16 planes)
    for (int k = 0; k < n_outs; ++k)
        if (ends[k] & 3) return IAF_ERR_UNSUPPORTED;
    int sh[4]; size_t lds = 0;
    conv_fn_t fn = s2_shape(L, 1, W, sh, &lds);
    if (!fn) return IAF_ERR_UNSUPPORTED;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.x = x; p.in_elu = elu_input ? 1 : 0;
    p.nsplit = n_outs;
    for (int k = 0; k < MAXSPLIT; ++k) {
        p.split_end[k] = ends[k < n_outs ? k : n_outs - 1];
        p.split_ptr[k] = outs[k < n_outs ? k : n_outs - 1];
    }
    const int tm = 16 * sh[1];
    p.wp = (const float*)L.wp3; p.bias = L.bias;
    p.s2_pb[0] = 0; p.s2_pb[1] = tm + W + 1; p.s2_pb[2] = p.s2_pb[1] + tm + W; p.s2_pb[3] = p.s2_pb[2] + tm + 1;
    p.nslot = p.s2_pb[3] + tm;
    p.halo_before = 0;
    for (int t = 0; t < MAXTAPS; ++t) {
        const int di = t / 3, dj = t % 3;
        p.tap_dh[t] = di >> 1; p.tap_dw[t] = dj >> 1;
        p.tap_off[t] = p.s2_pb[(di & 1) * 2 + (dj & 1)] + (di >> 1) * W + (dj >> 1);
    }
    p.cin = L.cin; p.cout = L.cout; p.nchunk = L.nchunk; p.ncot = L.ncot; p.cp = L.cin + 8;
    int rc = raise_lds_cap((const void*)fn, lds);
    if (rc) return rc;
    dim3 grid((p.P + tm - 1) / tm, L.ncot / (sh[0] * sh[3]));
    p.gx = (int)grid.x;
    p.lds_bytes = (int)lds;
    hipLaunchKernelGGL(fn, grid, dim3(64 * sh[2] * sh[3]), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// out = [residual upsampled +] 0.1 x deconv2d(name, [elu](concat(x[:, :c_split], x2)), n_out) (tf_train.py:87-94; layers.py:83-112) for a
// conv prepared by iaf_conv3x3_prepare_deconv.  x (, x2) [B,.,H,W] -- H, W = the INPUT size --, out [B,n_out,2H,2W], residual
// [B,n_out,H,W] or NULL (without it: out = the deconv itself).  Each of the four output phases (2i+a, 2j+b) is a conv of the
// low-resolution input with the 4 / 2 / 2 / 1 filter taps that meet non-zero rows of the zero-inserted image: nine taps per input pixel
// instead of 36.  IAF_ERR_UNSUPPORTED: run iaf_conv3x3_forward on the zero-inserted inputs (same numbers up to fp32 rounding).
extern "C" int iaf_conv3x3_forward_deconv(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                          const float* residual, float* out, int B, int H, int W, void* stream) {
    if (!c || !x || !out) return IAF_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0) return IAF_ERR_SHAPE;
    if ((long long)B * H * W > (1LL << 30) / 256) return IAF_ERR_SHAPE;
    if (!c->prepared) return IAF_ERR_NOT_PREPARED;
    if (x2 && (c_split <= 0 || c_split >= c->n_in)) return IAF_ERR_SHAPE;
    GemmLayer& L = c->L;
    if (!c->deconv) return IAF_ERR_NOT_PREPARED;
    if (c->generic || c->mask_mode || !L.wp3 || L.nchunk % 2 != 0 || !conv_split(c)) return IAF_ERR_UNSUPPORTED;
    if ((x2 && (c_split & 3)) || (L.cout & 3)) return IAF_ERR_UNSUPPORTED;
    int sh[4]; size_t lds = 0;
    conv_fn_t fn = s2_shape(L, 2, W, sh, &lds);
    if (!fn) return IAF_ERR_UNSUPPORTED;
    ConvP p;
    memset(&p, 0, sizeof(p));
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = B * H * W;
    p.x = x; p.x2 = x2; p.c_split = c_split; p.in_elu = elu_input ? 1 : 0; p.res = residual;
    p.nsplit = 1;
    for (int k = 0; k < MAXSPLIT; ++k) { p.split_end[k] = c->n_out; p.split_ptr[k] = out; }
    const int tm = 16 * sh[1];
    p.wp = (const float*)L.wp3; p.bias = L.bias;
    // tap (di,dj) of the rotated filter on the zero-inserted image = rows i-1 (di = 0) or i (di = 1, 2) of the input
    for (int t = 0; t < MAXTAPS; ++t) { p.tap_dh[t] = (t / 3 == 0) ? -1 : 0; p.tap_dw[t] = (t % 3 == 0) ? -1 : 0; }
    p.halo_before = W + 1;
    p.nslot = tm + W + 1;
    p.cin = L.cin; p.cout = L.cout; p.nchunk = L.nchunk; p.ncot = L.ncot; p.cp = L.cin + 8;
    int rc = raise_lds_cap((const void*)fn, lds);
    if (rc) return rc;
    dim3 grid((p.P + tm - 1) / tm, L.ncot / (sh[0] * sh[3]), 4);
    p.gx = (int)grid.x;
    p.lds_bytes = (int)lds;
    hipLaunchKernelGGL(fn, grid, dim3(64 * sh[2] * sh[3]), lds, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// arithmetic of the forward conv: IAF_PRECISION_BF16X3 (default: split products on the bf16 matrix cores where a launch
// shape covers the problem, fp32-grade) or IAF_PRECISION_F32 (the exact-fp32 MFMA kernel always)
// which packs the prep launches of a plain conv keep up to date (as iaf_stack_set_packs): a conv that runs at ONE size needs one
extern "C" int iaf_conv3x3_set_packs(iaf_conv3x3_t* c, int packs) {
    if (!c) return IAF_ERR_NULL;
    if (packs & ~(IAF_PACK_F32 | IAF_PACK_BF16X3 | IAF_PACK_F16X2)) return IAF_ERR_SHAPE;
    if (!packs) return IAF_ERR_SHAPE;
    const int all = IAF_PACK_F32 | IAF_PACK_BF16X3 | IAF_PACK_F16X2;
    if (packs != all && (c->generic || c->mask_mode || c->training || c->deconv)) return IAF_ERR_UNSUPPORTED;
    if ((packs & IAF_PACK_BF16X3) && !(packs & IAF_PACK_F32) && !c->L.wp3) return IAF_ERR_UNSUPPORTED;      // (no split pack for this conv)
    if ((packs & IAF_PACK_F16X2) && !(packs & (IAF_PACK_F32 | IAF_PACK_BF16X3)) && !conv_f16_active(c)) return IAF_ERR_UNSUPPORTED;
    if (packs & ~c->packs) c->prepared = false;              // a pack that was not kept up to date comes back: the next prepare fills it
    c->packs = packs;
    return IAF_OK;
}
extern "C" int iaf_conv3x3_set_precision(iaf_conv3x3_t* c, int precision) {
    if (!c) return IAF_ERR_NULL;
    if (precision != IAF_PRECISION_F32 && precision != IAF_PRECISION_BF16X3 && precision != IAF_PRECISION_F16X2) return IAF_ERR_SHAPE;
    if (precision == IAF_PRECISION_F16X2) {
        GemmLayer& L = c->L;
        if (c->generic || c->mask_mode || !L.wp3) return IAF_ERR_UNSUPPORTED;       // (plain convs with c_in % 32 == 0: the same fragments, two planes)
        if (!c->rng_err_host) {
            if (hipHostMalloc((void**)&c->rng_err_host, 64, hipHostMallocMapped) != hipSuccess) { c->rng_err_host = nullptr; return (int)hipErrorOutOfMemory; }
            *(volatile unsigned*)c->rng_err_host = 0u;
            if (hipHostGetDevicePointer((void**)&c->rng_err_dev, c->rng_err_host, 0) != hipSuccess) {
                (void)hipHostFree(c->rng_err_host); c->rng_err_host = nullptr; c->rng_err_dev = nullptr;
                return (int)hipErrorOutOfMemory;
            }
        }
        if (!L.wp2) {
            HIP_TRY(hipMalloc(&L.wp2, (size_t)(L.cin / 32) * MAXTAPS * L.ncot * 2 * 1024));
            c->prepared = false;                             // the next prepare fills it
        }
        if (c->f16_off || *(volatile unsigned*)c->rng_err_host) {        // re-armed after a range failure
            HIP_TRY(hipDeviceSynchronize());
            *(volatile unsigned*)c->rng_err_host = 0u;
            c->f16_off = false;
            c->prepared = false;
        }
        if (c->precision != IAF_PRECISION_F16X2) c->prepared = false;     // the fp16 pack has not been kept up to date
    }
    c->precision = precision;
    return IAF_OK;
}
extern "C" int iaf_conv3x3_range_errors(const iaf_conv3x3_t* c, unsigned* errors) {
    if (!c || !errors) return IAF_ERR_NULL;
    *errors = 0;
    if (!c->rng_err_host) return IAF_OK;
    HIP_TRY(hipDeviceSynchronize());
    *errors = *(volatile unsigned*)c->rng_err_host;
    return IAF_OK;
}
// 1 if a forward call at this size would run the bf16x3 kernel
extern "C" int iaf_conv3x3_runs_bf16x3(iaf_conv3x3_t* c, int B, int H, int W) {
    if (!c || c->generic || c->mask_mode || c->deconv || !conv_split(c)) return 0;
    GemmLayer t = c->L;
    return conv3x3_bf3_shape(t, c->bf3_choice, (long long)B * H * W, W, conv_f16_active(c)) ? 1 : 0;
}
// ... and on two fp16 planes (the split-product launch of this size is an F16 instantiation)
extern "C" int iaf_conv3x3_runs_f16x2(iaf_conv3x3_t* c, int B, int H, int W) {
    if (!iaf_conv3x3_runs_bf16x3(c, B, H, W) || !conv_f16_active(c)) return 0;
    GemmLayer t = c->L;
    if (!conv3x3_bf3_shape(t, c->bf3_choice, (long long)B * H * W, W, true)) return 0;
    return pick_bf3_plain_f16(t.b_nt, t.b_ppw, t.b_pxt, t.b_ks, t.b_wco) ? 1 : 0;
}

extern "C" int iaf_conv3x3_autotune(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                    const float* residual, float* const* outs, const int* out_channels, int n_outs, int B,
                                    int H, int W, int reps, void* stream, int* best_shape, float* best_us) {
    if (!c) return IAF_ERR_NULL;
    if (c->generic) return IAF_OK;
    if (reps <= 0) reps = 20;
    hipStream_t st = (hipStream_t)stream;
    GemmLayer& L = c->L;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float best = 1e30f;
    int bsh[4] = {L.nt, L.pxt, L.wco, L.ks};
    int rc = IAF_OK;
    c->bf3_choice = 3;                  // first the exact-fp32 kernel in every shape ...
    for (int si = 0; si < 8 && rc == IAF_OK; ++si)
        for (int nt = 5; nt >= 1 && rc == IAF_OK; --nt) {
            const int pxt = k_shapes[si][0], wco = k_shapes[si][1], ks = k_shapes[si][2];
            if (L.ncot % (nt * wco) != 0 || L.nchunk < ks) continue;
            GemmLayer t = L;
            t.nt = nt; t.pxt = pxt; t.wco = wco; t.ks = ks;
            if (conv_lds_bytes(t, W) > 160 * 1024) continue;
            L.nt = nt; L.pxt = pxt; L.wco = wco; L.ks = ks; L.user_tuned = true;
            for (int r = 0; r < 3 && rc == IAF_OK; ++r)
                rc = iaf_conv3x3_forward(c, x, x2, c_split, elu_input, residual, outs, out_channels, n_outs, B, H, W, stream);
            if (rc == IAF_ERR_UNSUPPORTED) { rc = IAF_OK; continue; }
            if (rc) break;
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps && rc == IAF_OK; ++r)
                rc = iaf_conv3x3_forward(c, x, x2, c_split, elu_input, residual, outs, out_channels, n_outs, B, H, W, stream);
            (void)hipEventRecord(e1, st);
            if (rc) break;
            if ((rc = (int)hipEventSynchronize(e1)) != 0) break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; bsh[0] = nt; bsh[1] = pxt; bsh[2] = wco; bsh[3] = ks; }
        }
    L.nt = bsh[0]; L.pxt = bsh[1]; L.wco = bsh[2]; L.ks = bsh[3]; L.user_tuned = true;
    // ... and the same conv on the bf16 matrix cores, every compiled 9-tap shape
    int bz[5] = {0, 0, 0, 0, 1};
    if (rc == IAF_OK && L.wp3 && conv_split(c) && !c->mask_mode && !c->deconv) {
        static const int nts[3] = {5, 4, 2};
        for (int si = 0; si < N_BF3P_SHAPES && rc == IAF_OK; ++si)
            for (int nt : nts) {
                const int* sh = k_bf3p_shapes[si];
                if (!bf3_ragged_ok(L.ncot, nt * sh[3])) continue;
                L.b_nt = nt; L.b_ppw = sh[0]; L.b_pxt = sh[1]; L.b_ks = sh[2]; L.b_wco = sh[3];
                c->bf3_choice = 2;
                GemmLayer t = L;
                if (!conv3x3_bf3_shape(t, 2, (long long)B * H * W, W)) continue;
                for (int r = 0; r < 3 && rc == IAF_OK; ++r)
                    rc = iaf_conv3x3_forward(c, x, x2, c_split, elu_input, residual, outs, out_channels, n_outs, B, H, W, stream);
                if (rc) break;
                (void)hipEventRecord(e0, st);
                for (int r = 0; r < reps && rc == IAF_OK; ++r)
                    rc = iaf_conv3x3_forward(c, x, x2, c_split, elu_input, residual, outs, out_channels, n_outs, B, H, W, stream);
                (void)hipEventRecord(e1, st);
                if (rc) break;
                if ((rc = (int)hipEventSynchronize(e1)) != 0) break;
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; bz[0] = nt; bz[1] = sh[0]; bz[2] = sh[1]; bz[3] = sh[2]; bz[4] = sh[3]; }
            }
        if (bz[0]) { L.b_nt = bz[0]; L.b_ppw = bz[1]; L.b_pxt = bz[2]; L.b_ks = bz[3]; L.b_wco = bz[4]; c->bf3_choice = 2; }
        else c->bf3_choice = 3;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (best_shape) {
        if (bz[0]) { best_shape[0] = -bz[0]; best_shape[1] = bz[1]; best_shape[2] = bz[4]; best_shape[3] = bz[3]; }   // (-nt, ppw, wco, ks)
        else for (int i = 0; i < 4; ++i) best_shape[i] = bsh[i];
    }
    if (best_us) *best_us = best * 1e3f / reps;
    return rc;
}

extern "C" int iaf_conv3x3_work(const iaf_conv3x3_t* c, int B, int H, int W, double* flops, double* bytes) {
    if (!c) return IAF_ERR_NULL;
    const double P = (double)B * H * W;
    if (flops) *flops = 2.0 * c->L.live_macs_per_px * P;
    // input + output activations once, raw V/g/b once
    if (bytes) *bytes = 4.0 * (P * (c->n_in + c->n_out) + 9.0 * c->n_in * c->n_out + 2.0 * c->n_out);
    return IAF_OK;
}

extern "C" int iaf_datainit_normalize(const float* x_init, const float* add, float* y, float* g, float* b, int B, int C,
                                      int HW, float init_scale, void* stream) {
    if (!x_init || !g || !b) return IAF_ERR_NULL;
    if (B <= 0 || C <= 0 || HW <= 0) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_datainit_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x_init, add, y, g, b, B, C, HW, init_scale);
    return (int)hipGetLastError();
}

extern "C" int iaf_discretized_logistic(const float* mean, const float* logscale, int logscale_is_scalar, const float* sample,
                                        float* out, int B, size_t n_per_row, float binsize, void* stream) {
    if (!mean || !logscale || !sample || !out) return IAF_ERR_NULL;
    if (B <= 0 || n_per_row == 0 || !(binsize > 0.f)) return IAF_ERR_SHAPE;
    hipLaunchKernelGGL(iaf_disc_logistic_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mean, logscale,
                       logscale_is_scalar ? 1 : 0, sample, out, n_per_row, binsize);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// backward of a plain conv (what TF's autodiff derives for layers.py:52-64 inside IAFLayer, tf_train.py:138):
//   forward   y = conv(a, w) + b,  a = act(concat(x, x2)),  w = exp(g) V / ||V||_o
//   given     dY (as up to 6 NCHW tensors -- the gradients of the split outputs -- times dy_scale)
//   computes  dX = [dx_residual +] act'(.) * (W^T dY)   -> NCHW, split like the forward concat
//             dV, dg (through the weight norm), db
// Passes: (1) pack dY and a pixel-major, (2) data gradient = the SAME conv kernel on the transposed packs with mirrored
// taps (EPI_DGRAD, 9 taps), (3) MFMA weight gradient over pixel ranges + reduce, (4) weight-norm backward.
// ---------------------------------------------------------------------------------------------
extern "C" int iaf_conv3x3_set_debug(iaf_conv3x3_t* c, void* buf, size_t bytes) {
    if (!c) return IAF_ERR_NULL;
    c->L.dbg = (unsigned long long*)buf; c->L.dbg_bytes = buf ? bytes : 0;
    c->T.dbg = c->L.dbg; c->T.dbg_bytes = c->L.dbg_bytes;      // (the data gradient runs the transposed problem T)
    return IAF_OK;
}

extern "C" int iaf_conv3x3_set_training(iaf_conv3x3_t* c, int on) {
    if (!c) return IAF_ERR_NULL;
    if (c->mask_mode) return IAF_ERR_UNSUPPORTED;
    if (!on) { c->training = false; return IAF_OK; }
    if (c->generic) {         // channel counts outside the MFMA path: the direct backward kernels (iaf_kernels_generic.hpp), no extra packs
        c->training = true;
        return IAF_OK;
    }
    GemmLayer& L = c->L;
    if (!L.wpt) HIP_TRY(hipMalloc(&L.wpt, (size_t)L.nchunk * MAXTAPS * L.ncot * 256 * sizeof(float)));
    // the transposed pack as bf16x3 (iaf_pack_t3_kernel): the data gradient on the bf16 matrix cores (even K tile counts)
    if (!L.wpt3 && L.ncot % 2 == 0) HIP_TRY(hipMalloc(&L.wpt3, (size_t)(L.ncot / 2) * MAXTAPS * L.nchunk * 3 * 64 * 16));
    // ... and as two fp16 planes: the data gradient of a conv whose arithmetic is IAF_PRECISION_F16X2 (iaf_conv_bf3.hpp DG16; IAF_DGRAD_F16=0: dev knob)
    static const bool dg16_env = !(getenv("IAF_DGRAD_F16") && getenv("IAF_DGRAD_F16")[0] == '0');
    if (dg16_env && L.wpt3 && !L.wpt2) HIP_TRY(hipMalloc(&L.wpt2, (size_t)(L.ncot / 2) * MAXTAPS * L.nchunk * 2 * 64 * 16));
    GemmLayer& T = c->T;
    T = GemmLayer();
    T.cin = L.cout; T.cout = L.cin; T.nchunk = L.ncot; T.ncot = L.nchunk; T.zerodiag = 0; T.npair = 1; T.full3x3 = true;
    T.wp = L.wpt; T.wp3 = L.wpt3; T.wp2 = L.wpt2; T.nt = 1; T.pxt = 4; T.wco = 1; T.ks = 1; T.user_tuned = false;
    T.dbg = L.dbg; T.dbg_bytes = L.dbg_bytes;
    c->training = true;
    c->packs = IAF_PACK_F32 | IAF_PACK_BF16X3 | IAF_PACK_F16X2;     // (training keeps every pack)
    c->prepared = false;      // the transposed pack is written by the next prepare
    return IAF_OK;
}

struct ConvTrainWs { float* xe; float* dyc; float* part; float* dW; float* dbp; unsigned short* tapmask; };
static size_t conv3x3_train_ws_floats(const iaf_conv3x3* c, long long P, ConvTrainWs* o, float* base) {
    size_t off = 0;
    auto take = [&](size_t n) { float* q = base ? base + off : nullptr; off += (n + 63) / 64 * 64; return q; };
    ConvTrainWs t;
    t.xe = take((size_t)P * c->n_in);
    t.dyc = take((size_t)P * c->n_out);
    t.part = take((size_t)WGRAD_MAX_RANGES * MAXTAPS * c->n_in * c->n_out);
    t.dW = take((size_t)MAXTAPS * c->n_in * c->n_out);
    t.dbp = take((size_t)256 * c->n_out);
    t.tapmask = (unsigned short*)take(((size_t)P + 1) / 2);
    if (o) *o = t;
    return off;
}

extern "C" size_t iaf_conv3x3_train_workspace_bytes(const iaf_conv3x3_t* c, int B, int H, int W) {
    if (!c || B <= 0 || H <= 0 || W <= 0) return 0;
    return conv3x3_train_ws_floats(c, (long long)B * H * W, nullptr, nullptr) * sizeof(float);
}

static void pack_fill(PackP& p, const float* const* src, const int* chans, int n, float* dst, int C, int HW, int P, float scale, int elu) {
    memset(&p, 0, sizeof(p));
    int tot = 0;
    for (int k = 0; k < n; ++k) { tot += chans[k]; p.src[k] = src[k]; p.end[k] = tot; }
    p.nsrc = n; p.dst = dst; p.C = C; p.HW = HW; p.P = P; p.scale = scale; p.elu = elu;
}
// the two backward operands of a conv in one launch (blockIdx.z): dY (scaled) and [elu](concat(x, x2))
static int pack_pixmajor2(const PackP& a, const PackP& b, hipStream_t st) {
    PackP2 pp;
    pp.t[0] = a; pp.t[1] = b;
    const int C = a.C > b.C ? a.C : b.C;
    hipLaunchKernelGGL(iaf_pack_pixmajor_kernel, dim3((a.P + 63) / 64, (C + 63) / 64, 2), dim3(256), 0, st, pp);
    return (int)hipGetLastError();
}

extern "C" int iaf_conv3x3_backward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                    const float* const* dys, const int* dy_channels, int n_dys, float dy_scale,
                                    float* const* dxs, const int* dx_channels, int n_dxs, const float* dx_residual,
                                    const float* V, const float* g, float* dV, float* dg, float* db, int B, int H, int W,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!c || !x || !dys || !dy_channels || !V || !g || !dV || !dg || !db || !workspace) return IAF_ERR_NULL;
    if (!c->training || !c->prepared) return IAF_ERR_NOT_PREPARED;
    if (B <= 0 || H <= 0 || W <= 0 || n_dys < 1 || n_dys > MAXSPLIT || n_dxs < 0 || n_dxs > MAXSPLIT) return IAF_ERR_SHAPE;
    if ((long long)B * H * W > (1LL << 30) / 64) return IAF_ERR_SHAPE;
    if (x2 && (c_split <= 0 || c_split >= c->n_in || (c_split & 3))) return IAF_ERR_SHAPE;
    if (n_dxs && (!dxs || !dx_channels)) return IAF_ERR_NULL;
    if (dx_residual && n_dxs != 1) return IAF_ERR_SHAPE;
    int tot = 0;
    for (int k = 0; k < n_dys; ++k) {
        if (!dys[k]) return IAF_ERR_NULL;
        if (dy_channels[k] <= 0 || (dy_channels[k] & 3)) return IAF_ERR_SHAPE;
        tot += dy_channels[k];
    }
    if (tot != c->n_out) return IAF_ERR_SHAPE;
    int dends[MAXSPLIT];
    tot = 0;
    for (int k = 0; k < n_dxs; ++k) {
        if (!dxs[k]) return IAF_ERR_NULL;
        if (dx_channels[k] <= 0 || (dx_channels[k] & 3)) return IAF_ERR_SHAPE;
        tot += dx_channels[k];
        dends[k] = tot;
    }
    if (n_dxs && tot != c->n_in) return IAF_ERR_SHAPE;
    if (((uintptr_t)workspace & 15) != 0 || workspace_bytes < iaf_conv3x3_train_workspace_bytes(c, B, H, W)) return IAF_ERR_WORKSPACE;
    const int P = B * H * W, HW = H * W;
    ConvTrainWs tw;
    conv3x3_train_ws_floats(c, P, &tw, (float*)workspace);
    hipStream_t st = (hipStream_t)stream;
    GemmLayer& L = c->L;
    int rc;
    if (c->generic) {
        // channel counts outside the MFMA path: direct loops over the NCHW tensors as they are (iaf_kernels_generic.hpp)
        if (c->deconv) return IAF_ERR_UNSUPPORTED;
        GenGradP p;
        memset(&p, 0, sizeof(p));
        int e = 0;
        for (int k = 0; k < n_dys; ++k) { e += dy_channels[k]; p.dy[k] = dys[k]; p.dy_end[k] = e; }
        p.ndy = n_dys; p.dy_scale = dy_scale;
        p.x = x; p.x2 = x2; p.c_split = c_split; p.in_elu = elu_input ? 1 : 0;
        p.w = L.wp; p.ntaps = MAXTAPS; p.B = B; p.H = H; p.W = W; p.cin = L.cin; p.cout = L.cout;
        for (int k = 0; k < n_dxs; ++k) { p.dx[k] = dxs[k]; p.dx_end[k] = dends[k]; }
        p.ndx = n_dxs; p.res = dx_residual;
        p.dW = tw.dW; p.db = tw.dbp;
        if (n_dxs) hipLaunchKernelGGL(iaf_generic_dgrad_kernel, ew_grid((size_t)P * L.cin), dim3(256), 0, st, p);
        hipLaunchKernelGGL(iaf_generic_wgrad_kernel, dim3((unsigned)((size_t)MAXTAPS * L.cin * L.cout + L.cout)), dim3(256), 0, st, p);
        GenWnBwdP wn;
        memset(&wn, 0, sizeof(wn));
        wn.V[0] = V; wn.g[0] = g; wn.dV[0] = dV; wn.dg[0] = dg; wn.db[0] = db;
        wn.dW = tw.dW; wn.dbsum = tw.dbp; wn.cin = L.cin; wn.cout_each = L.cout; wn.npair = 1; wn.ntaps = MAXTAPS;
        hipLaunchKernelGGL(iaf_generic_wn_bwd_kernel, dim3(L.cout), dim3(256), 0, st, wn);
        c->pending = false;
        return (int)hipGetLastError();
    }
    // (1) operands, pixel-major.  The column sums of dY (db's partials, one row per 64 pixels) come out of the same launch while they
    // fit the 256 rows the buffers hold (P <= 16384); beyond that the reduce launch below sums them over larger slabs, as before.
    const bool fold_db = (P + 63) / 64 <= 256;
    {
        const float* xs[2] = {x, x2};
        const int xc[2] = {x2 ? c_split : c->n_in, c->n_in - c_split};
        PackP pa, pb;
        pack_fill(pa, dys, dy_channels, n_dys, tw.dyc, c->n_out, HW, P, dy_scale, 0);
        if (fold_db) pa.colsum = c->defer_wn ? c->own_dbp : tw.dbp;      // the bias gradient's partials, while the tile is in LDS
        pack_fill(pb, xs, xc, x2 ? 2 : 1, tw.xe, c->n_in, HW, P, 1.0f, elu_input ? 1 : 0);
        if ((rc = pack_pixmajor2(pa, pb, st))) return rc;
    }
    // (2) data gradient
    if (n_dxs) {
        ConvP p;
        memset(&p, 0, sizeof(p));
        p.B = B; p.H = H; p.W = W; p.HW = HW; p.P = P;
        p.x = tw.dyc; p.mode = MODE_DGRAD_PLAIN;
        p.zin = elu_input ? tw.xe : nullptr;
        p.res = dx_residual;
        p.nsplit = n_dxs;
        for (int k = 0; k < MAXSPLIT; ++k) {
            p.split_end[k] = dends[k < n_dxs ? k : n_dxs - 1];
            p.split_ptr[k] = dxs[k < n_dxs ? k : n_dxs - 1];
        }
        // bf16x3 unless the conv's precision is fp32 or a backward search measured the fp32 kernel faster at this size
        const int choice3 = (!conv_split(c) || c->bf3_choice == 3 || !c->T.wp3) ? 3 : (c->T.tuned_P == (long long)P && c->T.tuned_W == W && !c->T.tuned_bf3) ? 3 : 1;
        const bool dgrad16 = c->precision == IAF_PRECISION_F16X2 && !c->f16_off && c->T.wp2 != nullptr && !c->deconv;
        if ((rc = conv3x3_launch(c->T, p, EPI_DGRAD9, IN_PIXMAJOR, false, true, st, IAF_VARIANT_TF, choice3, nullptr, dgrad16))) return rc;
    }
    // (3) weight gradient: partials over pixel ranges, then reduce (+ column sums of dY for db)
    const unsigned short* tapmask = nullptr;
    if ((rc = tapmask_for(B, H, W, st, tw.tapmask, &tapmask))) return rc;
    if ((rc = launch_wgrad(nullptr, L, tw.xe, tw.dyc, tw.part, tapmask, B, H, W, st, 1, conv_split(c)))) return rc;
    const int nslab = fold_db ? (P + 63) / 64 : 256;
    const int px_per_slab = (P + nslab - 1) / nslab;
    float* dWbuf = c->defer_wn ? c->own_dW : tw.dW;
    float* dbpbuf = c->defer_wn ? c->own_dbp : tw.dbp;
    {
        const size_t n4 = (size_t)MAXTAPS * L.cin * L.cout / 4;
        int nblk = (int)((n4 + 255) / 256);
        if (nblk > 1024) nblk = 1024;
        hipLaunchKernelGGL(iaf_wgrad_reduce_kernel, dim3(nblk + (fold_db ? 0 : nslab)), dim3(256), 0, st, tw.part, dWbuf,
                           wgrad_nrange(P, L.cin, MAXTAPS, L.cout, conv_split(c)), n4, nblk, (const float*)tw.dyc, dbpbuf, P, L.cout, px_per_slab);
    }
    // (4) through the weight norm -- now, or in iaf_conv3x3_wn_bwd_batch_run with every other conv of the model
    if (c->deconv) {      // deconv2d's norm runs per INPUT channel over the rotated filter (layers.py:104): its own two launches
        float* inv_norm = L.bias + (size_t)L.ncot * 16;       // (scratch behind the packed bias, as in prepare_deconv)
        float* S = tw.part;                                    // the range partials are reduced by now
        hipLaunchKernelGGL(iaf_deconv_bwd_channel_kernel, dim3(L.cin), dim3(256), 0, st, V, g, (const float*)dWbuf, inv_norm, S, L.cin,
                           L.cout);
        hipLaunchKernelGGL(iaf_deconv_bwd_apply_kernel, dim3(L.cout), dim3(256), 0, st, V, g, (const float*)dWbuf,
                           (const float*)inv_norm, (const float*)S, (const float*)dbpbuf, nslab, dV, dg, db, L.cin, L.cout);
        c->pending = false;
        return (int)hipGetLastError();
    }
    if (c->defer_wn) {
        c->pend_nslab = nslab; c->pending = true;
        return (int)hipGetLastError();
    }
    WnBwdLayer w;
    memset(&w, 0, sizeof(w));
    w.V = V; w.g = g; w.dW = dWbuf; w.dbp = dbpbuf; w.dV = dV; w.dg = dg; w.db = db;
    w.cin = L.cin; w.cout = L.cout; w.cout_packed = L.cout; w.nslab = nslab; w.pack_stride = 1;
    hipLaunchKernelGGL(iaf_wn_bwd_plain_kernel, dim3(L.cout / 16), dim3(256), 0, st, w);
    return (int)hipGetLastError();
}

// Launch-shape search for the data gradient (the transposed problem has its own best shape): times whole
// iaf_conv3x3_backward calls -- everything but the dgrad launch is the same for every candidate -- and pins the fastest.
extern "C" int iaf_conv3x3_autotune_backward(iaf_conv3x3_t* c, const float* x, const float* x2, int c_split, int elu_input,
                                             const float* const* dys, const int* dy_channels, int n_dys, float dy_scale,
                                             float* const* dxs, const int* dx_channels, int n_dxs, const float* dx_residual,
                                             const float* V, const float* g, float* dV, float* dg, float* db, int B, int H,
                                             int W, void* workspace, size_t workspace_bytes, int reps, void* stream,
                                             int* best_shape, float* best_us) {
    if (!c) return IAF_ERR_NULL;
    if (!c->training) return IAF_ERR_NOT_PREPARED;
    if (reps <= 0) reps = 10;
    hipStream_t st = (hipStream_t)stream;
    GemmLayer& T = c->T;
    if (n_dxs == 0) return IAF_OK;          // no data gradient requested: nothing to tune
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const bool was_pending = c->pending;
    float best = 1e30f;
    int bsh[4] = {T.nt, T.pxt, T.wco, T.ks};
    int rc = IAF_OK;
    auto run = [&]() {
        return iaf_conv3x3_backward(c, x, x2, c_split, elu_input, dys, dy_channels, n_dys, dy_scale, dxs, dx_channels, n_dxs,
                                    dx_residual, V, g, dV, dg, db, B, H, W, workspace, workspace_bytes, stream);
    };
    // (the fp32 candidates run with the bf16x3 data gradient switched off: tuned_bf3 = false for this size)
    T.tuned_P = (long long)B * H * W; T.tuned_W = W; T.tuned_bf3 = false;
    for (int si = 0; si < 8 && rc == IAF_OK; ++si)
        for (int nt = 5; nt >= 1 && rc == IAF_OK; --nt) {
            const int pxt = k_shapes[si][0], wco = k_shapes[si][1], ks = k_shapes[si][2];
            if (T.ncot % (nt * wco) != 0 || T.nchunk < ks) continue;
            GemmLayer t = T;
            t.nt = nt; t.pxt = pxt; t.wco = wco; t.ks = ks;
            if (conv_lds_bytes(t, W) > 160 * 1024) continue;
            if (!pick_kernel(nt, pxt, wco, ks, IN_PIXMAJOR, EPI_DGRAD9)) continue;
            T.nt = nt; T.pxt = pxt; T.wco = wco; T.ks = ks; T.user_tuned = true;
            for (int r = 0; r < 2 && rc == IAF_OK; ++r) rc = run();
            if (rc == IAF_ERR_UNSUPPORTED) { rc = IAF_OK; continue; }
            if (rc) break;
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps && rc == IAF_OK; ++r) rc = run();
            (void)hipEventRecord(e1, st);
            if (rc) break;
            if ((rc = (int)hipEventSynchronize(e1)) != 0) break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; bsh[0] = nt; bsh[1] = pxt; bsh[2] = wco; bsh[3] = ks; }
        }
    // ... and the bf16x3 data gradient in its rule shape (conv3x3_bf3_shape), if this conv has the transposed bf16x3 pack
    bool bf3_wins = false;
    if (rc == IAF_OK && T.wp3 && c->bf3_choice != 3 && conv_split(c)) {
        T.user_tuned = false;
        T.tuned_bf3 = true;
        for (int r = 0; r < 2 && rc == IAF_OK; ++r) rc = run();
        if (rc == IAF_OK) {
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < reps && rc == IAF_OK; ++r) rc = run();
            (void)hipEventRecord(e1, st);
            float ms = 0.f;
            if (rc == IAF_OK && (rc = (int)hipEventSynchronize(e1)) == 0) {
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; bf3_wins = true; }
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    T.tuned_bf3 = bf3_wins;
    if (bf3_wins) { bsh[0] = -T.b_nt; bsh[1] = T.b_ppw; bsh[2] = T.b_wco; bsh[3] = T.b_ks; }       // reported like the forward's bf16x3 shapes
    // remember the winner for THIS problem size only (t_* hold (nt, pxt, wco, ks) here): a backward at another size goes
    // back to the automatic shape instead of inheriting a shape tuned -- and LDS-sized -- for this one
    T.user_tuned = false;
    T.tuned_P = (long long)B * H * W; T.tuned_W = W;
    if (!bf3_wins) { T.t_nt = bsh[0]; T.t_ppw = bsh[1]; T.t_pxt = bsh[2]; T.t_ks = bsh[3]; }
    if (rc == IAF_OK) rc = run();           // leave the outputs as computed with the chosen shape
    (void)was_pending;
    if (best_shape) for (int i = 0; i < 4; ++i) best_shape[i] = bsh[i];
    if (best_us) *best_us = best * 1e3f / reps;
    return rc;
}

// ---- deferred weight-norm backward of many plain convs in one launch ----------------------------------------------
extern "C" int iaf_conv3x3_set_defer_weightnorm(iaf_conv3x3_t* c, int on) {
    if (!c) return IAF_ERR_NULL;
    if (on && !c->training) return IAF_ERR_NOT_PREPARED;
    if (on) {
        if (!c->own_dW) HIP_TRY(hipMalloc(&c->own_dW, (size_t)MAXTAPS * c->n_in * c->n_out * sizeof(float)));
        if (!c->own_dbp) HIP_TRY(hipMalloc(&c->own_dbp, (size_t)256 * c->n_out * sizeof(float)));
    }
    c->defer_wn = on != 0;
    c->pending = false;
    return IAF_OK;
}

struct iaf_conv3x3_wn_bwd_batch {
    int n, ntiles;
    iaf_conv3x3** convs;
    WnBwdLayer* h_layers;   // current descriptor table (host)
    DescTable tab;          // its way to the device (see DescTable in iaf_engine.hip)
    int* d_tile2layer;
    int* d_tile_begin;
};

extern "C" int iaf_conv3x3_wn_bwd_batch_destroy(iaf_conv3x3_wn_bwd_batch_t* b) {
    if (!b) return IAF_ERR_NULL;
    free(b->h_layers);
    desc_destroy(&b->tab);
    if (b->d_tile2layer) (void)hipFree(b->d_tile2layer);
    if (b->d_tile_begin) (void)hipFree(b->d_tile_begin);
    free(b->convs);
    delete b;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_wn_bwd_batch_create(iaf_conv3x3_wn_bwd_batch_t** out, iaf_conv3x3_t* const* convs, int n) {
    if (!out || !convs) return IAF_ERR_NULL;
    *out = nullptr;
    if (n <= 0) return IAF_ERR_SHAPE;
    iaf_conv3x3_wn_bwd_batch* b = new (std::nothrow) iaf_conv3x3_wn_bwd_batch();
    if (!b) return (int)hipErrorOutOfMemory;
    memset(b, 0, sizeof(*b));
    b->n = n;
    b->convs = (iaf_conv3x3**)calloc(n, sizeof(iaf_conv3x3*));
    int nt = 0;
    for (int i = 0; i < n; ++i) {
        if (!convs[i]) { iaf_conv3x3_wn_bwd_batch_destroy(b); return IAF_ERR_NULL; }
        if (convs[i]->generic || convs[i]->mask_mode) { iaf_conv3x3_wn_bwd_batch_destroy(b); return IAF_ERR_UNSUPPORTED; }
        b->convs[i] = convs[i];
        nt += convs[i]->L.ncot;
    }
    b->ntiles = nt;
    int* t2l = (int*)malloc(sizeof(int) * nt);
    int* tb = (int*)malloc(sizeof(int) * (n + 1));
    int rc;
    b->h_layers = (WnBwdLayer*)calloc(n, sizeof(WnBwdLayer));
    if (!b->h_layers) { free(t2l); free(tb); iaf_conv3x3_wn_bwd_batch_destroy(b); return (int)hipErrorOutOfMemory; }
    if ((rc = desc_init(&b->tab, sizeof(WnBwdLayer) * n)) != 0 ||
        (rc = (int)hipMalloc((void**)&b->d_tile2layer, sizeof(int) * nt)) != 0 ||
        (rc = (int)hipMalloc((void**)&b->d_tile_begin, sizeof(int) * (n + 1))) != 0) {
        free(t2l); free(tb); iaf_conv3x3_wn_bwd_batch_destroy(b); return rc;
    }
    int tile = 0;
    for (int i = 0; i < n; ++i) {
        const GemmLayer& L = convs[i]->L;
        WnBwdLayer& w = b->h_layers[i];
        w.cin = L.cin; w.cout = L.cout; w.cout_packed = L.cout; w.pack_stride = 1;
        tb[i] = tile;
        for (int t = 0; t < L.ncot; ++t) t2l[tile++] = i;
    }
    tb[n] = tile;
    rc = (int)hipMemcpy(b->d_tile2layer, t2l, sizeof(int) * nt, hipMemcpyHostToDevice);
    if (!rc) rc = (int)hipMemcpy(b->d_tile_begin, tb, sizeof(int) * (n + 1), hipMemcpyHostToDevice);
    free(t2l); free(tb);
    if (rc) { iaf_conv3x3_wn_bwd_batch_destroy(b); return rc; }
    *out = b;
    return IAF_OK;
}

extern "C" int iaf_conv3x3_wn_bwd_batch_run(iaf_conv3x3_wn_bwd_batch_t* b, const float* const* V, const float* const* g,
                                            float* const* dV, float* const* dg, float* const* db, void* stream) {
    if (!b || !V || !g || !dV || !dg || !db) return IAF_ERR_NULL;
    bool changed = false;
    for (int i = 0; i < b->n; ++i) {
        iaf_conv3x3* c = b->convs[i];
        if (!c->defer_wn || (!c->pending && !c->deconv)) return IAF_ERR_NOT_PREPARED;
        if (!V[i] || !g[i] || !dV[i] || !dg[i] || !db[i]) return IAF_ERR_NULL;
        WnBwdLayer& w = b->h_layers[i];
        changed |= (w.skip != (c->deconv ? 1 : 0));
        w.skip = c->deconv ? 1 : 0;        // a deconv2d pushes its gradient through its own norm inside iaf_conv3x3_backward
        changed |= (w.V != V[i]) | (w.g != g[i]) | (w.dV != dV[i]) | (w.dg != dg[i]) | (w.db != db[i]) |
                   (w.dW != c->own_dW) | (w.dbp != c->own_dbp) | (w.nslab != c->pend_nslab);
        w.V = V[i]; w.g = g[i]; w.dV = dV[i]; w.dg = dg[i]; w.db = db[i];
        w.dW = c->own_dW; w.dbp = c->own_dbp; w.nslab = c->pend_nslab;
    }
    hipStream_t st = (hipStream_t)stream;
    const void* d_layers = nullptr;
    { int rc = desc_upload(&b->tab, b->h_layers, changed, st, &d_layers); if (rc) return rc; }
    hipLaunchKernelGGL(iaf_wn_bwd_plain_batch_kernel, dim3(b->ntiles), dim3(256), 0, st, (const WnBwdLayer*)d_layers, b->d_tile2layer,
                       b->d_tile_begin);
    return (int)hipGetLastError();
}
