// iaf_kernels_generic.hpp -- direct-conv fallback kernels for channel counts outside the MFMA path.
// Part of the single translation unit iaf_engine.hip (included there, in order; not a standalone header).
#pragma once

// ---------------------------------------------------------------------------------------------
// GENERIC FALLBACK: direct (VALU) masked conv for channel counts the MFMA path does not cover (not multiples of 16,
// or > 256).  Same arithmetic, same fused epilogues, NCHW everywhere, one thread per output element.  Orders of
// magnitude slower than the MFMA path -- it exists so that every shape the reference accepts (layers.py:116 only asks
// that n_h and n_z divide each other) runs through the same C ABI; tests hold it to the reference's tiny golden cases.
// ---------------------------------------------------------------------------------------------
struct GenPrepLayer {
    const float* V[2]; const float* g[2]; const float* b[2];
    float* w;        // effective weights [NTAPS][cin][cout_total]
    float* bias;     // [cout_total]
    int cin, cout_each, npair, zerodiag, ch_begin;
    int ntaps;       // 5 = MADE-masked (default when 0), 9 = all nine filter positions stored
    int mask9;       // ntaps == 9 only: 0 = unmasked conv2d, 1 = ar_conv2d mask (dead taps stored as zeros)
};
struct GenPrepArgs { GenPrepLayer L[MAX_GEMM_LAYERS]; int nlayers; };

__global__ __launch_bounds__(256) void iaf_generic_prep_kernel(GenPrepArgs a) {
    __shared__ float red[256];
    int li = 0;
    for (int i = 1; i < a.nlayers; ++i)
        if ((int)blockIdx.x >= a.L[i].ch_begin) li = i;
    const GenPrepLayer& L = a.L[li];
    const int oc = blockIdx.x - L.ch_begin;              // channel inside this GEMM layer: [mean channels | logsd channels]
    const int which = oc / L.cout_each, o = oc - which * L.cout_each;
    const float* V = L.V[which];
    const int n_in = L.cin, n_out = L.cout_each, ctot = L.cout_each * L.npair;
    const bool full = (L.ntaps == MAXTAPS);
    const int ntaps = full ? MAXTAPS : NTAPS;
    float ss = 0.f;
    for (int e = threadIdx.x; e < ntaps * n_in; e += 256) {
        const int t = e / n_in, ci = e - t * n_in;
        const int kh = full ? t / 3 : ((t == 0 || t == 1) ? 1 : 2), kw = full ? t % 3 : ((t == 0) ? 1 : (t == 1 ? 2 : t - 2));
        const bool live = full ? (!L.mask9 || kh == 2 || (kh == 1 && kw == 2) ||
                                  (kh == 1 && kw == 1 && made_live(ci, o, n_in, n_out, L.zerodiag)))   // layers.py:134-141
                               : ((t != 0) || made_live(ci, o, n_in, n_out, L.zerodiag));
        const float v = live ? V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o] : 0.f;
        ss += v * v;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    const float scale = expf(L.g[which][o]) / sqrtf(fmaxf(red[0], 1e-12f));      // layers.py:60
    for (int e = threadIdx.x; e < ntaps * n_in; e += 256) {
        const int t = e / n_in, ci = e - t * n_in;
        const int kh = full ? t / 3 : ((t == 0 || t == 1) ? 1 : 2), kw = full ? t % 3 : ((t == 0) ? 1 : (t == 1 ? 2 : t - 2));
        const bool live = full ? (!L.mask9 || kh == 2 || (kh == 1 && kw == 2) ||
                                  (kh == 1 && kw == 1 && made_live(ci, o, n_in, n_out, L.zerodiag)))   // layers.py:134-141
                               : ((t != 0) || made_live(ci, o, n_in, n_out, L.zerodiag));
        L.w[((size_t)t * n_in + ci) * ctot + oc] = live ? V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o] * scale : 0.f;
    }
    if (threadIdx.x == 0) L.bias[oc] = L.b[which][o];
}

struct GenConvP {
    const float* x;       // NCHW input (NULL in posterior mode for the first layer)
    const float* w; const float* bias;
    const float* ctx; const float* ctx2;
    float* y;             // hidden output, NCHW
    const float* zin; float* out0; float* out1; float* kl_elem;
    const float* qm; const float* ql; const float* rm; const float* rl; const float* pm; const float* pl; const float* eps;
    int B, H, W, cin, cout, nz, mode, is_out, posterior_in;
};

__device__ __forceinline__ float gen_x(const GenConvP& p, int b, int ci, int hh, int ww) {
    const size_t i = (((size_t)b * p.cin + ci) * p.H + hh) * p.W + ww;
    if (!p.posterior_in) return p.x[i];
    return (p.qm[i] + p.rm[i]) + __expf(0.5f * (2.f * (p.ql[i] + p.rl[i]))) * p.eps[i];      // tf_train.py:57,63
}

__global__ __launch_bounds__(256) void iaf_generic_conv_kernel(GenConvP p) {
    const int tap_dh[NTAPS] = {0, 0, 1, 1, 1}, tap_dw[NTAPS] = {0, 1, -1, 0, 1};
    const int nout = p.is_out ? p.nz : p.cout;
    const size_t total = (size_t)p.B * nout * p.H * p.W;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ww = (int)(i % p.W);
        const int hh = (int)((i / p.W) % p.H);
        const int co = (int)((i / ((size_t)p.W * p.H)) % nout);
        const int b = (int)(i / ((size_t)p.W * p.H * nout));
        float a0 = 0.f, a1 = 0.f;
        for (int t = 0; t < NTAPS; ++t) {
            const int h2 = hh + tap_dh[t], w2 = ww + tap_dw[t];
            if (h2 < 0 || h2 >= p.H || w2 < 0 || w2 >= p.W) continue;
            const float* wt = p.w + (size_t)t * p.cin * p.cout;
            for (int ci = 0; ci < p.cin; ++ci) {
                const float xv = gen_x(p, b, ci, h2, w2);
                a0 = fmaf(xv, wt[(size_t)ci * p.cout + co], a0);
                if (p.is_out) a1 = fmaf(xv, wt[(size_t)ci * p.cout + p.nz + co], a1);
            }
        }
        if (!p.is_out) {
            float v = a0 + p.bias[co];
            if (p.ctx) { v += p.ctx[i]; if (p.ctx2) v += p.ctx2[i]; }
            p.y[i] = elu_f(v);
            continue;
        }
        const float m_raw = a0 + p.bias[co], s_raw = a1 + p.bias[p.nz + co];
        if (p.mode == MODE_RAW) { p.out0[i] = m_raw; p.out1[i] = s_raw; continue; }
        const float m = m_raw * 0.1f, sgm = s_raw * 0.1f;
        if (p.mode == MODE_IAF) { p.out0[i] = (p.zin[i] - m) / __expf(sgm); p.out1[i] = sgm; continue; }
        if (p.mode == MODE_INVERSE) { p.out0[i] = p.zin[i] * __expf(sgm) + m; p.out1[i] = sgm; continue; }
        const float mean = p.qm[i] + p.rm[i], logvar = 2.f * (p.ql[i] + p.rl[i]);
        const float z0 = mean + __expf(0.5f * logvar) * p.eps[i];
        const float d0 = z0 - mean;
        float logqs = -0.5f * (1.8378770664093453f + logvar + d0 * d0 / __expf(logvar)) + sgm;
        const float z = (z0 - m) / __expf(sgm);
        const float plv = 2.f * p.pl[i], d1 = z - p.pm[i];
        const float logps = -0.5f * (1.8378770664093453f + plv + d1 * d1 / __expf(plv));
        p.out0[i] = z;
        if (p.out1) p.out1[i] = sgm;
        p.kl_elem[i] = logqs - logps;
    }
}

// plain 3x3 SAME conv, generic channel counts: y = conv(elu?(concat(x, x2)), w) + b  [-> res + 0.1*y], output channels
// scattered to the split tensors.  One thread per output element.
struct GenPlainP {
    const float* x; const float* x2; const float* w; const float* bias; const float* res;
    int B, H, W, cin, cout, c_split, in_elu, nsplit;
    int split_end[MAXSPLIT]; float* split_ptr[MAXSPLIT];
};

__global__ __launch_bounds__(256) void iaf_generic_conv3x3_kernel(GenPlainP p) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * p.cout * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ww = (int)(i % p.W);
        const int hh = (int)((i / p.W) % p.H);
        const int co = (int)((i / HW) % p.cout);
        const int b = (int)(i / (HW * p.cout));
        float acc = 0.f;
        for (int t = 0; t < MAXTAPS; ++t) {
            const int h2 = hh + t / 3 - 1, w2 = ww + t % 3 - 1;
            if (h2 < 0 || h2 >= p.H || w2 < 0 || w2 >= p.W) continue;
            const float* wt = p.w + (size_t)t * p.cin * p.cout;
            for (int ci = 0; ci < p.cin; ++ci) {
                float xv;
                if (p.x2 && ci >= p.c_split) xv = p.x2[(((size_t)b * (p.cin - p.c_split) + (ci - p.c_split)) * p.H + h2) * p.W + w2];
                else xv = p.x[(((size_t)b * (p.x2 ? p.c_split : p.cin) + ci) * p.H + h2) * p.W + w2];
                if (p.in_elu) xv = elu_f(xv);
                acc = fmaf(xv, wt[(size_t)ci * p.cout + co], acc);
            }
        }
        const float v = acc + p.bias[co];
        int k = 0;
        while (k + 1 < p.nsplit && co >= p.split_end[k]) ++k;
        const int c0 = k ? p.split_end[k - 1] : 0;
        const size_t o = (((size_t)b * (p.split_end[k] - c0) + (co - c0)) * p.H + hh) * p.W + ww;
        p.split_ptr[k][o] = p.res ? p.res[i] + 0.1f * v : v;
    }
}

// ---------------------------------------------------------------------------------------------
// GENERIC FALLBACK, BACKWARD (round 5): what TF's autodiff derives for layers.py:52-64 / 158-166 (tf_train.py:138) at channel counts
// outside the MFMA path -- direct loops over NCHW tensors, one thread per output element or one workgroup per reduced element.
// Slow by design (a 72-channel conv: milliseconds); it exists so that the per-op and per-layer training entry points (iaf_step_backward,
// iaf_posterior_block_backward, iaf_conv3x3_backward, a whole IAFLayer) accept the channel counts the reference accepts (layers.py:116;
// checked by tests/test_hip_generic_backward.py against fp64 autograd of the restated forward).  What it does NOT cover: channel
// splits / concat pieces that are not multiples of 4 (iaf_conv3x3_backward returns IAF_ERR_SHAPE for those before it gets here), and the
// BATCHED prep / weight-norm-backward objects (iaf_prep_batch_*, iaf_wn_bwd_batch_*: IAF_ERR_UNSUPPORTED for generic members) that
// CVAE1.prepare_weights / set_grad_buckets always build -- a MODEL with such channel counts trains layer by layer, not through
// CVAE1.forward_backward.
//   dX  = [res +] act'(.) * (W^T dY)      (data gradient: the conv with mirrored taps)
//   dW[t][ci][co] = sum_p a[p + shift(t)][ci] dY[p][co],  db[co] = sum_p dY[p][co]
//   dV, dg through the mask and the weight norm (the arithmetic of wn_bwd_tile, iaf_kernels_backward.hpp)
// ---------------------------------------------------------------------------------------------
struct GenGradP {
    const float* dy[MAXSPLIT]; int dy_end[MAXSPLIT]; int ndy; float dy_scale;   // dY as channel pieces (NCHW each), times dy_scale
    const float* x; const float* x2; int c_split; int in_elu;                  // conv input a = act(concat(x, x2)); in_elu: act = elu
    const float* w;                  // effective weights [ntaps][cin][cout]
    int ntaps;                       // 5: (0,0) (0,1) (1,-1) (1,0) (1,1);  9: kh * 3 + kw, offsets -1 .. 1
    int B, H, W, cin, cout;
    float* dx[MAXSPLIT]; int dx_end[MAXSPLIT]; int ndx;                        // dX as channel pieces (0: no data gradient wanted)
    const float* res;                // optional [B][cin][H][W], added (plain conv: the residual path's gradient)
    const float* act_out;            // masked stack, hidden layers: h = elu(a) as saved by the forward; elu'(a) = h > 0 ? 1 : h + 1
    const float* dzn; const float* logsd;   // masked stack, first layer: dz = W^T dY + dzn exp(-logsd)   (tf_train.py:71)
    float* dx_copy;                  // optional second copy of dX (d context = d a_0)
    float* dW; float* db;            // [ntaps][cin][cout], [cout]
};

__device__ __forceinline__ void gen_tap(int ntaps, int t, int& dh, int& dw) {
    if (ntaps == MAXTAPS) { dh = t / 3 - 1; dw = t % 3 - 1; }
    else { dh = (t < 2) ? 0 : 1; dw = (t == 0) ? 0 : (t == 1) ? 1 : t - 3; }
}
__device__ __forceinline__ float gen_dy(const GenGradP& p, int b, int co, int hh, int ww) {
    int k = 0;
    while (k + 1 < p.ndy && co >= p.dy_end[k]) ++k;
    const int c0 = k ? p.dy_end[k - 1] : 0;
    return p.dy_scale * p.dy[k][(((size_t)b * (p.dy_end[k] - c0) + (co - c0)) * p.H + hh) * p.W + ww];
}
__device__ __forceinline__ float gen_in_raw(const GenGradP& p, int b, int ci, int hh, int ww) {
    if (p.x2 && ci >= p.c_split) return p.x2[(((size_t)b * (p.cin - p.c_split) + (ci - p.c_split)) * p.H + hh) * p.W + ww];
    return p.x[(((size_t)b * (p.x2 ? p.c_split : p.cin) + ci) * p.H + hh) * p.W + ww];
}

__global__ __launch_bounds__(256) void iaf_generic_dgrad_kernel(GenGradP p) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t total = (size_t)p.B * p.cin * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ww = (int)(i % p.W);
        const int hh = (int)((i / p.W) % p.H);
        const int ci = (int)((i / HW) % p.cin);
        const int b = (int)(i / (HW * p.cin));
        float acc = 0.f;
        for (int t = 0; t < p.ntaps; ++t) {
            int dh, dw;
            gen_tap(p.ntaps, t, dh, dw);
            const int h2 = hh - dh, w2 = ww - dw;              // the output pixel whose tap t read this input pixel
            if (h2 < 0 || h2 >= p.H || w2 < 0 || w2 >= p.W) continue;
            const float* wt = p.w + ((size_t)t * p.cin + ci) * p.cout;
            for (int co = 0; co < p.cout; ++co) acc = fmaf(gen_dy(p, b, co, h2, w2), wt[co], acc);
        }
        if (p.in_elu) { const float xr = gen_in_raw(p, b, ci, hh, ww); acc *= (xr > 0.f) ? 1.f : __expf(xr); }
        if (p.act_out) { const float h = p.act_out[i]; acc *= (h > 0.f) ? 1.f : h + 1.f; }
        if (p.dzn) acc += p.dzn[i] * __expf(-p.logsd[i]);
        if (p.res) acc += p.res[i];
        if (p.dx_copy) p.dx_copy[i] = acc;
        int k = 0;
        while (k + 1 < p.ndx && ci >= p.dx_end[k]) ++k;
        const int c0 = k ? p.dx_end[k - 1] : 0;
        p.dx[k][(((size_t)b * (p.dx_end[k] - c0) + (ci - c0)) * p.H + hh) * p.W + ww] = acc;
    }
}

// one workgroup per (tap, ci, co) [+ cout workgroups behind them for db]: the sum over all pixels, in a fixed order
__global__ __launch_bounds__(256) void iaf_generic_wgrad_kernel(GenGradP p) {
    __shared__ float red[256];
    const size_t nW = (size_t)p.ntaps * p.cin * p.cout;
    const size_t id = blockIdx.x;
    const int P = p.B * p.H * p.W, HW = p.H * p.W;
    float a = 0.f;
    if (id < nW) {
        const int co = (int)(id % p.cout), ci = (int)((id / p.cout) % p.cin), t = (int)(id / ((size_t)p.cout * p.cin));
        int dh, dw;
        gen_tap(p.ntaps, t, dh, dw);
        for (int px = threadIdx.x; px < P; px += 256) {
            const int b = px / HW, pp = px - b * HW, hh = pp / p.W, ww = pp - hh * p.W;
            const int h2 = hh + dh, w2 = ww + dw;
            if (h2 < 0 || h2 >= p.H || w2 < 0 || w2 >= p.W) continue;
            float xv = gen_in_raw(p, b, ci, h2, w2);
            if (p.in_elu) xv = elu_f(xv);
            a = fmaf(xv, gen_dy(p, b, co, hh, ww), a);
        }
    } else {
        const int co = (int)(id - nW);
        for (int px = threadIdx.x; px < P; px += 256) {
            const int b = px / HW, pp = px - b * HW, hh = pp / p.W, ww = pp - hh * p.W;
            a += gen_dy(p, b, co, hh, ww);
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (id < nW) p.dW[id] = red[0]; else p.db[id - nW] = red[0];
    }
}

// mask + weight norm backward of the convs behind one GEMM layer (the output pair: two convs, [mean channels | logsd channels]):
// one workgroup per packed output channel, the same liveness rule as iaf_generic_prep_kernel
struct GenWnBwdP {
    const float* V[2]; const float* g[2]; float* dV[2]; float* dg[2]; float* db[2];
    const float* dW;     // [ntaps][cin][cout_total]
    const float* dbsum;  // [cout_total]
    int cin, cout_each, npair, zerodiag, ntaps, mask9;
};
__global__ __launch_bounds__(256) void iaf_generic_wn_bwd_kernel(GenWnBwdP L) {
    __shared__ float red[2][256];
    const int oc = blockIdx.x, which = oc / L.cout_each, o = oc - which * L.cout_each;
    const float* V = L.V[which];
    const int n_in = L.cin, n_out = L.cout_each, ctot = L.cout_each * L.npair;
    const bool full = (L.ntaps == MAXTAPS);
    const int ntaps = full ? MAXTAPS : NTAPS;
    auto live_at = [&](int t, int ci, int& kh, int& kw) -> bool {
        kh = full ? t / 3 : ((t == 0 || t == 1) ? 1 : 2);
        kw = full ? t % 3 : ((t == 0) ? 1 : (t == 1 ? 2 : t - 2));
        return full ? (!L.mask9 || kh == 2 || (kh == 1 && kw == 2) || (kh == 1 && kw == 1 && made_live(ci, o, n_in, n_out, L.zerodiag)))
                    : ((t != 0) || made_live(ci, o, n_in, n_out, L.zerodiag));
    };
    float ss = 0.f, dot = 0.f;
    for (int e = threadIdx.x; e < ntaps * n_in; e += 256) {
        const int t = e / n_in, ci = e - t * n_in;
        int kh, kw;
        if (!live_at(t, ci, kh, kw)) continue;
        const float v = V[((size_t)(kh * 3 + kw) * n_in + ci) * n_out + o];
        ss += v * v;
        dot += v * L.dW[((size_t)t * n_in + ci) * ctot + oc];
    }
    red[0][threadIdx.x] = ss; red[1][threadIdx.x] = dot;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) { red[0][threadIdx.x] += red[0][threadIdx.x + k]; red[1][threadIdx.x] += red[1][threadIdx.x + k]; }
        __syncthreads();
    }
    const float n = sqrtf(fmaxf(red[0][0], 1e-12f)), e = expf(L.g[which][o]), du = red[1][0] / n;
    if (threadIdx.x == 0) { L.dg[which][o] = e * du; L.db[which][o] = L.dbsum[oc]; }
    // dV over all nine filter positions: dead taps and masked entries are exact zeros
    for (int e9 = threadIdx.x; e9 < 9 * n_in; e9 += 256) {
        const int k9 = e9 / n_in, ci = e9 - k9 * n_in, kh9 = k9 / 3, kw9 = k9 % 3;
        const int t = full ? k9 : ((kh9 == 1 && kw9 == 1) ? 0 : (kh9 == 1 && kw9 == 2) ? 1 : (kh9 == 2) ? 2 + kw9 : -1);
        float outv = 0.f;
        int kh, kw;
        if (t >= 0 && live_at(t, ci, kh, kw)) {
            const float v = V[((size_t)k9 * n_in + ci) * n_out + o];
            outv = (e / n) * (L.dW[((size_t)t * n_in + ci) * ctot + oc] - (v / n) * du);
        }
        L.dV[which][((size_t)k9 * n_in + ci) * n_out + o] = outv;
    }
}

// affine + log-det backward of the IAF step (tf_train.py:70-72), NCHW -> dY of the output pair [B][mean channels | logsd channels][H][W]
__global__ __launch_bounds__(256) void iaf_generic_bwd_affine_kernel(const float* __restrict__ z_new, const float* __restrict__ logsd,
                                                                    const float* __restrict__ dzn, const float* __restrict__ dls,
                                                                    float* __restrict__ dy, int n_z, int HW, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = i / HW, pp = i - bc * HW;
        const size_t b = bc / n_z, c = bc - b * n_z;
        const float g = dzn[i], ls = logsd[i];
        dy[((b * 2 * n_z) + c) * HW + pp] = -0.1f * g * __expf(-ls);
        dy[((b * 2 * n_z) + n_z + c) * HW + pp] = 0.1f * (dls[i] - g * z_new[i]);
    }
}
