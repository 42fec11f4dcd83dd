// iaf_kernels_misc.hpp -- small forward kernels: KL / free-bits reductions, Gaussian sample/logps, max-diff, Adamax+EMA, streaming lower bound, data-dependent init, discretized logistic.
// Part of the single translation unit iaf_engine.hip (included there, in order; not a standalone header).
#pragma once

// ---------------------------------------------------------------------------------------------
// KL / free-bits reduction, tf_train.py:77-85.  kl_elem [B,Z,H,W] -> kl_cost[B], kl_obj[B].
// One workgroup: deterministic tree order.  S[b,c] = sum_{H,W} kl.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iaf_kl_rowsum_kernel(const float* kl, float* S, int rows, int HW) {
    // one wave per (b,c) row
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float a = 0.f;
    for (int i = lane; i < HW; i += 64) a += kl[(size_t)row * HW + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
    if (lane == 0) S[row] = a;
}

// gate (optional, [Z]): 1 where the free-bits max() passes the gradient (mean_b S[b,c] > kl_min), else 0 -- what the
// backward of tf_train.py:79-80 needs; written here because the batch mean is already on hand.
// nrb > 0: S holds per-row-block partial sums [B][nrb][Z] (the one-launch step's StepP::kl_part) and the sum over the row
// blocks -- in row order, eight loads in flight -- is the first thing this launch does; nrb = 0: S is [B][Z].
// scratch ([B*Z] floats) is only touched when B*Z does not fit the LDS staging.
__global__ __launch_bounds__(256) void iaf_kl_finish_kernel(const float* S, float* kl_obj, float* kl_cost, int B, int Z,
                                                           float kl_min, float* gate, int nrb, float* scratch) {
    // S is tiny ([B, Z]); stage it through LDS in one coalesced sweep instead of B*Z dependent global loads
    __shared__ float sh[8192];
    __shared__ float part[256];
    __shared__ float s_fb;
    const int tid = threadIdx.x, n = B * Z;
    const bool in_lds = n <= 8192;
    if (nrb > 0) {
        // item = four consecutive channels of one image: its nrb partial sums are nrb independent 16-byte loads, eight in
        // flight per round (the whole table in ONE round trip at B = 32, 16x16: 256 items x 8 row blocks)
        float* dst = in_lds ? sh : scratch;
        typedef float kf4 __attribute__((ext_vector_type(4)));
        if ((Z & 3) == 0) {
            const int Z4 = Z >> 2;
            for (int i = tid; i < B * Z4; i += 256) {
                const int b = i / Z4, c4 = i - b * Z4;
                kf4 a = kf4{0.f, 0.f, 0.f, 0.f};
                for (int r0 = 0; r0 < nrb; r0 += 8) {
                    kf4 v8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = r0 + k < nrb ? r0 + k : nrb - 1;
                        v8[k] = *(const kf4*)(S + ((size_t)b * nrb + r) * Z + 4 * c4);
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (r0 + k < nrb) a += v8[k];
                }
                *(kf4*)(dst + (size_t)b * Z + 4 * c4) = a;
            }
        } else {
            for (int i = tid; i < n; i += 256) {
                const int b = i / Z, c = i - b * Z;
                float a = 0.f;
                for (int r = 0; r < nrb; ++r) a += S[((size_t)b * nrb + r) * Z + c];
                dst[i] = a;
            }
        }
        __syncthreads();
    } else if (in_lds) {
        for (int i = tid; i < n; i += 256) sh[i] = S[i];
        __syncthreads();
    }
    const float* src = in_lds ? sh : (nrb > 0 ? scratch : S);
    if (kl_min > 0.f) {
        // kl_ave[c] = max(mean_b S[b,c], kl_min); kl_obj[b] = sum_c kl_ave[c]   (tf_train.py:79-82)
        float a = 0.f;
        for (int c = tid; c < Z; c += 256) {
            float m = 0.f;
            for (int b = 0; b < B; ++b) m += src[(size_t)b * Z + c];
            a += fmaxf(m / (float)B, kl_min);
            if (gate) gate[c] = (m / (float)B > kl_min) ? 1.f : 0.f;
        }
        part[tid] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) part[tid] += part[tid + o];
            __syncthreads();
        }
        if (tid == 0) s_fb = part[0];
        __syncthreads();
    }
    for (int b = tid; b < B; b += 256) {
        float a = 0.f;
        for (int c = 0; c < Z; ++c) a += src[(size_t)b * Z + c];
        kl_cost[b] = a;                                        // tf_train.py:85
        kl_obj[b] = (kl_min > 0.f) ? s_fb : a;                 // tf_train.py:82 / 84
    }
}

// S[b][c] = sum over the row blocks of the partial sums [B][nrb][Z], in row order: the first phase of the kernel above as
// its own many-workgroup launch, for batches whose partial sums are too many for ONE workgroup to walk (config 5: B = 256)
__global__ __launch_bounds__(256) void iaf_kl_partsum_kernel(const float* part, float* S, int n, int Z, int nrb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = i / Z, c = i - b * Z;
    float a = 0.f;
    for (int r0 = 0; r0 < nrb; r0 += 8) {
        float v8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = r0 + k < nrb ? r0 + k : nrb - 1;
            v8[k] = part[((size_t)b * nrb + r) * Z + c];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) a += (r0 + k < nrb) ? v8[k] : 0.f;
    }
    S[i] = a;
}

// ---------------------------------------------------------------------------------------------
// elementwise distributions (tf_utils/distributions.py)
// ---------------------------------------------------------------------------------------------
// max |a - b| over n elements -> *out (float bits; non-negative floats order like unsigned ints).  NaN counts as +inf.
__global__ __launch_bounds__(256) void iaf_maxdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                         unsigned* out) {
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d > m || d != d) ? (d != d ? __builtin_inff() : d) : m;
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// ---- iaf_step_inverse without the host in its loop (round 6) -------------------------------------------------------------------
// The inverse of the IAF step -- z0 with z = (z0 - m(z0)) / exp(s(z0)) (tf_train.py:69-72 read backwards; the reference never inverts:
// it only evaluates densities of its own samples, tf_train.py:60-66, models.py:330-359) -- as JACOBI sweeps z0 <- z exp(s(z0)) + m(z0)
// on the forward kernels.  Why not the anti-diagonal wavefront north_star names: m, s at (pixel p, channel c) depend on z0 at the pixels
// right of / below p and on the lower channels at p (the MADE order), a DAG of depth H W n_z -- a true scan is H W n_z dependent steps of
// one channel of one pixel each (8192 launches-worth of latency at 16x16), while a Jacobi sweep is one full-rate forward launch that moves
// EVERY element one step down the DAG at once: exact after at most H W n_z sweeps (the DAG's depth), and -- the 0.1 on m and s makes the
// map a contraction in practice -- at fp32 resolution after ~8.  The control words live in device memory:
struct InvCtl { unsigned res_bits, done, sweeps, arrivals; float last_res; unsigned pad[3]; };
// max |a - b| of one sweep pair; the LAST workgroup to arrive compares it with tol, records (sweeps so far, residual), raises `done` (which the
// queued sweep launches read: StepP::skip) and re-arms the two scratch words.  Does nothing once done.  NaN counts as +inf (never converged).
__global__ __launch_bounds__(256) void iaf_inverse_check_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, InvCtl* c,
                                                               float tol, unsigned sweeps_so_far) {
    if (*(volatile unsigned*)&c->done) return;
    float m = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d > m || d != d) ? (d != d ? __builtin_inff() : d) : m;
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        __hip_atomic_fetch_max(&c->res_bits, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&c->arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            const float res = __uint_as_float(__hip_atomic_load(&c->res_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            c->last_res = res;
            c->sweeps = sweeps_so_far;
            c->res_bits = 0u; c->arrivals = 0u;
            if (res <= tol) __hip_atomic_store(&c->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// the result into z0: where the sweep launches return early once converged (skipping), the converged estimate sits in the buffer sweep
// number c->sweeps wrote, buf[(max_sweeps - sweeps) & 1] with buf[0] = z0; otherwise the last sweep has written z0 itself.  Also the two
// numbers the caller may ask for: out[0] = sweeps run (as an int), out[1] = the last residual (as a float; -1: never checked)
__global__ __launch_bounds__(256) void iaf_inverse_finish_kernel(const float* __restrict__ other, float* __restrict__ z0, size_t n, const InvCtl* c,
                                                                int max_sweeps, int skipping, int checked, unsigned* out) {
    const unsigned done = c->done, sw = c->sweeps;
    if (blockIdx.x == 0 && threadIdx.x == 0 && out) {
        out[0] = done ? sw : (unsigned)max_sweeps;
        out[1] = __float_as_uint(checked ? c->last_res : -1.f);
    }
    if (!(skipping && done && ((max_sweeps - (int)sw) & 1))) return;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) z0[i] = other[i];
}

// lvs: 1 when the second tensor holds log-variances (distributions.py), 2 when it holds log standard deviations (the callers'
// `2 * logsd`: tf_train.py:56-57, rand.py:81-86 -- exact in fp32, so the bits are those of a separate doubling)
__global__ void iaf_gauss_sample_kernel(const float* mean, const float* logvar, const float* noise, float* out, size_t n, float lvs) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = mean[i] + expf(0.5f * (lvs * logvar[i])) * noise[i];
}
__global__ void iaf_gauss_logps_kernel(const float* mean, const float* logvar, const float* sample, float* out, size_t n, float lvs) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = sample[i] - mean[i], lv = lvs * logvar[i];
        out[i] = -0.5f * (1.8378770664093453f + lv + d * d / expf(lv));
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of models.cvae_layer with posterior 'up_iaf2_nl' (models.py:168-176, 201-210, 295-298, 454-466), the elementwise
// parts on either side of iaf_step_backward.  kl = logq0 + logdet - logp(z);  G = d obj / d kl (free bits: gate[c] * gscale,
// else dko[b]).
//   pre:  dz_tot = dz + G (z - pz_mean) exp(-2 pz_logsd) [+ d_up_z]      (the gradient that enters the IAF step: -G dlogp/dz)
//         Gf = G                                                           (contiguous, the step's d logdet)
//         d_down_conv1 = [d_hdet | -G dlt e2 | G (1 - dlt^2 e2)]            (models.py:296-297 channel order)
//   post: d_up_conv1 = [d_up_hdet | dz0 | dz0 (z0 - qz_mean) - G | dctx]    (models.py:141-143; logq0 = -(log 2pi + 2 qz_logsd + eps^2)/2)
// d_h / d_up are [B, n_h + n_z, HW] (concat([h_det, z]) order, models.py:176,318), d_up may be NULL (zeros).
// ---------------------------------------------------------------------------------------------
struct UpIafBwdP {
    const float *z, *pz_mean, *pz_logsd, *d_h, *d_up, *gate, *dko;
    float *dz_tot, *Gf, *d_dc1;
    const float *dz0, *z0, *qz_mean, *dctx;
    float* d_uc1;
    float gscale;
    int B, n_h, n_z, HW;
};
__global__ __launch_bounds__(256) void iaf_up_iaf2_bwd_pre_kernel(UpIafBwdP p) {
    const size_t nh = (size_t)p.n_h * p.HW, nz = (size_t)p.n_z * p.HW, per_in = nh + nz, per_out = nh + 2 * nz;
    const size_t total = (size_t)p.B * per_out;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_out, r = i - b * per_out;
        if (r < nh) { p.d_dc1[i] = p.d_h[b * per_in + r]; continue; }                    // d h_det passes through
        const bool second = r >= nh + nz;
        const size_t e = r - nh - (second ? nz : 0), zi = b * nz + e;                     // element of a [B, n_z, HW] tensor
        const int c = (int)(e / p.HW);
        const float G = p.gate ? p.gate[c] * p.gscale : p.dko[b];
        const float e2 = expf(-2.0f * p.pz_logsd[zi]), dlt = p.z[zi] - p.pz_mean[zi];
        if (!second) {
            float t = p.d_h[b * per_in + nh + e] + G * dlt * e2;
            if (p.d_up) t += p.d_up[b * per_in + nh + e];
            p.dz_tot[zi] = t;
            p.Gf[zi] = G;
            p.d_dc1[i] = -G * dlt * e2;
        } else {
            p.d_dc1[i] = G * (1.0f - dlt * dlt * e2);
        }
    }
}
__global__ __launch_bounds__(256) void iaf_up_iaf2_bwd_post_kernel(UpIafBwdP p) {
    const size_t nh = (size_t)p.n_h * p.HW, nz = (size_t)p.n_z * p.HW, per_in = nh + nz, per_out = 2 * nh + 2 * nz;
    const size_t total = (size_t)p.B * per_out;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per_out, r = i - b * per_out;
        if (r < nh) p.d_uc1[i] = p.d_up ? p.d_up[b * per_in + r] : 0.f;
        else if (r < nh + nz) p.d_uc1[i] = p.dz0[b * nz + (r - nh)];
        else if (r < nh + 2 * nz) {
            const size_t zi = b * nz + (r - nh - nz);
            p.d_uc1[i] = p.dz0[zi] * (p.z0[zi] - p.qz_mean[zi]) - p.Gf[zi];
        } else p.d_uc1[i] = p.dctx[b * nh + (r - nh - 2 * nz)];
    }
}

// Adamax (tf_utils/adamax.py:40-56; NB the reference's slot naming: "v" = first moment, "m" = infinity norm) fused
// with the 1/N gradient averaging of average_grads (tf_utils/common.py:86) and the EMA of the parameters
// (tf_train.py:157-158, decay 0.999).  One pass over flat fp32 buffers: 5 reads + 4 writes per element, HBM-bound.
__global__ __launch_bounds__(256) void iaf_adamax_ema_kernel(float* __restrict__ var, const float* __restrict__ grad,
                                                            float* __restrict__ slot_m, float* __restrict__ slot_v,
                                                            float* __restrict__ ema, size_t n4, size_t n, float lr, float beta1,
                                                            float beta2, float eps, float ema_decay, float grad_scale) {
    auto upd = [&](float& w, float g, float& m, float& v, float& e) {
        g *= grad_scale;
        v = beta1 * v + (1.f - beta1) * g;                    // adamax.py:50
        m = fmaxf(beta2 * m + eps, fabsf(g));                 // adamax.py:52
        w -= lr * (v / m);                                    // adamax.py:53-55
        e -= (1.f - ema_decay) * (e - w);                     // ExponentialMovingAverage.apply
    };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 w = ((f32x4*)var)[i], g = ((const f32x4*)grad)[i], m = ((f32x4*)slot_m)[i], v = ((f32x4*)slot_v)[i];
        f32x4 e = ema ? ((f32x4*)ema)[i] : w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float wr_ = w[r], mr = m[r], vr = v[r], er = e[r];
            upd(wr_, g[r], mr, vr, er);
            w[r] = wr_; m[r] = mr; v[r] = vr; e[r] = er;
        }
        ((f32x4*)var)[i] = w; ((f32x4*)slot_m)[i] = m; ((f32x4*)slot_v)[i] = v;
        if (ema) ((f32x4*)ema)[i] = e;
    }
    for (size_t i = 4 * n4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {   // tail
        float e = ema ? ema[i] : var[i];
        upd(var[i], grad[i], slot_m[i], slot_v[i], e);
        if (ema) ema[i] = e;
    }
}

// streaming logsumexp over k importance weights per image: one wave per image
__global__ __launch_bounds__(256) void iaf_lb_update_kernel(float* run_max, float* run_sum, const float* log_pxz,
                                                           const float* sum_kl, int n, int kc) {
    const int img = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (img >= n) return;
    const float* a = log_pxz + (size_t)img * kc;
    const float* b = sum_kl + (size_t)img * kc;
    float m = -INFINITY;
    for (int i = lane; i < kc; i += 64) m = fmaxf(m, a[i] - b[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float old_m = run_max[img];
    const float new_m = fmaxf(old_m, m);
    float s = 0.f;
    for (int i = lane; i < kc; i += 64) s += expf((a[i] - b[i]) - new_m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float old_s = run_sum[img];
        run_sum[img] = (old_m == -INFINITY ? 0.f : old_s * expf(old_m - new_m)) + s;
        run_max[img] = new_m;
    }
}
__global__ void iaf_lb_init_kernel(float* run_max, float* run_sum, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { run_max[i] = -INFINITY; run_sum[i] = 0.f; }
}
__global__ void iaf_lb_finalize_kernel(const float* run_max, const float* run_sum, float* out, int n, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -(-logf((float)k) + run_max[i] + logf(run_sum[i]));   // distributions.py:62
}
__global__ void iaf_lb_k1_kernel(const float* log_pxz, const float* sum_kl, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sum_kl[i] - log_pxz[i];                                 // distributions.py:57
}

// ---------------------------------------------------------------------------------------------
// data-dependent init, tf_utils/layers.py:45-51: per-channel moments of x_init = conv(x, l2norm(mask*V)) over (N,H,W),
//   scale = init_scale / sqrt(var + 1e-10);  g = log(scale)/3;  b = -mean*scale;  y = scale*(x_init - mean)
// One workgroup per channel; two passes (mean, then centred variance) in a fixed tree order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void iaf_datainit_kernel(const float* x, const float* __restrict__ add, float* y,
                                                          float* __restrict__ g, float* __restrict__ b, int B, int C, int HW,
                                                          float init_scale) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int n = B * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { const int bb = i / HW; s += x[((size_t)bb * C + c) * HW + (i - bb * HW)]; }
    const float mean = block_sum_256(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int bb = i / HW;
        const float d = x[((size_t)bb * C + c) * HW + (i - bb * HW)] - mean;
        q += d * d;
    }
    const float var = block_sum_256(q, red) / (float)n;          // tf.nn.moments: biased
    const float scale = init_scale / sqrtf(var + 1e-10f);
    if (threadIdx.x == 0) { g[c] = logf(scale) / 3.0f; b[c] = -mean * scale; }
    if (y)
        for (int i = threadIdx.x; i < n; i += 256) {
            const int bb = i / HW;
            const size_t o = ((size_t)bb * C + c) * HW + (i - bb * HW);
            y[o] = scale * (x[o] - mean) + (add ? add[o] : 0.f);
        }
}

// discretized logistic log-likelihood, tf_utils/distributions.py:28-32 (call site tf_train.py:210): one workgroup per
// batch row, out[b] = sum log(sigmoid(s + binsize/scale) - sigmoid(s) + 1e-7), s = (floor(x/binsize)*binsize - mean)/scale
__global__ __launch_bounds__(256) void iaf_disc_logistic_kernel(const float* __restrict__ mean, const float* __restrict__ logscale,
                                                               int scalar_scale, const float* __restrict__ sample,
                                                               float* __restrict__ out, size_t n, float binsize) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * n;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 256) {
        const float scale = expf(scalar_scale ? logscale[0] : logscale[base + i]);
        const float s = (floorf(sample[base + i] / binsize) * binsize - mean[base + i]) / scale;
        // sigmoid(s+d) - sigmoid(s), evaluated on the side where both terms are small (sigmoid(t) = 1 - sigmoid(-t)):
        // the literal fp32 form cancels to ~1e-7 absolute in the upper tail, the size of the +1e-7 floor itself
        const float d = binsize / scale;
        float diff;
        if (s > 0.f) {
            const float e0 = expf(-s), e1 = expf(-(s + d));
            diff = e0 / (1.0f + e0) - e1 / (1.0f + e1);
        } else {
            diff = 1.0f / (1.0f + expf(-(s + d))) - 1.0f / (1.0f + expf(-s));
        }
        acc += logf(diff + 1e-7f);
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

