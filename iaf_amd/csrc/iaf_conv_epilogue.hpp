// iaf_conv_epilogue.hpp -- the fused epilogues of the masked-conv stack as per-unit device functions (one unit = one
// 16-channel co tile for a hidden layer, one (mean, logsd) tile pair for the output layer), for kernels whose lanes hold
// the 16x16 MFMA C/D layout: lane l has D[co = tile*16 + 4*(l>>4) + r][pixel l&15], r = 0..3.
// Used by iaf_conv_bf3.hpp; iaf_conv_kernel.hpp carries the same arithmetic inline (its operand prefetch is woven into
// its K loop).  Reference lines: tf_utils/layers.py:63-64,163-165 (bias, context add, ELU), tf_train.py:56-75 (posterior
// sample, logqs, affine transform, log-det term, logps, KL elements), graphy/nodes/conv.py:71-83 (Theano border channel).
#pragma once
#include "iaf_conv_kernel.hpp"

// q = n / d, r = n % d for 0 <= n < 2^30, d > 0 with rd = 1.0f / d: float estimate + one correction step either way
// (an integer division is ~40 instructions on this ISA; the epilogue geometry sits on every workgroup's critical path)
__device__ __forceinline__ void fast_divmod(int n, int d, float rd, int& q, int& r) {
    q = (int)((float)n * rd);
    r = n - q * d;
    if (r < 0) { q -= 1; r += d; }
    if (r < 0) { q -= 1; r += d; }
    if (r >= d) { q += 1; r -= d; }
    if (r >= d) { q += 1; r -= d; }
}

struct EpiGeom {
    int Pl, bimg, pp, kk;
    bool pvalid;
    unsigned outside;    // bit t: tap t of this lane's pixel falls outside the image (Theano border channel)
    int up;              // EPI_PLAIN, deconv2d by output phases (iaf_conv_bf3.hpp S2 = 2): 0, or 4 + 2a + b -> store at (2i+a, 2j+b) of a [2H,2W] map
};
struct EpiOps {
    f32x4 pre0, pre1, b0, b1;
};

__device__ __forceinline__ EpiGeom epi_geom(const ConvP& p, int Pl, int kk, bool active) {
    EpiGeom g;
    g.Pl = Pl; g.kk = kk;
    g.pvalid = active && Pl < p.P;
    int h, w;
    fast_divmod(Pl, p.HW, 1.0f / (float)p.HW, g.bimg, g.pp);
    fast_divmod(g.pp, p.W, 1.0f / (float)p.W, h, w);
    g.outside = 0;
    g.up = 0;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int dh = p.tap_dh[t], dw = p.tap_dw[t];
        const bool v = g.pvalid && (h + dh >= 0) && (h + dh < p.H) && (w + dw >= 0) && (w + dw < p.W);
        g.outside |= (v ? 0u : 1u) << t;
    }
    return g;
}

// operands of the epilogue that do not depend on the GEMM (issued before the split-K exchange so their latency hides)
template <int EPI>
__device__ __forceinline__ void epi_load(const ConvP& p, const EpiGeom& g, int cot, EpiOps& o) {
    if (!g.pvalid) return;
    const int HW = p.HW;
    if (EPI == EPI_PLAIN) {         // plain conv2d (layers.py:63-64): bias, and the residual of `input + 0.1 * h` (tf_train.py:44,94)
        o.b0 = *(const f32x4*)(p.bias + cot * 16 + 4 * g.kk);
        if (p.res) {
            const size_t cb = ((size_t)g.bimg * p.cout + cot * 16 + 4 * g.kk) * HW + g.pp;
#pragma unroll
            for (int r = 0; r < 4; ++r) o.pre0[r] = p.res[cb + (size_t)r * HW];
        }
    } else if (EPI == EPI_HIDDEN) {
        o.b0 = *(const f32x4*)(p.bias + cot * 16 + 4 * g.kk);
        if (p.ctx) {
            const size_t cb = ((size_t)g.bimg * p.cout + cot * 16 + 4 * g.kk) * HW + g.pp;
#pragma unroll
            for (int r = 0; r < 4; ++r) o.pre0[r] = p.ctx[cb + (size_t)r * HW];
            if (p.ctx2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o.pre1[r] = p.ctx2[cb + (size_t)r * HW];
            }
        }
    } else if (EPI == EPI_DGRAD) {  // data gradient (modes MODE_DGRAD_*, iaf_conv_kernel.hpp): no bias; operands of elu' / of the affine term
        const int co = cot * 16 + 4 * g.kk;
        const size_t cb = ((size_t)g.bimg * p.cout + co) * HW + g.pp;
        if (p.mode == MODE_DGRAD_ELU) {
            o.pre0 = *(const f32x4*)(p.zin + (size_t)g.Pl * p.cout + co);               // saved activation h, pixel-major
        } else if (p.mode == MODE_DGRAD_PLAIN) {
            if (p.zin) o.pre0 = *(const f32x4*)(p.zin + (size_t)g.Pl * p.cout + co);    // staged elu(x), pixel-major
            if (p.res) {                                                                  // `input + 0.1 h`: d_out passes through
#pragma unroll
                for (int r = 0; r < 4; ++r) o.pre1[r] = p.res[cb + (size_t)r * HW];
            }
        } else {                                                                          // MODE_DGRAD_Z: dz_new, logsd (NCHW)
#pragma unroll
            for (int r = 0; r < 4; ++r) { o.pre0[r] = p.qm[cb + (size_t)r * HW]; o.pre1[r] = p.ql[cb + (size_t)r * HW]; }
        }
    } else {
        o.b0 = *(const f32x4*)(p.bias + cot * 16 + 4 * g.kk);
        o.b1 = *(const f32x4*)(p.bias + (cot + 1) * 16 + 4 * g.kk);
        if (p.mode == MODE_IAF || p.mode == MODE_INVERSE) {
            const size_t zb = ((size_t)g.bimg * (p.cout >> 1) + (cot >> 1) * 16 + 4 * g.kk) * HW + g.pp;
#pragma unroll
            for (int r = 0; r < 4; ++r) o.pre0[r] = p.zin[zb + (size_t)r * HW];
        }
    }
}

// v0: the unit's (first) accumulator tile, v1: the logsd tile of an output pair (ignored for hidden layers)
template <int EPI, int NTP>
__device__ __forceinline__ void epi_apply(const ConvP& p, const EpiGeom& g, int cot, f32x4 v0, f32x4 v1, const EpiOps& o) {
    if (!g.pvalid) return;
    const int HW = p.HW;
    if (EPI == EPI_PLAIN) {         // NCHW store with the channel split of tf_train.py:37,54 fused in
        const int co = cot * 16 + 4 * g.kk;
        const f32x4 v = v0 + o.b0;
        int c0 = 0, c1 = p.split_end[0];
        float* base = p.split_ptr[0];
#pragma unroll
        for (int q = 1; q < MAXSPLIT; ++q)      // static indices only: the descriptor stays in SGPRs
            if (q < p.nsplit && co >= p.split_end[q - 1]) { c0 = p.split_end[q - 1]; c1 = p.split_end[q]; base = p.split_ptr[q]; }
        if (g.up) {                 // one output phase of deconv2d (layers.py:83-112): pixel (i,j) of the [H,W] grid -> (2i+a, 2j+b);
                                    // the residual is resize_nearest_neighbor(input, 2) (tf_train.py:90,94) = the low-res value
            const int i = g.pp / p.W, j = g.pp - i * p.W;
            float* dst = base + ((size_t)g.bimg * (c1 - c0) + (co - c0)) * (4 * (size_t)HW) + (size_t)(2 * i + ((g.up >> 1) & 1)) * (2 * p.W) +
                         2 * j + (g.up & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)r * 4 * HW] = p.res ? o.pre0[r] + 0.1f * v[r] : v[r];
            return;
        }
        float* dst = base + ((size_t)g.bimg * (c1 - c0) + (co - c0)) * HW + g.pp;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(size_t)r * HW] = p.res ? o.pre0[r] + 0.1f * v[r] : v[r];
    } else if (EPI == EPI_HIDDEN) {
        const int co = cot * 16 + 4 * g.kk;
        f32x4 v = v0 + o.b0;
        if (p.border) {   // Theano pad_channel (conv.py:71-83, ar.py:229-233): taps that fall outside see a 1
#pragma unroll
            for (int t = 1; t < NTP; ++t)
                if (g.outside & (1u << t)) v += *(const f32x4*)(p.border + (size_t)(t - 1) * p.cout + co);
        }
        if (p.ctx) {      // x += context (layers.py:163-164); context = up_context + down_context (tf_train.py:58)
            if (p.ctx2) v += (o.pre0 + o.pre1);
            else v += o.pre0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = elu_f(v[r]);   // layers.py:165
        *(f32x4*)(p.y + (size_t)g.Pl * p.cout + co) = v;
    } else if (EPI == EPI_DGRAD) {      // dX = W^T dY through the transposed packs and mirrored taps (iaf_conv_kernel.hpp, same modes)
        const int co = cot * 16 + 4 * g.kk;
        f32x4 v = v0;
        if (p.mode == MODE_DGRAD_ELU) {                            // d a_{l-1} = (W_l^T dY) elu'(h_{l-1}); elu'(a) = h > 0 ? 1 : h + 1
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= (o.pre0[r] > 0.f ? 1.f : o.pre0[r] + 1.f);
            *(f32x4*)(p.y + (size_t)g.Pl * p.cout + co) = v;
            if (p.out0) {                                          // (+ NCHW copy = d context)
                const size_t cb = ((size_t)g.bimg * p.cout + co) * HW + g.pp;
#pragma unroll
                for (int r = 0; r < 4; ++r) p.out0[cb + (size_t)r * HW] = v[r];
            }
        } else if (p.mode == MODE_DGRAD_PLAIN) {                   // plain conv: dx = [d_out +] act'(.) (W^T dY), NCHW through the split table
            if (p.zin) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= (o.pre0[r] > 0.f ? 1.f : o.pre0[r] + 1.f);
            }
            int c0 = 0, c1 = p.split_end[0];
            float* base = p.split_ptr[0];
#pragma unroll
            for (int q = 1; q < MAXSPLIT; ++q)
                if (q < p.nsplit && co >= p.split_end[q - 1]) { c0 = p.split_end[q - 1]; c1 = p.split_end[q]; base = p.split_ptr[q]; }
            float* dst = base + ((size_t)g.bimg * (c1 - c0) + (co - c0)) * HW + g.pp;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(size_t)r * HW] = p.res ? o.pre1[r] + v[r] : v[r];
        } else {                                                   // MODE_DGRAD_Z: dz = W_0^T dY + dz_new exp(-logsd) (tf_train.py:71)
            const size_t cb = ((size_t)g.bimg * p.cout + co) * HW + g.pp;
#pragma unroll
            for (int r = 0; r < 4; ++r) p.out0[cb + (size_t)r * HW] = v[r] + o.pre0[r] * __expf(-o.pre1[r]);
        }
    } else {
        const int nz = p.cout >> 1;
        const int c0 = (cot >> 1) * 16 + 4 * g.kk;       // packed tiles (cot, cot+1) = (mean, logsd) of channel group cot/2
        f32x4 bm = o.b0, bs = o.b1;
        if (p.border) {
#pragma unroll
            for (int t = 1; t < NTP; ++t)
                if (g.outside & (1u << t)) {
                    bm += *(const f32x4*)(p.border + (size_t)(t - 1) * p.cout + cot * 16 + 4 * g.kk);
                    bs += *(const f32x4*)(p.border + (size_t)(t - 1) * p.cout + (cot + 1) * 16 + 4 * g.kk);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t idx = ((size_t)g.bimg * nz + c0 + r) * HW + g.pp;
            const float m_raw = v0[r] + bm[r];
            const float s_raw = v1[r] + bs[r];
            if (p.mode == MODE_RAW) {
                p.out0[idx] = m_raw;
                p.out1[idx] = s_raw;
            } else if (p.mode == MODE_IAF) {
                const float m = m_raw * 0.1f, s = s_raw * 0.1f;        // tf_train.py:70
                p.out0[idx] = (o.pre0[r] - m) / __expf(s);             // tf_train.py:71
                p.out1[idx] = s;                                        // tf_train.py:72 (logqs += s)
            } else if (p.mode == MODE_INVERSE) {
                const float m = m_raw * 0.1f, s = s_raw * 0.1f;
                p.out0[idx] = o.pre0[r] * __expf(s) + m;               // tf_train.py:71 solved for the input
                p.out1[idx] = s;
            } else {
                const float m = m_raw * 0.1f, s = s_raw * 0.1f;
                const float mean = p.qm[idx] + p.rm[idx];               // tf_train.py:57
                const float logvar = 2.f * (p.ql[idx] + p.rl[idx]);
                const float z0 = mean + __expf(0.5f * logvar) * p.eps[idx];                           // :63
                const float d0 = z0 - mean;
                float logqs = -0.5f * (1.8378770664093453f + logvar + d0 * d0 / __expf(logvar));     // :68
                const float z = (z0 - m) / __expf(s);                                                 // :71
                logqs += s;                                                                           // :72
                const float plv = 2.f * p.pl[idx];                                                    // :56
                const float d1 = z - p.pm[idx];
                const float logps = -0.5f * (1.8378770664093453f + plv + d1 * d1 / __expf(plv));      // :73
                p.out0[idx] = z;
                if (p.out1) p.out1[idx] = s;
                p.kl_elem[idx] = logqs - logps;                                                       // :75
            }
        }
    }
}
