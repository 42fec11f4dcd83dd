/*
 * iaf_oracle.c -- plain-C restatement of the IAF step.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A second, independent CPU statement of the reference algorithm (direct loops, no im2col, no BLAS), used by
 * tests/ to cross-check the NumPy oracle (oracle/iaf_oracle.py) and available to bench.py as a scalar
 * cpu_baseline.  Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may load it.
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared) -> oracle/_build/libiaf_oracle_c.so
 *
 * Reference lines followed (paths relative to the reference tree):
 *   mask            tf_utils/layers.py:115-141
 *   weight-norm     tf_utils/layers.py:56-60
 *   conv + bias     tf_utils/layers.py:63-64   (NCHW, SAME, stride 1, cross-correlation, HWIO filter)
 *   stack           tf_utils/layers.py:158-166 (context after the first conv, ELU after every hidden conv)
 *   IAF step        tf_train.py:69-72
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int made_live(int i, int o, int n_in, int n_out, int zerodiag) { /* layers.py:115-131, Python-2 '/' */
    if (n_out >= n_in) {
        int k = n_out / n_in, grp = o / k;
        return zerodiag ? (i < grp) : (i <= grp);
    } else {
        int k = n_in / n_out;
        return zerodiag ? (i < o * k) : (i < (o + 1) * k);
    }
}

static int conv_mask(int kh, int kw, int i, int o, int n_in, int n_out, int zerodiag) { /* layers.py:134-141, 3x3 */
    if (kh < 1) return 0;
    if (kh == 1 && kw < 1) return 0;
    if (kh == 1 && kw == 1) return made_live(i, o, n_in, n_out, zerodiag);
    return 1;
}

/* y[B,n_out,H,W] = conv(x[B,n_in,H,W], w) + b with w = exp(g) * mask*V / sqrt(max(sum (mask*V)^2, 1e-12)) */
static void ar_conv2d(const double* x, const double* V, const double* g, const double* b, int n_in, int n_out,
                      int zerodiag, int B, int H, int W, double* y) {
    double* w = (double*)malloc(sizeof(double) * 9 * n_in * n_out);
    for (int o = 0; o < n_out; ++o) {
        double ss = 0;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw)
                for (int i = 0; i < n_in; ++i) {
                    double v = conv_mask(kh, kw, i, o, n_in, n_out, zerodiag) ? V[((kh * 3 + kw) * n_in + i) * n_out + o] : 0.0;
                    w[((kh * 3 + kw) * n_in + i) * n_out + o] = v;
                    ss += v * v;
                }
        double sc = exp(g[o]) / sqrt(ss > 1e-12 ? ss : 1e-12);
        for (int t = 0; t < 9 * n_in; ++t) w[t * n_out + o] *= sc;
    }
#pragma omp parallel for collapse(2)
    for (int n = 0; n < B; ++n)
        for (int o = 0; o < n_out; ++o)
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww) {
                    double acc = 0;
                    for (int kh = 0; kh < 3; ++kh) {
                        int hh = h + kh - 1;
                        if (hh < 0 || hh >= H) continue;
                        for (int kw = 0; kw < 3; ++kw) {
                            int wx = ww + kw - 1;
                            if (wx < 0 || wx >= W) continue;
                            for (int i = 0; i < n_in; ++i)
                                acc += x[((size_t)(n * n_in + i) * H + hh) * W + wx] * w[((kh * 3 + kw) * n_in + i) * n_out + o];
                        }
                    }
                    y[((size_t)(n * n_out + o) * H + h) * W + ww] = acc + b[o];
                }
    free(w);
}

/* V/g/b: depth_ar + 2 arrays in the order layer_0.., layer_out_0, layer_out_1.  Returns 0. */
int iaf_oracle_c_step(const double* z, const double* context, const double* const* V, const double* const* g,
                      const double* const* b, int n_z, int n_h, int depth_ar, int B, int H, int W, double* z_new,
                      double* logsd, double* m_raw_out, double* s_raw_out) {
    size_t px = (size_t)B * H * W;
    int c_in = n_z;
    double* cur = (double*)malloc(sizeof(double) * px * (n_z > n_h ? n_z : n_h));
    double* nxt = (double*)malloc(sizeof(double) * px * (n_z > n_h ? n_z : n_h));
    memcpy(cur, z, sizeof(double) * px * n_z);
    for (int l = 0; l < depth_ar; ++l) {
        ar_conv2d(cur, V[l], g[l], b[l], c_in, n_h, 0, B, H, W, nxt);                 /* layers.py:162 */
        for (size_t i = 0; i < px * n_h; ++i) {
            double v = nxt[i] + (l == 0 ? context[i] : 0.0);                           /* layers.py:163-164 */
            nxt[i] = v > 0 ? v : expm1(v);                                             /* layers.py:165 (elu) */
        }
        double* t = cur; cur = nxt; nxt = t;
        c_in = n_h;
    }
    double* m = (double*)malloc(sizeof(double) * px * n_z);
    double* s = (double*)malloc(sizeof(double) * px * n_z);
    ar_conv2d(cur, V[depth_ar], g[depth_ar], b[depth_ar], c_in, n_z, 1, B, H, W, m);   /* layers.py:166 */
    ar_conv2d(cur, V[depth_ar + 1], g[depth_ar + 1], b[depth_ar + 1], c_in, n_z, 1, B, H, W, s);
    for (size_t i = 0; i < px * n_z; ++i) {
        if (m_raw_out) m_raw_out[i] = m[i];
        if (s_raw_out) s_raw_out[i] = s[i];
        double am = 0.1 * m[i], as = 0.1 * s[i];                                       /* tf_train.py:70 */
        z_new[i] = (z[i] - am) / exp(as);                                              /* tf_train.py:71 */
        logsd[i] = as;                                                                 /* tf_train.py:72 */
    }
    free(cur); free(nxt); free(m); free(s);
    return 0;
}
