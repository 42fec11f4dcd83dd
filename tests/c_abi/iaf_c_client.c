/*
 * iaf_c_client.c -- a plain C program (no Python, no torch) that drives the engine through include/iaf_hip.h exactly
 * the way a foreign-language binding would: device buffers from the HIP runtime, pointers and sizes across the ABI,
 * status codes back.  TEST CODE: it checks the result against the plain-C oracle (oracle/iaf_oracle.c), which is why
 * it lives under tests/.  Built and run by tests/test_c_abi_client.py (gpu-marked).
 *
 *   usage: iaf_c_client n_z n_h depth_ar B H W        exit 0 = within tolerance
 *
 * What it exercises: iaf_stack_create / _prepare / _workspace_bytes / iaf_step_forward / iaf_step_inverse /
 * iaf_conv3x3_create / _prepare / _forward / destroy, and the error paths a binding has to map (NULL, not prepared,
 * short workspace).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "iaf_hip.h"

/* oracle/iaf_oracle.c */
int iaf_oracle_c_step(const double* z, const double* context, const double* const* V, const double* const* g,
                      const double* const* b, int n_z, int n_h, int depth_ar, int B, int H, int W, double* z_new,
                      double* logsd, double* m_raw_out, double* s_raw_out);

#define CHECK_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %d at line %d\n", (int)_e, __LINE__); return 2; } } while (0)
#define CHECK_IAF(e) do { int _r = (e); if (_r != IAF_OK) { fprintf(stderr, "iaf error %d (%s) at line %d\n", _r, iaf_error_string(_r), __LINE__); return 3; } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {      /* xorshift64*, then to (0,1) */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return ((rng_state * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0) + 1e-17;
}
static double nrand(void) { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

static float* to_device(const double* h, size_t n) {
    float* tmp = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) tmp[i] = (float)h[i];
    float* d = NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    hipMemcpy(d, tmp, n * sizeof(float), hipMemcpyHostToDevice);
    free(tmp);
    return d;
}
static void round_to_f32(double* h, size_t n) { for (size_t i = 0; i < n; ++i) h[i] = (double)(float)h[i]; }
static double max_err(const float* d, const double* want, size_t n) {
    float* tmp = (float*)malloc(n * sizeof(float));
    hipMemcpy(tmp, d, n * sizeof(float), hipMemcpyDeviceToHost);
    double m = 0;
    for (size_t i = 0; i < n; ++i) { double e = fabs((double)tmp[i] - want[i]); if (!(e <= m)) m = e; }
    free(tmp);
    return m;
}

int main(int argc, char** argv) {
    if (argc != 7) { fprintf(stderr, "usage: %s n_z n_h depth_ar B H W\n", argv[0]); return 1; }
    const int n_z = atoi(argv[1]), n_h = atoi(argv[2]), d = atoi(argv[3]), B = atoi(argv[4]), H = atoi(argv[5]), W = atoi(argv[6]);
    if (iaf_abi_version() < 2) { fprintf(stderr, "ABI too old\n"); return 1; }
    if (iaf_device_count() < 1) { fprintf(stderr, "no device\n"); return 1; }
    const size_t P = (size_t)B * H * W;
    const int nconv = d + 2;

    /* variables in the reference's layout: V HWIO [3,3,n_in,n_out], g, b [n_out]  (layers.py:53-55) */
    double** V = (double**)calloc(nconv, sizeof(double*));
    double** g = (double**)calloc(nconv, sizeof(double*));
    double** b = (double**)calloc(nconv, sizeof(double*));
    const float** dV = (const float**)calloc(nconv, sizeof(float*));
    const float** dg = (const float**)calloc(nconv, sizeof(float*));
    const float** db = (const float**)calloc(nconv, sizeof(float*));
    for (int c = 0; c < nconv; ++c) {
        const int n_in = (c == 0) ? n_z : n_h, n_out = (c < d) ? n_h : n_z;
        const size_t nv = (size_t)9 * (d == 0 ? n_z : n_in) * n_out;
        V[c] = (double*)malloc(nv * sizeof(double));
        g[c] = (double*)malloc(n_out * sizeof(double));
        b[c] = (double*)malloc(n_out * sizeof(double));
        for (size_t i = 0; i < nv; ++i) V[c][i] = 0.05 * nrand();
        for (int i = 0; i < n_out; ++i) { g[c][i] = 0.1 * nrand(); b[c][i] = 0.1 * nrand(); }
        round_to_f32(V[c], nv); round_to_f32(g[c], n_out); round_to_f32(b[c], n_out);
        dV[c] = to_device(V[c], nv); dg[c] = to_device(g[c], n_out); db[c] = to_device(b[c], n_out);
        if (!dV[c] || !dg[c] || !db[c]) return 2;
    }
    double* z = (double*)malloc(P * n_z * sizeof(double));
    double* ctx = (double*)malloc(P * n_h * sizeof(double));
    for (size_t i = 0; i < P * n_z; ++i) z[i] = nrand();
    for (size_t i = 0; i < P * n_h; ++i) ctx[i] = nrand();
    round_to_f32(z, P * n_z); round_to_f32(ctx, P * n_h);
    float* dz = to_device(z, P * n_z);
    float* dctx = to_device(ctx, P * n_h);
    float *dznew = NULL, *dlogsd = NULL, *dback = NULL, *dls2 = NULL;
    CHECK_HIP(hipMalloc((void**)&dznew, P * n_z * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dlogsd, P * n_z * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dback, P * n_z * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&dls2, P * n_z * sizeof(float)));

    iaf_stack_t* st = NULL;
    CHECK_IAF(iaf_stack_create(&st, n_z, n_h, d, IAF_VARIANT_TF));
    const size_t wsb = iaf_stack_workspace_bytes(st, B, H, W);
    void* ws = NULL;
    CHECK_HIP(hipMalloc(&ws, wsb));

    /* the statuses a binding must map to exceptions */
    if (iaf_step_forward(st, dz, dctx, dznew, dlogsd, B, H, W, ws, wsb, NULL) != IAF_ERR_NOT_PREPARED) { fprintf(stderr, "expected NOT_PREPARED\n"); return 4; }
    CHECK_IAF(iaf_stack_prepare(st, dV, dg, db, NULL));
    if (iaf_step_forward(st, NULL, dctx, dznew, dlogsd, B, H, W, ws, wsb, NULL) != IAF_ERR_NULL) { fprintf(stderr, "expected ERR_NULL\n"); return 4; }
    if (wsb > 256 && iaf_step_forward(st, dz, dctx, dznew, dlogsd, B, H, W, ws, wsb - 256, NULL) != IAF_ERR_WORKSPACE) { fprintf(stderr, "expected ERR_WORKSPACE\n"); return 4; }

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_IAF(iaf_step_forward(st, dz, dctx, dznew, dlogsd, B, H, W, ws, wsb, (void*)stream));
    int sweeps = 0; float res = -1.f;
    CHECK_IAF(iaf_step_inverse(st, dznew, dctx, dback, dls2, B, H, W, ws, wsb, 100, 1e-6f, 2, (void*)stream, &sweeps, &res));
    CHECK_HIP(hipStreamSynchronize(stream));

    double* ez = (double*)malloc(P * n_z * sizeof(double));
    double* es = (double*)malloc(P * n_z * sizeof(double));
    iaf_oracle_c_step(z, ctx, (const double* const*)V, (const double* const*)g, (const double* const*)b, n_z, n_h, d, B, H, W, ez, es, NULL, NULL);
    const double e1 = max_err(dznew, ez, P * n_z), e2 = max_err(dlogsd, es, P * n_z), e3 = max_err(dback, z, P * n_z);
    printf("iaf_step_forward: max|z_new - oracle| = %.3g, max|logsd - oracle| = %.3g; inverse round trip %.3g after %d sweeps (last update %.2g)\n",
           e1, e2, e3, sweeps, (double)res);

    /* one plain conv with fused ELU and a two-way split, against the same C oracle used as a single unmasked-less check:
     * here only the call sequence and finiteness are checked -- numeric parity of the plain convs is tests/test_hip_layer.py */
    int rc_plain = 0;
    if (n_h % 16 == 0) {
        iaf_conv3x3_t* cv = NULL;
        CHECK_IAF(iaf_conv3x3_create(&cv, n_h, 2 * n_h));
        const size_t nv = (size_t)9 * n_h * 2 * n_h;
        double* Vp = (double*)malloc(nv * sizeof(double));
        double* gp = (double*)calloc(2 * n_h, sizeof(double));
        for (size_t i = 0; i < nv; ++i) Vp[i] = 0.05 * nrand();
        float* dVp = to_device(Vp, nv); float* dgp = to_device(gp, 2 * n_h); float* dbp = to_device(gp, 2 * n_h);
        float *o0 = NULL, *o1 = NULL;
        CHECK_HIP(hipMalloc((void**)&o0, P * n_h * sizeof(float)));
        CHECK_HIP(hipMalloc((void**)&o1, P * n_h * sizeof(float)));
        CHECK_IAF(iaf_conv3x3_prepare(cv, dVp, dgp, dbp, (void*)stream));
        float* outs[2] = {o0, o1};
        const int chans[2] = {n_h, n_h};
        CHECK_IAF(iaf_conv3x3_forward(cv, dctx, NULL, 0, 1, NULL, outs, chans, 2, B, H, W, (void*)stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        float* tmp = (float*)malloc(P * n_h * sizeof(float));
        hipMemcpy(tmp, o1, P * n_h * sizeof(float), hipMemcpyDeviceToHost);
        double ss = 0;
        for (size_t i = 0; i < P * n_h; ++i) { if (!isfinite(tmp[i])) rc_plain = 5; ss += (double)tmp[i] * tmp[i]; }
        printf("iaf_conv3x3_forward: rms of the second split output %.4f (unit-norm filters on elu(N(0,1)) input)\n", sqrt(ss / (P * n_h)));
        if (!(ss > 0)) rc_plain = 5;
        CHECK_IAF(iaf_conv3x3_destroy(cv));
        free(tmp); free(Vp); free(gp);
    }
    CHECK_IAF(iaf_stack_destroy(st));
    const double tol = 1e-4;
    if (!(e1 <= tol && e2 <= tol && e3 <= tol) || rc_plain) { fprintf(stderr, "FAILED (tolerance %.1g)\n", tol); return 10; }
    printf("OK\n");
    return 0;
}
