// dev probe: iaf_wgrad_bf3_kernel alone -- checks it against a double-precision host sum on sampled outputs, times it over
// pixel-range counts, and prints the phase stamps (IAF_WSTAMP) of its workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIAF_WSTAMP -Iinclude -Iiaf_amd/csrc tools/probe/wgrad_probe.hip -o tools/probe/bin/wgrad_probe
//   tools/probe/bin/wgrad_probe [cin cout ntaps B H W]
#include "../../iaf_amd/csrc/iaf_wgrad_bf3.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    int cin = argc > 1 ? atoi(argv[1]) : 160, cout = argc > 2 ? atoi(argv[2]) : 224, ntaps = argc > 3 ? atoi(argv[3]) : 9;
    int B = argc > 4 ? atoi(argv[4]) : 32, H = argc > 5 ? atoi(argv[5]) : 16, W = argc > 6 ? atoi(argv[6]) : 16;
    const int P = B * H * W;
    const int ncob = iaf_wgrad_bf3_ncob(cin, cout);
    if (!ncob) { printf("not covered\n"); return 1; }
    std::vector<float> hx((size_t)P * cin), hy((size_t)P * cout);
    srand(1);
    for (auto& v : hx) v = (float)((double)rand() / RAND_MAX) - 0.5f;
    for (auto& v : hy) v = ((float)((double)rand() / RAND_MAX) - 0.5f) * 0.01f;
    std::vector<unsigned short> hm(P);
    for (int p = 0; p < P; ++p) {
        const int h = (p / W) % H, w = p % W;
        unsigned short m = 0;
        for (int dh = -1; dh <= 1; ++dh)
            for (int dw = -1; dw <= 1; ++dw)
                if (h + dh >= 0 && h + dh < H && w + dw >= 0 && w + dw < W) m |= 1u << ((dh + 1) * 3 + dw + 1);
        hm[p] = m;
    }
    float *dx, *dy, *part; unsigned short* dm; unsigned long long* dbg;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dy, hy.size() * 4)); CK(hipMalloc(&dm, P * 2));
    CK(hipMalloc(&part, (size_t)32 * ntaps * cin * cout * 4)); CK(hipMalloc(&dbg, 4096 * 4 * 8));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, hm.data(), P * 2, hipMemcpyHostToDevice));
    WgradP p;
    memset(&p, 0, sizeof(p));
    p.x = dx; p.dy = dy; p.part = part; p.tapmask = dm;
#ifdef IAF_WSTAMP
    p.dbg = dbg;
#endif
    p.B = B; p.H = H; p.W = W; p.HW = H * W; p.P = P; p.cin = cin; p.cout = cout; p.ntaps = ntaps;
    static const int tf_dh[5] = {0, 0, 1, 1, 1}, tf_dw[5] = {0, 1, -1, 0, 1};
    for (int t = 0; t < ntaps; ++t) {
        p.tap_dh[t] = ntaps == 9 ? t / 3 - 1 : tf_dh[t];
        p.tap_dw[t] = ntaps == 9 ? t % 3 - 1 : tf_dw[t];
        int gi = 0;
        while (gi < p.ngroups && p.grp_dh[gi] != p.tap_dh[t]) ++gi;
        if (gi == p.ngroups) { p.grp_dh[gi] = p.tap_dh[t]; p.grp_n[gi] = 0; ++p.ngroups; }
        p.grp_tap[gi][p.grp_n[gi]++] = t;
    }
    p.gx = p.ngroups * (cin / 32); p.gz = (cout / 16) / ncob;
    const int units = p.gx * p.gz;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("cin %d cout %d taps %d P %d ncob %d: %d workgroups per range\n", cin, cout, ntaps, P, ncob, units);
    const int cand[] = {256 / units, 512 / units, 768 / units, 1024 / units, 8, 16, 32};
    for (int nr : cand) {
        if (nr < 1 || nr > 32 || nr > P / 64) continue;
        p.nrange = nr;
        p.px_per_range = ((P + nr - 1) / nr + 31) / 32 * 32;
        CK(hipMemset(dbg, 0, 4096 * 4 * 8));
        int rc = iaf_launch_wgrad_bf3(&p, ncob, 0);
        if (rc) { printf("launch rc %d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) iaf_launch_wgrad_bf3(&p, ncob, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const int grid = units * nr;
        std::vector<unsigned long long> st((size_t)grid * 4);
        CK(hipMemcpy(st.data(), dbg, st.size() * 8, hipMemcpyDeviceToHost));
        double s[4] = {0, 0, 0, 0};
        for (int g = 0; g < grid; ++g) for (int k = 0; k < 4; ++k) s[k] += (double)st[(size_t)g * 4 + k];
        const int nkb = (p.px_per_range + 31) / 32;
        printf("nrange %2d (%3d workgroups, %2d K blocks): %7.1f us   per K block, mean over workgroups: store %5.0f  barrier %5.0f  mfma %5.0f  rest %5.0f ticks\n",
               nr, grid, nkb, ms * 1000 / 20, s[0] / grid / nkb, s[1] / grid / nkb, s[2] / grid / nkb, s[3] / grid / nkb);
        // check sampled outputs (sum of the range partials) against a double sum
        std::vector<float> hp((size_t)nr * ntaps * cin * cout);
        CK(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int sidx = 0; sidx < 200; ++sidx) {
            const int t = rand() % ntaps, ci = rand() % cin, co = rand() % cout;
            double ref = 0, scale = 0;
            for (int q = 0; q < P; ++q) {
                if (!((hm[q] >> ((p.tap_dh[t] + 1) * 3 + p.tap_dw[t] + 1)) & 1)) continue;
                const double a = hx[(size_t)(q + p.tap_dh[t] * W + p.tap_dw[t]) * cin + ci], b = hy[(size_t)q * cout + co];
                ref += a * b; scale += fabs(a * b);
            }
            double got = 0;
            for (int r = 0; r < nr; ++r) got += hp[(((size_t)r * ntaps + t) * cin + ci) * cout + co];
            worst = std::max(worst, fabs(got - ref) / (scale + 1e-30));
        }
        printf("           worst |got - ref| / sum|terms| over 200 sampled outputs: %.3g\n", worst);
    }
    return 0;
}
