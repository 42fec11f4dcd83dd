// iaf_step_fused_inst.hip -- instantiations of the one-launch IAF step (iaf_step_fused.hpp) for the geometries the
// BASELINE configs use: n_h = 160 / n_z = 32 / depth_ar = 2 (configs 1-2, 5: README run) and n_h = 64 / depth_ar = 1
// (config 0), images 16, 8 and 4 pixels wide (4-pixel rows: one workgroup per four rows, i.e. per 4x4 image); and for the deep
// stack of config 3 (n_z = 64, depth_ar = 4, n_h = 64 / 128 / 192) wherever its five LDS regions fit 160 KiB.  Built as its own translation unit by iaf_amd/build.py.
#include "iaf_step_fused.hpp"

template <int NHT, int NZT, int DEPTH, int W, int R, int XCH = 0>
static step_fn_t inst(int var, size_t* lds, size_t* xrow) {
    typedef StepGeom<NHT, NZT, DEPTH, W, R, XCH> G;
    static_assert((G::CSTR & 15) == 4 || (G::CSTR & 15) == 12, "context rows: 4 channel groups x 16 pixels must hit 64 distinct banks");
    // does the geometry fit 160 KiB, the staged context its region, the exchange buffer the regions that are dead by then?
    constexpr bool fits = G::lds_bytes() <= 160 * 1024 && (DEPTH < 2 || G::ctx_bytes() <= (size_t)(G::END - G::HREG1) * 16) &&
                          (DEPTH % 2 != 0 || G::xb_bytes() <= (size_t)G::HREG1 * 16);
    if constexpr (!fits) {
        (void)var;
        return nullptr;
    } else {
        *lds = G::lds_bytes();
        if (xrow) *xrow = G::xrow_bytes();
        if constexpr (XCH) {
            switch (var) {                                    // (the statement's variant, in the exchange form)
                case 0: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 0, 1>;
                case 1: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 1, 1>;
                case 2: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 2, 1>;
            }
            return nullptr;
        } else {
            switch (var) {
                case 0: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 0>;
                case 1: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 1>;
                case 2: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 2>;
            }
            return nullptr;
        }
    }
}

// the recomputing kernel WITH helper waves (posterior block: the free-bits reductions inside the launch)
template <int NHT, int NZT, int DEPTH, int W, int R>
static step_fn_t inst_h(int var, size_t* lds) {
    typedef StepGeom<NHT, NZT, DEPTH, W, R, 0> G;
    constexpr bool fits = G::lds_bytes() <= 160 * 1024 && (DEPTH < 2 || G::ctx_bytes() <= (size_t)(G::END - G::HREG1) * 16) &&
                          (DEPTH % 2 != 0 || G::xb_bytes() <= (size_t)G::HREG1 * 16);
    if constexpr (!fits) {
        (void)var;
        return nullptr;
    } else {
        *lds = G::lds_bytes();
        switch (var) {
            case 0: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 0, 0, 1>;
            case 1: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 1, 0, 1>;
            case 2: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 2, 0, 1>;
        }
        return nullptr;
    }
}

// the pair form (iaf_step_fused.hpp, PAIR): two workgroups per (image, row block), each half of the last hidden layer and of the output pair
template <int NHT, int NZT, int DEPTH, int W, int R>
static step_fn_t inst_p(int var, size_t* lds, size_t* prow) {
    typedef StepGeom<NHT, NZT, DEPTH, W, R, 0, 1> G;
    static_assert((G::CSTR & 15) == 4 || (G::CSTR & 15) == 12, "context rows: 4 channel groups x 16 pixels must hit 64 distinct banks");
    static_assert(G::lds_bytes() <= 160 * 1024 && G::ctx_bytes() <= (size_t)(G::END - G::HREG1) * 16 && G::xb_bytes() <= (size_t)G::HREG1 * 16,
                  "the pair form's regions");
    *lds = G::lds_bytes();
    *prow = G::prow_bytes();
    switch (var) {
        case 0: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 0, 0, 1, 1>;
        case 1: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 1, 0, 1, 1>;
        case 2: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 2, 0, 1, 1>;
    }
    return nullptr;
}

template <int NHT, int NZT, int DEPTH>
static step_fn_t inst_wr(int W, int R, int var, size_t* lds) {
    if (W == 16 && R == 2) return inst<NHT, NZT, DEPTH, 16, 2>(var, lds, nullptr);
    if (W == 8 && R == 1) return inst<NHT, NZT, DEPTH, 8, 1>(var, lds, nullptr);
    if (W == 8 && R == 2) return inst<NHT, NZT, DEPTH, 8, 2>(var, lds, nullptr);
    if (W == 4 && R == 4) return inst<NHT, NZT, DEPTH, 4, 4>(var, lds, nullptr);
    return nullptr;
}

// two translation units (iaf_amd/build.py: -DIAF_FUSED_PART=0 / 1) so that the build compiles them side by side
// (no -DIAF_FUSED_PART: both parts in one unit)
#if !defined(IAF_FUSED_PART) || IAF_FUSED_PART == 0
// the halo-exchange kernels: the BASELINE run's 16-pixel geometry, all three statements (at 8-pixel rows, one row per workgroup,
// the exchange costs more than the recompute it saves: 40.6 k against 35.6 k cycles)
extern "C" step_fn_t iaf_pick_step_fused_xch(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow) {
    *lds = 0; *xrow = 0;
    if (nht == 10 && nzt == 2 && depth == 2 && W == 16 && R == 2) return inst<10, 2, 2, 16, 2, 1>(var, lds, xrow);
    return nullptr;
}
// ... with helper waves: the BASELINE run's 8-pixel geometry (same LDS layout as the plain form: call with the plain form's R)
extern "C" step_fn_t iaf_pick_step_fused_h(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    *lds = 0;
    if (nht == 10 && nzt == 2 && depth == 2 && W == 8 && R == 1) return inst_h<10, 2, 2, 8, 1>(var, lds);
    if (nht == 10 && nzt == 2 && depth == 2 && W == 8 && R == 2) return inst_h<10, 2, 2, 8, 2>(var, lds);
    return nullptr;
}
extern "C" step_fn_t iaf_pick_step_fused_a(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    *lds = 0;
    if (nht == 10 && nzt == 2 && depth == 2) return inst_wr<10, 2, 2>(W, R, var, lds);      // configs 1-2, 5 (README run)
    if (nht == 4 && nzt == 2 && depth == 1) return inst_wr<4, 2, 1>(W, R, var, lds);        // config 0
    return nullptr;
}
#endif
#if !defined(IAF_FUSED_PART) || IAF_FUSED_PART == 2
// depth_ar = 3 (models.py:92 allows any depth; README.md:49 sweeps it): the n_z = 32 families of configs 0-2.  Hidden layers ping-pong
// between the two LDS regions, so an odd depth is the same code (the output pair's exchange buffer moves: StepGeom::XB_OFF).
extern "C" step_fn_t iaf_pick_step_fused_xch_c(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow) {
    *lds = 0; *xrow = 0;
    if (nht == 10 && nzt == 2 && depth == 3 && W == 16 && R == 2) return inst<10, 2, 3, 16, 2, 1>(var, lds, xrow);
    return nullptr;
}
extern "C" step_fn_t iaf_pick_step_fused_c(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    *lds = 0;
    if (nht == 10 && nzt == 2 && depth == 3) return inst_wr<10, 2, 3>(W, R, var, lds);
    if (nht == 4 && nzt == 2 && depth == 3) return inst_wr<4, 2, 3>(W, R, var, lds);
    return nullptr;
}
#endif
#if !defined(IAF_FUSED_PART) || IAF_FUSED_PART == 1
// config 3 (up_iaf2_nl, n_z = 64, depth_ar = 4; n_h is not fixed by the reference's scripts): the geometries that fit
// ... in the halo-exchange form: the regions hold R + 1 rows instead of R + depth_ar, which is what lets n_h = 128 / 192 fit
// 160 KiB at 16-pixel rows (150 KiB at n_h = 192; the recomputing form needs 210)
extern "C" step_fn_t iaf_pick_step_fused_xch_b(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow) {
    *lds = 0; *xrow = 0;
    if (nzt == 4 && depth == 4 && W == 16 && R == 2) {
        if (nht == 4) return inst<4, 4, 4, 16, 2, 1>(var, lds, xrow);
        if (nht == 8) return inst<8, 4, 4, 16, 2, 1>(var, lds, xrow);
        if (nht == 12) return inst<12, 4, 4, 16, 2, 1>(var, lds, xrow);
    }
    return nullptr;
}
extern "C" step_fn_t iaf_pick_step_fused_b(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    *lds = 0;
    if (nht == 4 && nzt == 4 && depth == 4) return inst_wr<4, 4, 4>(W, R, var, lds);
    if (nht == 8 && nzt == 4 && depth == 4) return inst_wr<8, 4, 4>(W, R, var, lds);
    if (nht == 12 && nzt == 4 && depth == 4) return inst_wr<12, 4, 4>(W, R, var, lds);
    return nullptr;
}
#endif

#if !defined(IAF_FUSED_PART) || IAF_FUSED_PART == 3
// the pair form: the BASELINE run's 8-pixel geometry (R = 2: the pair's 16 pixels)
extern "C" step_fn_t iaf_pick_step_fused_pair(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* prow) {
    *lds = 0; *prow = 0;
    if (nht == 10 && nzt == 2 && depth == 2 && W == 8 && R == 2) return inst_p<10, 2, 2, 8, 2>(var, lds, prow);
    return nullptr;
}
#endif

#if !defined(IAF_FUSED_PART) || IAF_FUSED_PART == 4
// siblings of the README run (round 5; models.py:92 takes any n_h, README.md:49 sweeps the flow's depth): n_z = 32 with depth_ar = 2 at
// n_h = 64 and n_h = 128 -- the 16-pixel rows in the exchange form, 8- and 4-pixel rows recomputing
extern "C" step_fn_t iaf_pick_step_fused_xch_d(int nht, int nzt, int depth, int W, int R, int var, size_t* lds, size_t* xrow) {
    *lds = 0; *xrow = 0;
    if (nzt == 2 && depth == 2 && W == 16 && R == 2) {
        if (nht == 4) return inst<4, 2, 2, 16, 2, 1>(var, lds, xrow);
        if (nht == 8) return inst<8, 2, 2, 16, 2, 1>(var, lds, xrow);
    }
    return nullptr;
}
extern "C" step_fn_t iaf_pick_step_fused_d(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    *lds = 0;
    if (nht == 4 && nzt == 2 && depth == 2) return inst_wr<4, 2, 2>(W, R, var, lds);
    if (nht == 8 && nzt == 2 && depth == 2) return inst_wr<8, 2, 2>(W, R, var, lds);
    return nullptr;
}
#endif

// the two-plane fp16 kernels (round 6, "f16x2": three part-products per K step instead of six): the BASELINE run's geometries, TF
// statement -- 16-pixel rows in the exchange form (form 1), 8-pixel rows recomputing with helper waves (form 0) -- and config 3's
template <int NHT, int NZT, int DEPTH, int W, int R, int XCH>
static step_fn_t inst_f16(int var, size_t* lds, size_t* xrow) {
    typedef StepGeom<NHT, NZT, DEPTH, W, R, XCH, 0, 1> G;
    static_assert((G::CSTR & 15) == 4 || (G::CSTR & 15) == 12, "context rows: 4 channel groups x 16 pixels must hit 64 distinct banks");
    // (the staged context may reach past the h_odd region's end here -- two-plane regions are smaller than the fp32 context rows --;
    //  StepGeom::lds_bytes() covers it, and nothing else lives behind that region)
    static_assert(G::lds_bytes() <= 160 * 1024 && (DEPTH % 2 != 0 || G::xb_bytes() <= (size_t)G::HREG1 * 16), "the two-plane regions");
    *lds = G::lds_bytes();
    if (xrow) *xrow = XCH ? G::xrow_bytes() : 0;
    switch (var) {
        case 0: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 0, XCH, 1, 0, 1>;
        case 1: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 1, XCH, 1, 0, 1>;
        case 2: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 2, XCH, 1, 0, 1>;
    }
    return nullptr;
}
// TF statement only
template <int NHT, int NZT, int DEPTH, int W, int R, int XCH>
static step_fn_t inst_f16_tf(int var, size_t* lds, size_t* xrow) {
    typedef StepGeom<NHT, NZT, DEPTH, W, R, XCH, 0, 1> G;
    static_assert(G::lds_bytes() <= 160 * 1024 && (DEPTH % 2 != 0 || G::xb_bytes() <= (size_t)G::HREG1 * 16), "the two-plane regions");
    if (var != 0) return nullptr;
    *lds = G::lds_bytes();
    if (xrow) *xrow = XCH ? G::xrow_bytes() : 0;
    return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 0, XCH, 1, 0, 1>;
}
#if !defined(IAF_FUSED_PART) || IAF_FUSED_PART == 5
extern "C" step_fn_t iaf_pick_step_fused_f16_a(int nht, int nzt, int depth, int W, int R, int var, int form, size_t* lds, size_t* xrow) {
    *lds = 0; *xrow = 0;
    if (nht != 10 || nzt != 2 || depth != 2) return nullptr;
    if (form == 1 && W == 16 && R == 2) return inst_f16_tf<10, 2, 2, 16, 2, 1>(var, lds, xrow);
    if (form == 0 && W == 8 && R == 1) return inst_f16_tf<10, 2, 2, 8, 1, 0>(var, lds, xrow);
    if (form == 0 && W == 8 && R == 2) return inst_f16_tf<10, 2, 2, 8, 2, 0>(var, lds, xrow);
    return nullptr;
}
#endif
#if !defined(IAF_FUSED_PART) || IAF_FUSED_PART == 6
// config 3 (n_z = 64, depth_ar = 4; n_h = 64 / 128) in the exchange form on fp16 planes, all three statements
extern "C" step_fn_t iaf_pick_step_fused_f16_b(int nht, int nzt, int depth, int W, int R, int var, int form, size_t* lds, size_t* xrow) {
    *lds = 0; *xrow = 0;
    if (nzt != 4 || depth != 4 || form != 1 || W != 16 || R != 2) return nullptr;
    if (nht == 4) return inst_f16<4, 4, 4, 16, 2, 1>(var, lds, xrow);
    if (nht == 8) return inst_f16_tf<8, 4, 4, 16, 2, 1>(var, lds, xrow);      // (its Theano variants spill 28 VGPRs: bf16x3 for those)
    return nullptr;            // (n_h = 192: the second accumulator set of three tiles per pixel tile does not fit 256 registers -- bf16x3 there)
}
#endif
