// iaf_step_fused_inst.hip -- instantiations of the one-launch IAF step (iaf_step_fused.hpp) for the geometries the
// BASELINE configs use: n_h = 160 / n_z = 32 / depth_ar = 2 (configs 1-2, 5: README run) and n_h = 64 / depth_ar = 1
// (config 0), images 16, 8 and 4 pixels wide (4-pixel rows: one workgroup per four rows, i.e. per 4x4 image).  Built as its own translation unit by iaf_amd/build.py.
#include "iaf_step_fused.hpp"

template <int NHT, int NZT, int DEPTH, int W, int R>
static step_fn_t inst(int var, size_t* lds) {
    typedef StepGeom<NHT, NZT, DEPTH, W, R> G;
    static_assert(DEPTH < 2 || G::xb_bytes() <= (size_t)G::HREG1 * 16, "the exchange buffer must fit the z + h_0 regions");
    static_assert(DEPTH < 2 || G::ctx_bytes() <= (size_t)(G::END - G::HREG1) * 16, "the staged context must fit the h_1 region");
    static_assert((G::CSTR & 15) == 4 || (G::CSTR & 15) == 12, "context rows: 4 channel groups x 16 pixels must hit 64 distinct banks");
    *lds = G::lds_bytes();
    switch (var) {
        case 0: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 0>;
        case 1: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 1>;
        case 2: return iaf_step_fused_kernel<NHT, NZT, DEPTH, W, R, 2>;
    }
    return nullptr;
}

extern "C" step_fn_t iaf_pick_step_fused(int nht, int nzt, int depth, int W, int R, int var, size_t* lds) {
    *lds = 0;
    if (nht == 10 && nzt == 2 && depth == 2) {
        if (W == 16 && R == 2) return inst<10, 2, 2, 16, 2>(var, lds);
        if (W == 8 && R == 1) return inst<10, 2, 2, 8, 1>(var, lds);
        if (W == 8 && R == 2) return inst<10, 2, 2, 8, 2>(var, lds);
        if (W == 4 && R == 4) return inst<10, 2, 2, 4, 4>(var, lds);
    }
    if (nht == 4 && nzt == 2 && depth == 1) {
        if (W == 16 && R == 2) return inst<4, 2, 1, 16, 2>(var, lds);
        if (W == 8 && R == 1) return inst<4, 2, 1, 8, 1>(var, lds);
        if (W == 8 && R == 2) return inst<4, 2, 1, 8, 2>(var, lds);
        if (W == 4 && R == 4) return inst<4, 2, 1, 4, 4>(var, lds);
    }
    return nullptr;
}
