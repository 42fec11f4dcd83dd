// iaf_step_fused.hpp -- one IAF step (tf_train.py:69-72: ar_multiconv2d + affine transform + log-det term; with the
// posterior sample in front and the KL elements behind it: tf_train.py:56-75) as ONE kernel launch.
//
// Why: at BASELINE batch sizes a step is depth_ar + 1 dependent launches of 6-18 us each, every one a latency chain
// (descriptor fetch, first loads, tile staging, split-K exchange, epilogue, the gap to the next launch) that is longer
// than its arithmetic (docs/LAB_NOTEBOOK_r01-r03.md 5).  The masked convs only look right and below -- taps (0,0) (0,1) (1,-1) (1,0) (1,1)
// -- so a workgroup that owns R full-width output rows of one image needs, per layer down the stack, ONE more row of the
// layer below and no columns beyond the image: it can compute all layers for its rows by itself, with the hidden
// activations in LDS and no synchronisation with any other workgroup.  Rows needed (depth_ar = D):
//     z: R+D+1   h_0: R+D   h_1: R+D-1  ...  h_{D-1}: R+1   output: R
// The halo rows are recomputed by the neighbouring workgroup (R = 2, D = 2: the first layer is computed twice, the second
// 1.5 times); what is bought with that is one launch per step instead of three, no HBM round trip of the hidden
// activations, no split-K exchange, and weight / context loads of the next phase in flight during the current one.
// XCH = 1 (16-pixel rows): the halo rows are EXCHANGED instead -- every hidden layer computes the R rows its workgroup owns, its
// region holds one row more, and that row is the first row of the block below, handed over through a per-stack, per-stream buffer
// in device memory.  Round 4's form (round 3's: docs/LAB_NOTEBOOK_r01-r03.md 4.9 item 7):
//   order      a workgroup takes a TICKET and only waits for lower tickets -- nothing rests on dispatch order or placement (xch_take_*);
//   hand-over  the data is the flag: rows are all-ones between launches, copied LDS -> memory -> LDS in 16-byte coalesced sc1
//              accesses, re-armed by their consumer (xch_export / xch_import);
//   who        four HELPER waves per workgroup (HLP), one beside each compute wave, do all of it in vector-memory queues of their
//              own while the compute waves multiply the taps of their own rows (StepPart) -- the helper section;
//   failure    bounded waits; a wait that gives up imports NaN and raises a sticky device word and a host-visible one.
//
// Arithmetic: the bf16x3 scheme of iaf_conv_bf3.hpp (three bf16 planes per operand, six products, fp32 accumulate) on
// v_mfma_f32_16x16x32_bf16, reading the same fragment-ordered weight packs (PrepLayer.wp3).
// LDS: three regions (z, h_even, h_odd) of pixel slots [row][W + 2 columns][plane h/m/l][channels] -- the two extra
// columns and the rows past the image bottom hold zeros, so tap addressing is pure arithmetic (no validity masks).
// Two waves per SIMD did not help: round 3 built an 8-wave variant (first layer split by pixel tiles, later layers and the
// output pair by K steps, partial sums handed over through the dead input region; bit-compatible, 233 GPU tests green) and
// measured 29.1 vs 27.4 us at 16-pixel rows, 18.1 vs 16.0 at 8 (same box, alternating runs).  Its stamps showed the older wave
// of a SIMD running its half of a K loop at the single-wave rate and the younger wave mostly after it, so the K loops took
// what they took, and issuing the prologue's loads, the barriers and the hand-over cost more.  Removed.  (What one wave of a
// SIMD can and cannot overlap, in wall-clock time: tools/probe/mfma_shadow.hip, profiles/r03/experiments/mfma_shadow.txt -- an
// unrolled MFMA stream of ONE wave runs at the pipe's rate, 7.3 ns per 16x16x32 MFMA; every VALU instruction between two
// MFMAs adds 1.1 - 1.6 ns, an LDS read 8 ns, an LDS write 13 ns; s_memtime ticks, the unit of the stamps below, change rate
// with the number of waves per SIMD and only compare within one configuration.)
// Work split: 4 waves, one per SIMD.  Hidden layers: wave w owns co tiles w, 7 - w, 8 + w, ... (serpentine over rounds of 4)
// of every pixel tile (its weight stream is disjoint from the other waves'; the activation fragments come from LDS; the
// serpentine evens out the dead centre-tap blocks a triangular layer skips, see TRI below); the tiles left over when the count is
// not a multiple of 4 (n_h = 160: tiles 8, 9) are dealt out per (tile, pixel tile) so that every wave multiplies the
// same number of units.  Output pair: 2 n_z / 16 tiles dealt out over the 4 waves (one each at n_z = 32), every wave runs the
// whole K range of its tiles; the results change from the MFMA layout to image rows through an exchange buffer.
// Ring depth: two steps of look-ahead for the hidden layers, three for the output pair.  Round 3 measured deeper rings sized
// by MFMAs per step (up to six steps at 8-pixel rows): the K loops gained 2 k cycles per workgroup and the bursts of preload
// instructions in front of the epilogues cost 4.4 k -- at 8-pixel rows the K loops are bound by the CU's vector-memory
// ISSUE rate (one 1 KiB wave-load per ~16 cycles = the 64 B/clk port: 750 wave-loads = 12 k of the second conv's 15 k cycles),
// not by fetch latency.
#pragma once
#include "iaf_conv_bf3.hpp"
#include "iaf_step_fused_types.hpp"
#ifndef IAF_HELPER_PRIO
#define IAF_HELPER_PRIO 0
#endif

#ifndef IAF_EXP_HOUT
#define IAF_EXP_HOUT 1
#endif
#ifndef IAF_EXP_HOUT8
#define IAF_EXP_HOUT8 1
#endif
// XCH = 1: neighbouring row blocks EXCHANGE their halo rows instead of recomputing them (see the kernel's header note): a
// hidden layer computes only the R rows its workgroup owns, its region holds one more row -- the first row of the block
// below, imported -- and z has R + 1 rows.
// PAIR = 1 (8-pixel rows; the header note of the kernel, "PAIR"): TWO workgroups per (image, row block), each streaming half of the last
// hidden layer's and of the output pair's weights; the regions are those of the recomputing form.
// F16 = 1 (round 6): the operands as TWO fp16 planes -- x = hi + lo' 2^-11, hi = fp16(x), lo' = fp16((x - hi) 2^11) -- and THREE
// part-products per K step on v_mfma_f32_16x16x32_f16: hi hi into the main accumulator, hi lo' + lo' hi into a cross accumulator that
// is scaled by 2^-11 where the sums meet (the kernel's header note, "f16x2").  Two planes per pixel slot and per weight fragment.
template <int NHT, int NZT, int DEPTH, int W, int R, int XCH = 0, int PAIR = 0, int F16 = 0>
struct StepGeom {
    static constexpr int NZ = 16 * NZT, NH = 16 * NHT;
    static constexpr int NPL = F16 ? 2 : 3;                    // planes per operand
    static constexpr int RS = W + 2;                           // slots per row (zero column on either side)
    static constexpr int Z8 = NZ / 8, Z16 = NPL * Z8 + 2;      // z slot: planes + pad, in 16-byte units (stride = 8 * odd dwords)
    static constexpr int H8 = NH / 8, H16 = NPL * H8 + 2;
    static constexpr int RZ = XCH ? R + 1 : R + DEPTH + 1;
    static constexpr int rows_h(int l) { return XCH ? R : R + DEPTH - l; }           // rows hidden layer l computes
    static constexpr int rows_reg(int l) { return XCH ? R + 1 : R + DEPTH - l; }     // rows its region holds
    static constexpr int ZREG = 0;                                               // region offsets in 16-byte units
    static constexpr int HREG0 = ZREG + RZ * RS * Z16;
    static constexpr int HREG1 = HREG0 + rows_reg(0) * RS * H16;
    static constexpr int END = HREG1 + (DEPTH >= 2 ? rows_reg(1) * RS * H16 : 0);
    static constexpr int XB_STRIDE = 2 * NZ + 1;                                 // output exchange buffer [pixel][2 n_z] floats
    // K parts of the output pair's sums: PAIR -- two waves per tile; HOUT (exchange form, n_z = 32, two hidden layers) -- the taps of the
    // row below split between a compute wave and its helper
    static constexpr bool HOUT = IAF_EXP_HOUT && XCH && !PAIR && NZT == 2 && DEPTH == 2;
    // HOUT8 (8-pixel rows, one row per workgroup, the kernels WITH helper waves): the whole output pair split the same way -- its K loop is
    // 6 MFMAs per step behind one wave's fetch latency; two waves per tile halve each wave's chain (the four-wave kernels of this geometry
    // size the buffer alike and use its first part only)
    static constexpr bool HOUT8 = IAF_EXP_HOUT8 && !XCH && !PAIR && NZT == 2 && DEPTH == 2 && W == 8 && R == 1;
    static constexpr int XKP = (PAIR || HOUT || HOUT8) ? 2 : 1;
    static constexpr size_t xb_bytes() { return (size_t)XKP * R * W * XB_STRIDE * 4; }
    // (the last hidden layer sits in the h_odd region for an even depth: z + h_even are dead then; for an odd depth it sits in
    // h_even: the buffer goes into h_odd if it fits there, else behind everything)
    static constexpr int XB_OFF = (DEPTH % 2 == 0) ? 0 : (DEPTH >= 3 && xb_bytes() <= (size_t)(END - HREG1) * 16) ? HREG1 : END;
    // context of the first epilogue, staged [channel][pixel of the h_0 rows] (row stride = 4 or 12 mod 16 floats: conflict-free for the epilogue's
    // lanes = 16 pixels x 4 channel groups): in the h_1 region, which the first conv does not touch; behind everything when
    // there is a single hidden layer
    static constexpr int CPX = rows_h(0) * W, CSTR = (CPX % 16 == 0 || CPX % 16 == 8) ? CPX + 4 : CPX;
    static constexpr int CTX_OFF = DEPTH >= 2 ? HREG1 : END;
    static constexpr size_t ctx_bytes() { return (size_t)NH * CSTR * 4; }
    static constexpr size_t lds_bytes() {
        size_t a = (size_t)END * 16;
        const size_t b = (size_t)XB_OFF * 16 + xb_bytes(), c = (size_t)CTX_OFF * 16 + ctx_bytes();
        a = a > b ? a : b;
        return a > c ? a : c;
    }
    // XCH: one exported row = W pixel slots of a hidden region, as stored in LDS
    static constexpr size_t xrow_bytes() { return (size_t)W * H16 * 16; }
    // PAIR: what one workgroup hands its partner = its channel half of the last hidden layer's region: rows x W slots x 3 planes x
    // H8 / 2 16-byte units
    static constexpr int PUS = 3 * (H8 / 2);                                     // units per pixel slot
    static constexpr int PUNITS = rows_reg(DEPTH - 1) * W * PUS;
    static constexpr size_t prow_bytes() { return (size_t)PUNITS * 16; }
};

// pixel tiles [lo, hi) of a phase with NPT tiles that group g of GN wave groups covers with its left-over co tile
constexpr int fused_extra_mask(int npt, int gn, int g) {
    const int lo = (g * npt + gn - 1) / gn, hi = ((g + 1) * npt + gn - 1) / gn;
    int m = 0;
    for (int q = lo; q < hi; ++q) m |= 1 << q;
    return m;
}

// Independent accumulators per (pixel tile, co tile) unit.  The six part-products of a step go to the SAME accumulator; a
// phase whose waves hold few units (the output pair: 2 units per wave at 16-pixel rows, 1 at 8) then issues chains of
// dependent MFMAs, which the matrix pipe does not overlap: measured 26-27 cycles per MFMA there against 19 in the hidden
// layers (2-3 co tiles per pixel tile: the same accumulator every second or third MFMA).  Such phases spread the six products
// over PSG accumulator groups (summed once, after the K loop), so that an accumulator is touched every third MFMA at most.
constexpr int fused_acc_groups(int tiles_per_pixel_tile) { return tiles_per_pixel_tile >= 3 ? 1 : tiles_per_pixel_tile == 2 ? 2 : 3; }

// The K steps of one conv_phase call: NT taps from tap T0 on, pair-major, every tile slot live; then -- TAIL -- the centre tap x
// pairs of a channel-triangular layer with its dead slots skipped (see TRI in the kernel).  Taps: 0 (0,0)  1 (0,+1)  2 (+1,-1)
// 3 (+1,0)  4 (+1,+1).  A whole layer is <0, 5, 0> (or <1, 4, 1> when triangular); the halo-exchange kernels run a layer
// that reads an imported row as two calls, <0, 2, 0> / <1, 1, 1> (the taps of the own rows) and <2, 3, 0> (the row below).
template <int T0_, int NT_, int TAIL_, int NPAIR_>
struct StepPart {
    static constexpr int ILV = 0;
    static constexpr int T0 = T0_, NT = NT_, TAIL = TAIL_, NPAIR = NPAIR_, NF = NPAIR_ * NT_, NSTEP = NF + (TAIL_ ? NPAIR_ : 0);
    __device__ static void at(int sq, int& pair, int& tap) {              // sequence index -> (input pair, tap)
        if (!TAIL_ || sq < NF) { pair = sq / NT_; tap = T0_ + sq - pair * NT_; }
        else { pair = sq - NF; tap = 0; }
    }
};

// The own-row taps of a channel-triangular layer in the exchange form -- tap (0,+1), every slot live, and the centre tap with its dead
// blocks skipped -- INTERLEAVED (round 5): full step (pair 0), centre step (pair NPAIR-1: the shortest), full (1), centre (NPAIR-2), ...
// A centre step multiplies as few as 6 MFMAs per wave; five of them in a row at the end of the part (the order until round 4) ran the
// 2-step ring dry -- its look-ahead covers the time the two steps in between take, and two short steps are ~400 cycles against a fetch
// latency of ~900: 37.8 cycles per MFMA in that part (profiles/r05/phase_table.md).  Between two full steps every gap is >= 36 MFMAs.
template <int NPAIR_>
struct StepPartIlv {
    static constexpr int ILV = 1;
    static constexpr int T0 = 1, NT = 1, TAIL = 1, NPAIR = NPAIR_, NF = 0, NSTEP = 2 * NPAIR_;
    static constexpr int centre_pair(int sq) { return NPAIR_ - 1 - (sq >> 1); }
    __device__ static void at(int sq, int& pair, int& tap) {
        if (sq & 1) { pair = NPAIR_ - 1 - (sq >> 1); tap = 0; } else { pair = sq >> 1; tap = 1; }
    }
};

constexpr int fused_popcount(int m) { int n = 0; for (; m; m &= m - 1) ++n; return n; }
// fragments [lo, hi) (3 per tile slot) of a ring refill that belong to live slots
constexpr int fused_live_frags(int live, int lo, int hi, int npl = 3) {
    int n = 0;
    for (int f = lo; f < hi; ++f) n += (live >> (f / npl)) & 1;
    return n;
}
// Tile slots of a hidden layer with a live centre-tap block at input pair c of a channel-triangular layer, for the waves of
// group gi of gn (the waves that run one instantiation of a hidden phase): slot j < nfull holds tile 4 j + w of wave w
// (serpentine: 4 (j + 1) - 1 - w for odd j), the left-over slot a tile >= 4 nfull; tile t is live at c iff c <= t / 2
// (t0: PAIR -- the wave's tiles are tiles t0 .. t0 + nht - 1 of the layer; its left-over slot holds tile t0 + 4 nfull)
constexpr int tri_live(int ntw, int nfull, int nht, int c, int gn, int gi, int t0 = 0) {
    int m = 0;
    for (int j = 0; j < ntw; ++j)
        for (int w = gi * (4 / gn); w < (gi + 1) * (4 / gn); ++w) {
            const int t = j >= nfull ? nht - 1 : (j & 1) ? 4 * (j + 1) - 1 - w : 4 * j + w;
            if (c <= (t0 + t) / 2) m |= 1 << j;
        }
    return m;
}
// HOUT: the taps of the row below (2, 3, 4) x NPAIR input pairs, every second step (parity PAR): a compute wave takes the even steps of
// its tiles, its helper wave -- idle once the row is in LDS -- the odd ones; the two sums meet in the exchange buffer
template <int NPAIR_, int PAR_>
struct BelowPar {
    static constexpr int ILV = 0;
    static constexpr int T0 = 2, NT = 3, TAIL = 0, NPAIR = NPAIR_, NSTEP = (3 * NPAIR_ + 1 - PAR_) / 2, NF = NSTEP;
    __device__ static void at(int sq, int& pair, int& tap) {
        const int q = 2 * sq + PAR_;
        pair = q / 3; tap = 2 + q - 3 * pair;
    }
};
// PAIR: the output pair's K steps of one of the two waves that share a tile -- every second step (parity PAR) of NP input pairs x 5
// taps, pair-major; `pair` is relative to the part's first input pair (conv_phase's pair0)
template <int NP_, int PAR_>
struct PairPart {
    static constexpr int ILV = 0;
    static constexpr int T0 = 0, NT = NTAPS, TAIL = 0, NPAIR = NP_, NSTEP = (NP_ * NTAPS + 1 - PAR_) / 2, NF = NSTEP;
    __device__ static void at(int sq, int& pair, int& tap) {
        const int q = 2 * sq + PAR_;
        pair = q / NTAPS; tap = q - pair * NTAPS;
    }
};

// HLP: helper waves (512 threads, 256 registers per wave; see the kernel).  Every exchange-form kernel has them (XCH implies HLP); the
// recomputing kernels of the BASELINE run's 8-pixel geometry exist in both forms: with helpers for the posterior block (their last
// workgroup does the block's free-bits reductions, StepP::fin_*: 24.0 -> 20.9 us), without for the bare IAF step (same-box rocprof
// averages of the 8x8 step: 16.62 us without, 16.89 us with -- profiles/r04/experiments/ab_round3_tree_vs_head_same_box.txt).
// The launch code reads the thread count off the kernel (hipFuncGetAttributes).
// VAR: the statement of the operator -- 0 TF, 1 Theano (image rotated by 180 degrees + border channel), 2 Theano with
// flipmask=True (TF geometry + border channel).  Compile time: as run-time flags these cost the TF path ~1 us per launch
// (branches around the border loads split the epilogue's basic blocks).
// PAIR (8-pixel rows, R = 2; round 5): at BASELINE batch sizes an 8x8 level gives a CU 8 pixels, and the step is bound by the 1.23 MB of
// weight fragments every workgroup pulls through its CU's 64 B/clk port (8.7 us; profiles/r04/fused_step_stamps.txt: the second
// conv's K loop = 900 wave-loads x 16 cycles).  Here TWO workgroups share one (image, row block of R = 2 rows = 16 pixels: full MFMA
// tiles): both compute the first hidden layer in full (its pack is 12 % of the bytes), each computes HALF of the last hidden layer's
// output channels and half of the output pair (z channels) -- 0.65 MB per CU -- and they hand each other their half of the last hidden
// region (R + 1 rows, 11.5 KB) through device memory with the machinery of XCH: tickets (partners hold adjacent tickets of one list, so
// at most one workgroup per list ever waits for a partner that is not running yet), the data as the flag, helper waves.  The output
// pair multiplies the input channels of its own half first -- they are in LDS already -- and the partner's behind the import.
template <int NHT, int NZT, int DEPTH, int W, int R, int VAR = 0, int XCH = 0, int HLP = XCH, int PAIR = 0, int F16 = 0>
__global__ __launch_bounds__(HLP ? 512 : 256)
__attribute__((amdgpu_waves_per_eu(HLP ? 2 : 1, HLP ? 2 : 1))) void iaf_step_fused_kernel(StepP p) {
    static_assert(HLP || !XCH, "the exchange form runs with helper waves");
    static_assert(!PAIR || (HLP && !XCH && DEPTH == 2 && NZT == 2 && NHT % 2 == 0 && (NHT / 2) % 4 == 1 && (16 * NHT / 8) % 2 == 0),
                  "PAIR: recomputing form with helpers, two hidden layers, n_z = 32, co tiles per half = 4 k + 1");
    constexpr bool HELP = HLP != 0;
    constexpr bool TKT = XCH != 0 || PAIR != 0;                  // the workgroup's item comes from a ticket (xch_take_*)
    static_assert(!F16 || !PAIR, "the pair form exists for the bf16x3 planes only");
    typedef StepGeom<NHT, NZT, DEPTH, W, R, XCH, PAIR, F16> G;
    constexpr int NPL = G::NPL;                                  // planes per operand: 3 bf16 (six part-products) or 2 fp16 (three)
    // exchanged rows and flags: AGENT scope (sc1: coherent across the XCDs, served by the memory side)
    constexpr int XSCOPE = __HIP_MEMORY_SCOPE_AGENT;
    constexpr bool FLIP = (VAR == 1), BORDER = (VAR != 0);
    static_assert(DEPTH >= 1 && DEPTH <= 4, "hidden layers ping-pong between two LDS regions (h_even, h_odd)");
    static_assert((W & (W - 1)) == 0 && W <= 16, "full-width rows of 4, 8 or 16 pixels");
    // a sweep of a converged inverse (iaf_step_inverse: the sweeps of one call are queued without the host in the loop; the word is
    // written by the residual check between two sweeps, never during one, so every workgroup of a launch reads the same value)
    if (p.mode == MODE_INVERSE && p.skip && *(const volatile unsigned*)p.skip) return;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    char* smem = (char*)smem4;
    constexpr int NZ = G::NZ, NH = G::NH, RS = G::RS, Z8 = G::Z8, Z16 = G::Z16, H8 = G::H8, H16 = G::H16, RZ = G::RZ;
    constexpr int NW = 4, NW_COMPUTE = 4;                        // (compute) waves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCH: four more waves, the HELPERS, one beside every compute wave (hence 256 registers per wave: two per SIMD).  They own the
    // workgroup's long-latency traffic -- the rows handed over between row blocks and the context rows of the first epilogue
    // (the helper section below) -- in vector-memory queues of their own: a compute wave's queue returns in order and feeds its K
    // loop, so a 1 us agent-scope load in it stalls its MFMAs (round 4, four waves: 8 k of 51.6 k cycles per workgroup were
    // imports).  One helper wave for all of it was too slow (16 KiB of sc1 loads from ONE wave: 6 k cycles; from four: 2.2 k).
    const bool is_helper = HELP && wave >= NW_COMPUTE;
    const int htid = tid - 64 * NW_COMPUTE;                      // helpers: 0 .. 255
    const int pl = lane & 15, kk = lane >> 4;
    // Which (image b, row block rbk) this workgroup computes.  Recomputing kernels: blockIdx, statically -- no workgroup depends on
    // another.  XCH: a workgroup waits for the rows of the block BELOW it, and HIP promises no dispatch order, so the order is one
    // the kernel creates (xch_take_item below): a workgroup takes a TICKET from a work list whose items are dealt out bottom row
    // block first, and therefore only ever waits for the holder of a LOWER ticket of the same list -- a workgroup that is already
    // running and, by induction, never waits for anything not running.  Any grid size, any dispatch order, any placement.
    int b = blockIdx.x / p.nrb, rbk = blockIdx.x - b * p.nrb, r0 = rbk * R;
    const int H = p.H, HW = p.HW;
    // pixel offset inside the image of position (image row ir, column col) of the space the kernel computes in
    // ((H-1-ir) W + (W-1-col) = HW-1 - (ir W + col))
    auto gpix = [&](int ir, int col) -> int { return FLIP ? (HW - 1 - (ir * W + col)) : (ir * W + col); };
    // Global addresses = a UNIFORM base (this workgroup's image: 64-bit scalar arithmetic, once) + a 32-bit per-lane BYTE offset
    // inside the image: the loads / stores then take the SGPR-base + VGPR-offset form.  Written with size_t element indices
    // the same accesses cost two to four 64-bit VALU instructions each -- round 3's stamps: 4.4 k cycles to ISSUE the
    // prologue's ~50 loads per wave, 130 v_lshl_add_u64 among ~800 prologue instructions.
    auto ldf = [](const float* base, unsigned boff) -> float { return *(const float*)((const char*)base + boff); };
    auto ldf4 = [](const float* base, unsigned boff) -> f32x4 { return *(const f32x4*)((const char*)base + boff); };
    auto stf = [](float* base, unsigned boff, float v) { *(float*)((char*)base + boff) = v; };
    size_t img_z = (size_t)b * NZ * HW, img_h = (size_t)b * NH * HW;            // element offsets of image b (z-shaped / context-shaped tensors)
    // sum of the border-indicator weights of the taps that leave the image at (ir, col): taps (0,1) (1,-1) (1,0) (1,1)
    auto border_terms = [&](const float* bt, int cstride, int ch, int ir, int col) -> f32x4 {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool last_row = ir == H - 1, c0 = col == 0, cl = col == W - 1;
        if (cl) a += *(const f32x4*)(bt + 0 * cstride + ch);
        if (last_row || c0) a += *(const f32x4*)(bt + 1 * cstride + ch);
        if (last_row) a += *(const f32x4*)(bt + 2 * cstride + ch);
        if (last_row || cl) a += *(const f32x4*)(bt + 3 * cstride + ch);
        return a;
    };
    // one f32x4 (4 consecutive channels of a pixel slot) -> the planes of an LDS region; F16: the largest magnitude this lane has split
    // (NaNs pass fmaxf by: a NaN in the inputs is the caller's, not a range failure)
    [[maybe_unused]] float rngmax = 0.f;
    auto split_store4 = [&](char* region, int slot, int q, f32x4 v, int s16, int c8) __attribute__((always_inline)) {
        if constexpr (F16) {
            rngmax = fmaxf(fmaxf(rngmax, fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
            f16s_store4(region, slot, q, v, s16, c8);
        } else {
            bf3_store4(region, slot, q, v, s16, c8);
        }
    };
    // F16: an operand beyond fp16's largest finite number went into the planes (this launch's outputs carry inf / NaN): say so where the
    // host reads it at its next call on the stack (StepP::rng_err), which then goes back to the bf16x3 kernels
    auto raise_range = [&]() __attribute__((always_inline)) {
        if constexpr (F16) {
            if (__any(rngmax > IAF_F16_MAX) && lane == 0 && p.rng_err)
                __hip_atomic_fetch_or(p.rng_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
#define IAF_FSTAMP(k) do { if (p.dbg && tid == 0) p.dbg[(size_t)blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
    IAF_FSTAMP(0);

    // ---- co tiles of a hidden layer per wave: NFULL rounds of 4 + (NX left-over tiles shared by groups of GN waves) ------
    constexpr int NFULL = NHT / NW, NX = NHT % NW;
    constexpr bool XSPLIT = NX > 0 && NW % NX == 0;              // left-over tiles dealt out per (tile, pixel tile)
    constexpr int GN = XSPLIT ? NW / NX : 1;                     // waves sharing one left-over tile
    constexpr int NTWH = NFULL + (NX ? 1 : 0);                   // tile slots per wave
    constexpr int NTWO = 2 * NZT / NW;                           // output pair: tiles per wave
    static_assert((2 * NZT) % NW == 0, "output tiles must split evenly over the waves");
    // HLEFT (round 5): in the second hidden layer of a two-layer stack the left-over (tile, pixel tile) units -- one per compute wave: 0.5 of
    // its 2.5 units at n_h = 160 -- are computed by the HELPER waves, one each, K loop and epilogue: a compute wave's K loops run at what one
    // wave per SIMD can issue (~22 cycles per MFMA, profiles/r05/phase_table.md), its helper is idle from the moment the row below is in LDS,
    // and the unit needs no partial sum from anybody (the helper runs all of the unit's taps).  The compute waves then hold NFULL slots.
#ifndef IAF_EXP_HLEFT
#define IAF_EXP_HLEFT 1
#endif
    constexpr bool HLEFT = IAF_EXP_HLEFT && HELP && !PAIR && DEPTH == 2 && XSPLIT && NX > 0 && NFULL > 0 &&
                           (NHT - NX) / 2 >= NH / 32 - 1;     // (the left-over tiles' centre-tap blocks are all live: the helper walks the plain step order)
    constexpr int NTW1 = HLEFT ? NFULL : NTWH;                   // slots of a compute wave in hidden layers l >= 1
    // HL0: the same for the FIRST hidden layer -- a helper computes its unit between issuing the context loads and staging what they
    // brought (5 K steps, all fragments requested at once), and runs the unit's epilogue behind the barrier that makes the context visible
#ifndef IAF_EXP_HL0
#define IAF_EXP_HL0 1
#endif
    // (not on the fp16 planes: there the compute waves' first K loop is half as long and the helper's unit sits on the critical path to
    //  the first epilogue's barrier -- 8x8 step 13.25 -> 12.87 us without it, 16x16 equal, profiles/r06/experiments/ab_helper_knobs_f16_same_box.txt)
    constexpr bool HL0 = IAF_EXP_HL0 && HLEFT && !F16;
    constexpr int NTW0 = HL0 ? NFULL : NTWH;
    int htile[NTWH];
#pragma unroll
    for (int j = 0; j < NTWH; ++j)      // (serpentine over the rounds: the dead centre-tap blocks of a wave's tiles add up evenly, see TRI)
        htile[j] = (j < NFULL) ? ((j & 1) ? NW * (j + 1) - 1 - wave : wave + NW * j) : (XSPLIT ? NW * NFULL + wave % NX : wave + NW * j);
    const int xg = XSPLIT ? wave / NX : 0;                       // this wave's group for the left-over tile

    // ---- weight fragments: [step = pair * 5 + tap][co tile][plane][lane][8 bf16] -> ring of U step slots per phase ----
    constexpr int NPT0 = (G::rows_h(0) * W + 15) / 16;
    constexpr int NPTO = (R * W + 15) / 16;
    // look-ahead per phase: first hidden layer, the other hidden layers, output pair (see the header comment)
#ifndef IAF_EXP_RDO_XCH
#define IAF_EXP_RDO_XCH 3
#endif
    // (IAF_EXP_RDO_XCH: dev knob -- the output pair's look-ahead in the exchange-form TF kernels, which have the registers for more)
    // F16: a K step carries half the MFMA time of a bf16x3 step (12 MFMAs = 190 cycles at 16-pixel rows, 6 at 8), so the same fetch latency
    // (~900 cycles from L2) wants a deeper look-ahead -- and the two-plane rings have the registers for it (profiles/r06/experiments/f16x2_ring_depth.txt)
#ifndef IAF_F16_RD0
#define IAF_F16_RD0 2
#endif
#ifndef IAF_F16_RDH
#define IAF_F16_RDH 2
#endif
#ifndef IAF_F16_RDO
#define IAF_F16_RDO 3
#endif
    constexpr int RD0 = F16 ? IAF_F16_RD0 : 2, RDH = F16 ? IAF_F16_RDH : 2,
                  RDO = F16 ? IAF_F16_RDO : (XCH && VAR == 0 && NZT == 2) ? IAF_EXP_RDO_XCH : (XCH && NZT == 4) ? 2 : 3;
    // (XCH && NZT == 4, config 3's depth-4 stacks: two waves per SIMD = 256 registers, and with a 3-step look-ahead of the output pair's
    //  ring -- 2 tiles x 3 planes x 4 slots = 96 registers -- all nine instantiations spilled 13-27 VGPRs inside their MFMA regions
    //  (VERDICT r05 weak #7); with 2 steps none does: tests/test_build_resources.py)
    constexpr int UA = (RD0 > RDH || DEPTH < 3 ? RD0 : RDH) + 1;     // slots of ring array A (layers 0, 2): layer 0, and layer 2 if any
    constexpr int UB = RDH + 1;                                      // ring array B (layers 1, 3)
    constexpr int UO = RDO + 1;
    // fragments [LO, HI) of step s (steps past s_end reload its last step; tiles past the layer's are clamped)
    // (wbase: the layer's pack, UNIFORM; step, tile and plane are wave-uniform too: the whole fragment address is scalar
    // arithmetic, the lane contributes its 16-byte slot as the load's 32-bit offset)
    const unsigned lane16 = 16u * (unsigned)lane;
    // TRI (hidden layer l >= 1 of the TF statement): its centre tap is the channel-triangular MADE mask itself (layers.py:115-124
    // with k = 1: input i reaches output o iff i <= o), so the block (input pair c = 32 channels, co tile t = 16 channels) of that
    // tap is all zeros -- stored zeros in the pack -- whenever 32 c > 16 t + 15, i.e. c > t / 2: 20 of the 50 blocks at n_h = 160.
    // Such a layer walks its steps in the order (four full taps x pairs), then (centre tap x pairs), and in a centre-tap step
    // neither loads nor multiplies a tile SLOT whose tiles are dead at that pair for every wave that runs this instantiation of
    // the phase (the wave groups of the left-over tiles: waves {0, 1} and {2, 3} at n_h = 160) -- known at compile time
    // (tri_live).  With the serpentine tile order a wave of either group skips 5 of its 62.5 (tile, step) units: all 20 dead
    // blocks, 8 % of the layer's MFMAs and weight bytes.  (Per-wave masks at RUN time -- wave-uniform branches around loads and
    // MFMAs in the K loop -- were measured first: 57.6 k instead of 23.3 k cycles for the second conv's K loop.)
    constexpr int NPAIR_H = NH / 32;
    auto ring_load = [&](auto lo_c, auto hi_c, f32x4 (*dst)[3], const f32x4* wbase, int ncot, const int* tiles, auto part_c, int s,
                         auto live_c, int pair0) __attribute__((always_inline)) {
        typedef decltype(part_c) P;
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value, LIVE = decltype(live_c)::value;
        int pair, tap;
        P::at(s < P::NSTEP ? s : P::NSTEP - 1, pair, tap);
        const f32x4* q = wbase + (size_t)((pair + pair0) * NTAPS + tap) * ncot * NPL * 64;
#pragma unroll
        for (int f = LO; f < HI; ++f) {
            const int j = f / NPL, pn = f - NPL * j;
            if (!((LIVE >> j) & 1)) continue;                              // (compile time) a dead slot of the step being fetched
            const int tc = tiles[j] < ncot ? tiles[j] : ncot - 1;
            dst[j][pn] = *(const f32x4*)((const char*)(q + ((size_t)tc * NPL + pn) * 64) + lane16);
        }
    };
    typedef StepPart<0, NTAPS, 0, NZ / 32> PartL0;                          // the first layer (input z)
    typedef StepPart<0, NTAPS, 0, NPAIR_H> PartFull;                        // a whole layer over n_h input channels, plain order
    typedef StepPart<1, NTAPS - 1, 1, NPAIR_H> PartTri;                     // ... channel-triangular: full taps, then the centre tap
    typedef StepPart<0, 2, 0, NPAIR_H> PartOwn;                             // XCH: the taps that read the workgroup's own rows
    typedef StepPartIlv<NPAIR_H> PartOwnTri;                                // ... of a triangular layer (full and centre steps interleaved)
    typedef StepPart<2, 3, 0, NPAIR_H> PartBelow;                           // XCH: the taps that read the row below (the imported one)
    typedef std::conditional_t<VAR == 0, PartTri, PartFull> PartHid;        // hidden layers l >= 1
    constexpr std::integral_constant<int, -1> ALL{};                       // every slot live

    // ---- prologue: z rows first (the first conv cannot start without them), then the first weight steps; zero columns
    // while both travel; z -> LDS.  XCH: the weight steps and the zero columns do not depend on which rows the workgroup
    // computes -- they cover the travel time of its ticket, and the z rows follow -------------------------------------------
    constexpr int NPX = RZ * W, NIT = NPX * (NZ / 4), ZU = (NIT + 255) / 256;
    f32x4 zv[ZU], zq[4][ZU];     // posterior input: the five tensors as raw loads, combined once all of them are on their way
    auto load_z = [&]() {
#pragma unroll
        for (int u = 0; u < ZU; ++u) {
            const int idx = tid + u * 256;
            const int ic = idx < NIT ? idx : NIT - 1;
            const int q = ic / NPX, px = ic - q * NPX;                 // pixel fastest: coalesced along a row
            const int row = px / W, col = px - row * W;
            const int rr = r0 + row < H ? r0 + row : H - 1;            // rows past the image: a valid address, zeroed below
            const unsigned gb = 4u * (unsigned)(4 * q * HW + gpix(rr, col));        // byte offset inside image b
            if (p.z) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zv[u][r] = ldf(p.z + img_z, gb + 4u * (unsigned)(r * HW));
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned i = gb + 4u * (unsigned)(r * HW);
                    zv[u][r] = ldf(p.qm + img_z, i); zq[0][u][r] = ldf(p.rm + img_z, i); zq[1][u][r] = ldf(p.ql + img_z, i);
                    zq[2][u][r] = ldf(p.rl + img_z, i); zq[3][u][r] = ldf(p.eps + img_z, i);
                }
            }
        }
    };
    f32x4 wr0[UA][NTWH][3];
    f32x4 wrh0[HL0 ? UA : 1][NTW0][3];       // HL0: the first layer's ring of NFULL slots (wr0 is then unused)
    const f32x4* wb0 = (const f32x4*)p.wp3[0];
    auto preload_w0 = [&]() __attribute__((always_inline)) {
        static_for<RD0>([&](auto i) {
            if constexpr (HL0)
                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTW0 * NPL>{}, wrh0[decltype(i)::value], wb0, NHT, htile,
                          PartL0{}, decltype(i)::value, ALL, 0);
            else
                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWH * NPL>{}, wr0[decltype(i)::value], wb0, NHT, htile,
                          PartL0{}, decltype(i)::value, ALL, 0);
        });
    };
    // context rows of this workgroup: per channel one contiguous run of CPX pixels (full-width rows) -> 16-byte loads,
    // 4 channels per wave instruction; summed with the second context here
    constexpr int CPX = G::CPX, CSTR = G::CSTR, CG = CPX / 4, NCIT = NH * CG, NCI = (NCIT + 255) / 256;
    f32x4 cv[NCI], cv2[NCI];     // raw loads: nothing consumes them before the z rows are in LDS.  UNCONDITIONAL loads from
    unsigned cvalid = 0;         // clamped addresses: `x = 0; if (inside) x = load` is a select on the loaded value, i.e. a
    auto load_ctx = [&]() {      // vmcnt(0) behind every load -- ten serial HBM round trips in this prologue (found in the
                                 // round-3 ISA: 7 k cycles of prologue at 16-pixel rows).  Rows past the image bottom are zeroed
                                 // when the values are used (store_ctx), from the bit mask.  (The host never launches this
                                 // kernel without a context: depth_ar >= 1.)
        const int vpx = (H - r0) * W < CPX ? (H - r0) * W : CPX;       // pixels of those rows that lie inside the image
        // item idx = tid + 256 u -> (channel c = idx / CG, pixel group g4).  Where 256 is a multiple of CG and the items
        // fill the rounds exactly (16-pixel rows), every round reads the SAME pixel group of a channel 256 / CG further on:
        // one address computation, then a constant stride per round (the general form costs ~22 VALU instructions per load)
        constexpr bool REG = (256 % CG == 0) && (NCIT % 256 == 0);
        unsigned gi0 = 0, gstep = 0;
        bool in0 = true;
        if constexpr (REG) {
            const int c = tid / CG, g4 = (tid - c * CG) * 4;
            in0 = g4 < vpx;
            const int g4c = in0 ? g4 : 0;
            const int row = g4c / W, col4 = g4c - row * W;
            gi0 = 4u * (unsigned)(c * HW + gpix(r0 + row, col4 + (FLIP ? 3 : 0)));
            gstep = 4u * (unsigned)((256 / CG) * HW);
        }
        auto ctx_off = [&](int u, bool& inside) -> unsigned {
            if constexpr (REG) { inside = in0; return gi0 + (unsigned)u * gstep; }
            const int idx = tid + u * 256;
            const int ic = idx < NCIT ? idx : NCIT - 1;
            const int c = ic / CG, g4 = (ic - c * CG) * 4;
            inside = g4 < vpx;
            const int g4c = inside ? g4 : 0;                           // a valid address either way
            const int row = g4c / W, col4 = g4c - row * W;             // 4 consecutive columns of one row
            return 4u * (unsigned)(c * HW + gpix(r0 + row, col4 + (FLIP ? 3 : 0)));
        };
#pragma unroll
        for (int u = 0; u < NCI; ++u) {
            bool inside;
            const unsigned gi = ctx_off(u, inside);
            cv[u] = ldf4(p.ctx + img_h, gi);               // (column order fixed up at store time)
            cvalid |= (inside ? 1u : 0u) << u;
        }
        if (p.ctx2) {                                      // (uniform branch: no select on loaded values)
#pragma unroll
            for (int u = 0; u < NCI; ++u) {
                bool inside;
                const unsigned gi = ctx_off(u, inside);
                cv2[u] = ldf4(p.ctx2 + img_h, gi);
            }
        }
    };
    auto zero_cols = [&]() {
        constexpr int ZROWS = RZ, H0ROWS = G::rows_reg(0), H1ROWS = 0;    // (h_1's zero columns: after the first layer)
        for (int i = tid; i < ZROWS * 2 * Z16; i += 256) {
            const int rs = i / Z16, u = i - rs * Z16;
            smem4[G::ZREG + ((rs >> 1) * RS + (rs & 1) * (W + 1)) * Z16 + u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int i = tid; i < (H0ROWS + H1ROWS) * 2 * H16; i += 256) {
            const int rs = i / H16, u = i - rs * H16;
            const int row = rs >> 1;
            const int base = row < H0ROWS ? G::HREG0 + row * RS * H16 : G::HREG1 + (row - H0ROWS) * RS * H16;
            smem4[base + (rs & 1) * (W + 1) * H16 + u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_z = [&]() {
#pragma unroll
        for (int u = 0; u < ZU; ++u) {
            const int idx = tid + u * 256;
            if (idx < NIT) {
                const int q = idx / NPX, px = idx - q * NPX;
                const int row = px / W, col = px - row * W;
                f32x4 v = zv[u];
                if (!p.z) {      // z0 = (qm+rm) + exp(ql+rl) * eps   (tf_train.py:57,63; distributions.py:21)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = (zv[u][r] + zq[0][u][r]) + __expf(0.5f * (2.f * (zq[1][u][r] + zq[2][u][r]))) * zq[3][u][r];
                }
                if (r0 + row >= H) v = f32x4{0.f, 0.f, 0.f, 0.f};
                split_store4(smem + (size_t)G::ZREG * 16, row * RS + col + 1, q, v, Z16, Z8);
            }
        }
    };

    // ---- XCH: which rows this workgroup computes -- a ticket, not blockIdx -----------------------------------------------------
    // p.xctl (64-bit words; every counter in a 128-byte line of its own -- 512 read-modify-writes of ONE line took 12 k cycles of
    // this prologue when heads and arrivals shared one): [32 y] head of work list y = tickets taken; [32 y + 16] holders of list y
    // that are done with the heads; [IAF_XCTL_DONE] lists whose holders all are; [IAF_XCTL_STICKY] sticky error.
    // List y (y < 8) holds the images b = y, y + 8, y + 16, ... < B, their row blocks bottom first: ticket t of list y is (image
    // y + 8 (t / nrb), row block nrb - 1 - t % nrb).  A workgroup asks the list of the XCD it runs on first (HW_REG_XCC_ID) -- the
    // dispatcher deals workgroups round robin over the XCDs, so at B = 8 k every list has exactly as many pullers as items and nobody
    // steals -- and the other lists in turn when that one is dry (lists that hold nothing at B < 8 are never asked): grid = number of
    // items, so every workgroup finds exactly one.  The placement decides which head answers, never what is computed.  (A ticket costs
    // ~3.3 k cycles: the latency of one agent-scope read-modify-write with return, not contention -- four lists per XCD, picked by CU id,
    // changed nothing in the balanced case and cost up to 30 k cycles where the surplus workgroups of an unevenly used list had to steal:
    // profiles/r05/experiments/ticket_lists.txt.)
    // The last holder of a list to arrive counts its list at [IAF_XCTL_DONE]; the one that completes that count (every ticket of the
    // launch is taken by then) puts heads and arrival counts back to zero for the next launch on these buffers.
    constexpr unsigned XNLST = IAF_XCTL_LISTS;
    unsigned xcc = 0, xdead = 0;
    unsigned long long xdone = 0, xlists_done = 0;
    [[maybe_unused]] unsigned xmine = 0, xmine_n = 0;                    // (thread 0: the list its ticket came from, that list's items)
    int xslot = 0;
    int half = 0;                                                        // PAIR: which channel half this workgroup computes
    [[maybe_unused]] unsigned long long xt0 = 0;
    [[maybe_unused]] unsigned xlist = 0, xnl = 1;
    unsigned* xmail = (unsigned*)(smem + (size_t)G::CTX_OFF * 16);          // (the context staging area: not written before the first conv is done)
    auto xch_take_begin = [&]() {
        if constexpr (TKT) {
            if (tid == 0) {
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
                xcc &= 7u;
                xlist = xcc & (XNLST - 1u);                                               // one list per XCD
                if (p.xknob & 1u) xlist = (blockIdx.x * 2654435761u >> 13) & (XNLST - 1u);        // test knob: lists that ignore the placement
                xnl = (unsigned)p.B < XNLST ? (unsigned)p.B : XNLST;                           // lists that hold anything: y < min(B, 8)
                if (xnl < XNLST) xlist %= xnl;
                if (p.xknob & 2u)                                                       // test knob: tickets out of dispatch order
                    for (unsigned i = 0, n = (blockIdx.x * 40503u >> 4) & 63u; i < n; ++i) __builtin_amdgcn_s_sleep(64);
                xt0 = __hip_atomic_fetch_add(p.xctl + 32 * xlist, 1ull, __ATOMIC_RELAXED, XSCOPE);
                xdead = (unsigned)__hip_atomic_load(p.xctl + IAF_XCTL_STICKY, __ATOMIC_RELAXED, XSCOPE);
            }
        }
    };
    auto xch_take_finish = [&]() {
        if constexpr (TKT) {
            if (tid == 0) {
                unsigned long long v = xt0;
                unsigned y = xlist, t = 0, ny = 0;
                bool found = false;
                for (unsigned k = 0; k < xnl; ++k) {
                    if (k) { y = (xlist + k) % xnl; v = __hip_atomic_fetch_add(p.xctl + 32 * y, 1ull, __ATOMIC_RELAXED, XSCOPE); }
                    // (PAIR: two tickets per (image, row block) -- 2 k and 2 k + 1 are the two halves of item k)
                    ny = (int)y < p.B ? (PAIR ? 2u : 1u) * (unsigned)p.nrb * (unsigned)((p.B - (int)y + (int)XNLST - 1) / (int)XNLST) : 0u;
                    t = (unsigned)v;
                    if (t < ny) { found = true; break; }
                }
                // (the heads are final for this workgroup: count it -- the result is looked at after the first conv)
                xmine = y; xmine_n = found ? ny : 0u;
                if (found) xdone = __hip_atomic_fetch_add(p.xctl + 32 * y + 16, 1ull, __ATOMIC_RELAXED, XSCOPE);
                unsigned ib = 0, ir = (unsigned)p.nrb - 1;
                const unsigned ti = PAIR ? t >> 1 : t;
                if (found) { ib = y + XNLST * (ti / (unsigned)p.nrb); ir = (unsigned)p.nrb - 1u - ti % (unsigned)p.nrb; }
                else xdead |= 2u;                                           // (grid != B * nrb: a host bug -- loud, not a hang)
                xmail[0] = ib; xmail[1] = ir; xmail[2] = xdead; xmail[3] = t & 1u;
                if (p.dbg) p.dbg[(size_t)blockIdx.x * 32 + 30] = 1ull + (((unsigned long long)y << 32) | t);      // (dev tool: which ticket of which list)
            }
            __syncthreads();
            b = __builtin_amdgcn_readfirstlane((int)xmail[0]);
            rbk = __builtin_amdgcn_readfirstlane((int)xmail[1]);
            xdead = __builtin_amdgcn_readfirstlane(xmail[2]);
            if constexpr (PAIR) half = __builtin_amdgcn_readfirstlane((int)xmail[3]);
            r0 = rbk * R;
            img_z = (size_t)b * NZ * HW; img_h = (size_t)b * NH * HW;
            xslot = b * p.nrb + rbk;
            if (xdead && tid == 0) {
                if (p.xerr) __hip_atomic_store(p.xerr, 0x100u | xdead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(p.xctl + IAF_XCTL_STICKY, (unsigned long long)xdead, __ATOMIC_RELAXED, XSCOPE);
            }
        }
    };
    // ... and, once all of this launch's tickets are taken, the counters of the next launch: two steps, so that no wave ever waits
    // for a counter (each result is looked at a phase after its request)
    auto xch_next_epoch_a = [&]() {
        if constexpr (TKT) {
            if (tid == 0 && xmine_n && (unsigned)xdone == xmine_n - 1u)
                xlists_done = __hip_atomic_fetch_add(p.xctl + IAF_XCTL_DONE, 1ull, __ATOMIC_RELAXED, XSCOPE) + 1ull;
        }
    };
    auto xch_next_epoch_b = [&]() {
        if constexpr (TKT) {
            if (tid == 0 && (unsigned)xlists_done == (unsigned)(p.B < (int)XNLST ? p.B : (int)XNLST)) {
                for (int y = 0; y < (int)XNLST; ++y) {
                    __hip_atomic_store(p.xctl + 32 * y, 0ull, __ATOMIC_RELAXED, XSCOPE);
                    __hip_atomic_store(p.xctl + 32 * y + 16, 0ull, __ATOMIC_RELAXED, XSCOPE);
                }
                __hip_atomic_store(p.xctl + IAF_XCTL_DONE, 0ull, __ATOMIC_RELAXED, XSCOPE);
            }
        }
    };

    if constexpr (TKT) {
        xch_take_begin();
        if (!is_helper) {
            preload_w0();
            zero_cols();
        }
        xch_take_finish();
        if (!is_helper) {
            load_z();
            __builtin_amdgcn_sched_barrier(0);
            IAF_FSTAMP(8);
            stage_z();                                           // (the context rows: the helper stages them during the first conv)
        }
    } else if (!is_helper) {
        load_z();
        __builtin_amdgcn_sched_barrier(0);
        preload_w0();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!HELP) load_ctx();                         // (HELP: the helpers stage the context rows during the first conv)
        __builtin_amdgcn_sched_barrier(0);
        IAF_FSTAMP(8);
        zero_cols();
        stage_z();
    }
    IAF_FSTAMP(9);
    __syncthreads();
    IAF_FSTAMP(1);
    // the context rows go to LDS after the first conv's K loop (they had all of it to arrive), then a barrier, then its epilogue
    auto store_ctx = [&]() {
        float* creg = (float*)(smem + (size_t)G::CTX_OFF * 16);
#pragma unroll
        for (int u = 0; u < NCI; ++u) {
            const int idx = tid + u * 256;
            if (NCIT % 256 == 0 || idx < NCIT) {
                const int c = idx / CG, g4 = (idx - c * CG) * 4;
                constexpr bool fl = FLIP;                        // rotated image: the 4 columns arrived in reverse order
                f32x4 v = cv[u];
                if (p.ctx2) v += cv2[u];                         // up_context + down_context (tf_train.py:58)
                if (!((cvalid >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};      // rows past the image bottom
                *(f32x4*)(creg + c * CSTR + g4) = f32x4{fl ? v[3] : v[0], fl ? v[2] : v[1], fl ? v[1] : v[2], fl ? v[0] : v[3]};
            }
        }
    };

    // ---- one conv phase: acc[q][j] (+)= sum over the steps of `part` of W[step][tiles[j]] x X[pixel tile q, step] -------------
    // in_reg / in_s16 / in_c8: the input region (16-byte units); ROWS * W output pixels in NPT tiles.  The LAST tile slot is
    // multiplied only for the pixel tiles in EMASK (the wave's share of a left-over tile); all others for every pixel tile.
    // The caller has requested the part's first RD steps into ring slots 0 .. RD - 1.  accum_c: add to acc_out (second part).
    // (qbase: the first pixel tile a conv_phase / hidden_epilogue call works on -- 0 everywhere but in the helper waves' left-over units, HLEFT)
    int qbase = 0;
    auto conv_phase = [&](auto rd_c, auto npt_c, auto ntw_c, auto rows_c, auto emask_c, int in_reg, int in_s16, int in_c8, const f32x4* wbase,
                          int ncot, const int* tiles, f32x4 (*wr)[decltype(ntw_c)::value][3],
                          f32x4 (*acc_out)[decltype(ntw_c)::value], auto part_c, auto grp_c, auto accum_c, int pair0) __attribute__((always_inline)) {
        // pair0: the part's first input pair (PAIR's output parts; 0 everywhere else)
        typedef decltype(part_c) P;
        constexpr int NPT = decltype(npt_c)::value, NTW = decltype(ntw_c)::value, ROWS = decltype(rows_c)::value;
        constexpr bool TRI = P::TAIL != 0;                         // triangular centre tap: step order and dead slots, see StepPart
        constexpr int nstep = P::NSTEP, s0 = 0;
        constexpr int EMASK = decltype(emask_c)::value;
        // a wave group whose share of the left-over tile is EMPTY (fewer pixel tiles than groups: the second hidden layer at 8-pixel rows,
        // one pixel tile for two groups) never multiplies its last slot: it does not fetch it either.  (Until round 5 it did: 150 of the
        // 900 wave-loads of that K loop, which is bound by exactly those loads -- the CU's 64 B/clk port.)
        constexpr int LALLV = (EMASK == 0 && NTW > 1) ? ((1 << (NTW - 1)) - 1) : -1;
        constexpr std::integral_constant<int, LALLV> LALL{};
        constexpr int RD = decltype(rd_c)::value, U = RD + 1;      // this phase's look-ahead; it uses slots 0 .. RD of its ring array
        // (grp_c: the wave group; PAIR's half h, whose waves own tiles h NHT/2 .. of the layer, one group per wave: 100 (1 + h) + wave;
        //  300: a helper wave's left-over unit (HLEFT) -- every centre-tap block live, ONE accumulator group: the same sums in the same
        //  order as the compute wave that holds the tile in the kernels without helpers, bit for bit)
        constexpr int GRP = decltype(grp_c)::value;
        //  400 + group: a compute wave of an HLEFT kernel in hidden layer 1 -- NFULL slots, but the accumulator groups of NTWH slots, so that
        //  its sums too are those of the kernels without helpers)
        constexpr bool TALL = GRP >= 300 && GRP < 400, PTRI = GRP >= 100 && GRP < 300, HCMP = GRP >= 400;
        // F16: group 0 = the main products (hi hi), groups 1 .. = the cross products (hi lo' + lo' hi; two groups where a wave holds a single
        // tile per pixel tile, so that no MFMA waits for the one in front of it), scaled by 2^-11 where the sums meet
        constexpr int PSG = F16 ? (NTW >= 2 ? 2 : 3) : TALL ? 1 : fused_acc_groups(HCMP ? NTWH : NTW);   // accumulator groups of a unit's part-products
        f32x4 acc[PSG][NPT][NTW];
        int xb[NPT];
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            int pix = (q + qbase) * 16 + pl;
            pix = pix < ROWS * W ? pix : ROWS * W - 1;             // partially filled tile: a valid address, result unused
            const int row = pix / W, col = pix - row * W;
            xb[q] = in_reg + (row * RS + col + 1) * in_s16 + kk;
#pragma unroll
            for (int g = 0; g < PSG; ++g)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[g][q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto xaddr = [&](auto q_c, int s) __attribute__((always_inline)) -> int {
            int pair, tap;
            P::at(s < nstep ? s : nstep - 1, pair, tap);
            const int toff = tap < 2 ? tap : RS + tap - 3;          // slots: (0,0) (0,1) (1,-1) (1,0) (1,1)
            return xb[decltype(q_c)::value] + toff * in_s16 + (pair + pair0) * 4;
        };
        // x operand (three planes per pixel tile): read from LDS one pixel tile ahead of its MFMAs -- or, in phases whose
        // pixel tiles carry only 6-12 MFMAs (the output pair: less MFMA time than an LDS round trip), a whole STEP ahead:
        // all pixel tiles of step s + 1 are requested while step s multiplies (double buffer by ring-slot parity)
        constexpr bool XAHEAD = (NTW <= 2) && (U % 2 == 0);
        f32x4 xn[3];
        f32x4 xs[XAHEAD ? 2 : 1][XAHEAD ? NPT : 1][3];
        if constexpr (XAHEAD) {
            static_for<NPT>([&](auto q_c) {
                const int a = xaddr(q_c, s0);
                xs[0][decltype(q_c)::value][0] = smem4[a]; xs[0][decltype(q_c)::value][1] = smem4[a + in_c8];
                if constexpr (NPL == 3) xs[0][decltype(q_c)::value][2] = smem4[a + 2 * in_c8];
            });
        } else {
            const int a = xaddr(std::integral_constant<int, 0>{}, s0);
            xn[0] = smem4[a]; xn[1] = smem4[a + in_c8];
            if constexpr (NPL == 3) xn[2] = smem4[a + 2 * in_c8];
        }
        // live_c: tile slots multiplied in this step; next_c: those of step s + RD, whose weights this step requests (bit masks;
        // -1 = all, 0 = a step past the end: nothing to fetch)
        auto step_body = [&](auto slot_c, int s, auto live_c, auto next_c) __attribute__((always_inline)) {
            constexpr int I = decltype(slot_c)::value, LIVE = decltype(live_c)::value;
            static_for<NPT>([&](auto q_c) {
                constexpr int q = decltype(q_c)::value;
                constexpr int NTQ = ((EMASK >> q) & 1) ? NTW : NTW - 1;      // tile slots multiplied for this pixel tile
                // this pixel tile's share of the refill of the slot consumed RD steps from now
                constexpr int LO = (q * NTW * NPL) / NPT, HI = ((q + 1) * NTW * NPL) / NPT;
                ring_load(std::integral_constant<int, LO>{}, std::integral_constant<int, HI>{}, wr[(I + RD) % U], wbase, ncot, tiles, part_c,
                          s + RD, next_c, pair0);
                f32x4 xr[3];                                      // the pixel tile's x planes (raw 16 bytes per lane)
                if constexpr (XAHEAD) {
                    xr[0] = xs[I & 1][q][0]; xr[1] = xs[I & 1][q][1];
                    if constexpr (NPL == 3) xr[2] = xs[I & 1][q][2];
                    const int a = xaddr(q_c, s + 1);
                    xs[(I + 1) & 1][q][0] = smem4[a]; xs[(I + 1) & 1][q][1] = smem4[a + in_c8];
                    if constexpr (NPL == 3) xs[(I + 1) & 1][q][2] = smem4[a + 2 * in_c8];
                } else {
                    xr[0] = xn[0]; xr[1] = xn[1];
                    if constexpr (NPL == 3) xr[2] = xn[2];
                    const int a = (q + 1 < NPT) ? xaddr(std::integral_constant<int, (q + 1) % NPT>{}, s)
                                                : xaddr(std::integral_constant<int, 0>{}, s + 1);
                    xn[0] = smem4[a]; xn[1] = smem4[a + in_c8];
                    if constexpr (NPL == 3) xn[2] = smem4[a + 2 * in_c8];
                }
                constexpr int NLV = fused_popcount(LIVE & ((1 << NTQ) - 1));
                // part-product K of the step: weight plane WP x x plane XP into accumulator group G
#define IAF_FPROD(G, WP, XP)                                                                                       \
    static_for<NTQ>([&](auto j_c) {                                                                                \
        constexpr int j = decltype(j_c)::value;                                                                    \
        if constexpr ((LIVE >> j) & 1) {                                                                           \
            if constexpr (F16)                                                                                     \
                acc[(G)][q][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wr[I][j][WP]),   \
                                                                        __builtin_bit_cast(f16x8, xr[XP]), acc[(G)][q][j], 0, 0, 0); \
            else                                                                                                   \
                acc[(G)][q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[I][j][WP]), \
                                                                         __builtin_bit_cast(bf16x8, xr[XP]), acc[(G)][q][j], 0, 0, 0); \
        }                                                                                                          \
    });
                if constexpr (F16) {
                    IAF_FPROD(1, 0, 1)                            // hi x lo'
                    IAF_FPROD(0, 0, 0)                            // hi x hi   (between the two cross products: an accumulator every third MFMA at most)
                    IAF_FPROD(PSG - 1, 1, 0)                      // lo' x hi
                } else {
                    IAF_FPROD(0 % PSG, 2, 0)
                    IAF_FPROD(1 % PSG, 0, 2)
                    IAF_FPROD(2 % PSG, 1, 1)
                    IAF_FPROD(3 % PSG, 1, 0)
                    IAF_FPROD(4 % PSG, 0, 1)
                    IAF_FPROD(5 % PSG, 0, 0)
                }
#undef IAF_FPROD
                constexpr int NLD = fused_live_frags(decltype(next_c)::value, LO, HI, NPL);
                // (F16 at n_z = 64, n_h = 128: with the LDS reads pinned between the MFMAs as well the scheduler's pipeline stretches live ranges
                //  until 244 VGPRs spill; with the weight loads alone pinned none does)
                if constexpr (F16 && NZT == 4 && NHT == 8) sched_interleave<3 * NLV, 0, 0, NLD>();
                else sched_interleave<(F16 ? 3 : 6) * NLV, NPL, 0, NLD>();
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        int s = s0;                                                  // ring slot of step s: s % U
        constexpr int TGN = PTRI ? 4 : (XSPLIT ? GN : 1), TGI = GRP % 100;
        constexpr int TNF = PTRI ? (NHT / 2) / 4 : NFULL, TNH = PTRI ? NHT / 2 : NHT, TT0 = PTRI ? (GRP / 100 - 1) * (NHT / 2) : 0;
        // the pair-major body: every slot live (its look-ahead into the first centre-tap steps may fetch a dead block: unused)
        constexpr int NF = P::NF, MAIN = (NF / U) * U;
        if constexpr (P::ILV != 0) {
            // full and centre steps interleaved (StepPartIlv): step sq multiplies the slots live in it and requests those live in sq + RD
            static_for<P::NSTEP>([&](auto s_c) {
                constexpr int sq = decltype(s_c)::value, sn = sq + RD;
                constexpr int live = (sq & 1) ? (tri_live(NTW, TNF, TNH, P::centre_pair(sq), TGN, TGI, TT0) & LALLV) : LALLV;
                constexpr int next = sn >= P::NSTEP ? 0 : (sn & 1) ? (tri_live(NTW, TNF, TNH, P::centre_pair(sn), TGN, TGI, TT0) & LALLV) : LALLV;
                step_body(std::integral_constant<int, sq % U>{}, sq, std::integral_constant<int, live>{}, std::integral_constant<int, next>{});
            });
        } else {
        for (; s + U <= MAIN; s += U)
            static_for<U>([&](auto i) { step_body(i, s + decltype(i)::value, LALL, LALL); });
        static_for<NF - MAIN>([&](auto i) { step_body(i, MAIN + decltype(i)::value, LALL, LALL); });
        if constexpr (TRI) {
            // the centre tap: pair c multiplies the slots live at c and requests those live at c + RD
            static_for<P::NPAIR>([&](auto c_c) {
                constexpr int c = decltype(c_c)::value, sq = NF + c;
                constexpr int live = TALL ? LALLV : (tri_live(NTW, TNF, TNH, c, TGN, TGI, TT0) & LALLV);
                constexpr int next = (c + RD < P::NPAIR) ? (TALL ? LALLV : (tri_live(NTW, TNF, TNH, c + RD, TGN, TGI, TT0) & LALLV)) : 0;
                step_body(std::integral_constant<int, sq % U>{}, sq, std::integral_constant<int, live>{}, std::integral_constant<int, next>{});
            });
        }
        }
#pragma unroll
        for (int q = 0; q < NPT; ++q)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                f32x4 a = acc[0][q][j];
                if constexpr (F16) {
                    f32x4 c = acc[1][q][j];
                    if constexpr (PSG == 3) c += acc[2][q][j];
                    a += c * (1.0f / 2048.0f);                    // the cross products carry lo' = lo 2^11
                } else {
#pragma unroll
                    for (int g = 1; g < PSG; ++g) a += acc[g][q][j];
                }
                if constexpr (decltype(accum_c)::value) acc_out[q][j] += a; else acc_out[q][j] = a;
            }
    };
    constexpr std::integral_constant<bool, false> SET{};
    constexpr std::integral_constant<bool, true> ADD{};

    // hidden epilogue: bias (+ context) + ELU (layers.py:63-64,163-165) -> the three planes of the next region; rows past the
    // image bottom become the zero rows the layer above pads with
    // (ntw_c slots holding tiles tl[]: NTWH / htile everywhere but in PAIR's last hidden layer)
    auto load_bias = [&](auto ntw_c, const int* tl, const float* bias, f32x4* bi) __attribute__((always_inline)) {       // issued before a phase's K loop, used by its epilogue
#pragma unroll
        for (int j = 0; j < decltype(ntw_c)::value; ++j) bi[j] = *(const f32x4*)(bias + (tl[j] < NHT ? tl[j] : NHT - 1) * 16 + 4 * kk);
    };
    // save_half: PAIR's first hidden layer, which both workgroups compute in full -- each writes the tiles of its own half to hsave
    auto hidden_epilogue = [&](auto npt_c, auto rows_c, auto emask_c, auto ctx_c, auto ntw_c, const int* htile,
                               f32x4 (*acc)[decltype(ntw_c)::value], const f32x4* bias, int out_reg, float* hsave, const float* bt,
                               bool save_half) __attribute__((always_inline)) {
        constexpr int NPT = decltype(npt_c)::value, ROWS = decltype(rows_c)::value, EMASK = decltype(emask_c)::value;
        constexpr int NTWH = decltype(ntw_c)::value;
        constexpr bool WITH_CTX = decltype(ctx_c)::value != 0;
        f32x4 cxv[WITH_CTX ? NPT : 1][NTWH];                     // all context reads in flight together, ahead of the arithmetic
        if constexpr (WITH_CTX) {
#pragma unroll
            for (int j = 0; j < NTWH; ++j)
#pragma unroll
                for (int q = 0; q < NPT; ++q) {
                    int pix = (q + qbase) * 16 + pl;
                    pix = pix < ROWS * W ? pix : ROWS * W - 1;
                    const int tl = htile[j] < NHT ? htile[j] : NHT - 1;
                    const float* cr = (const float*)(smem + (size_t)G::CTX_OFF * 16) + (tl * 16 + 4 * kk) * G::CSTR + pix;
#pragma unroll
                    for (int r = 0; r < 4; ++r) cxv[q][j][r] = cr[r * G::CSTR];
                }
        }
#pragma unroll
        for (int oi = 0; oi < NTWH * NPT; ++oi) {
            const int j = oi / NPT, q = oi % NPT;
            if (htile[j] >= NHT) continue;
            const f32x4 bi = bias[j];
            {
                if (j == NTWH - 1 && !((EMASK >> q) & 1)) continue;
                const int pix = (q + qbase) * 16 + pl;
                if (pix >= ROWS * W) continue;
                const int row = pix / W, col = pix - row * W;
                f32x4 v = acc[q][j] + bi;
                if constexpr (BORDER) { if (r0 + row < H) v += border_terms(bt, NH, htile[j] * 16 + 4 * kk, r0 + row, col); }   // conv.py:71-83
                if constexpr (WITH_CTX) v += cxv[q][j];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu_f(v[r]);
                if (r0 + row >= H) v = f32x4{0.f, 0.f, 0.f, 0.f};
                split_store4(smem + (size_t)out_reg * 16, row * RS + col + 1, htile[j] * 4 + kk, v, H16, H8);
                // training: the rows this workgroup OWNS (not its halo) go to HBM for the backward pass
                if (hsave && row < R && r0 + row < H && (!save_half || (htile[j] >= NHT / 2) == (half != 0)))
                    *(f32x4*)(hsave + ((size_t)b * HW + (size_t)gpix(r0 + row, col)) * NH + htile[j] * 16 + 4 * kk) = v;
            }
        }
    };

    // operands of the final transform (tf_train.py:56-75): requested BEFORE the last hidden layer's epilogue -- the vector
    // memory counter retires in order, so a request issued right in front of the output pair's K loop would stall that loop's
    // third step (the first to wait for a refill issued behind it) for a whole HBM round trip; here they have an epilogue
    // and a barrier to arrive
    constexpr int NZF = PAIR ? NZ / 2 : NZ;                      // z channels this workgroup finishes (PAIR: those of its half, from fcb)
    constexpr int NEL = (NZF * R * W + 255) / 256;
    float fz[NEL], fq[NEL][6];
    float fb[NEL][2];
    auto load_final_operands = [&]() {
        const int fcb = PAIR ? half * NZF : 0;
#pragma unroll
        for (int e = 0; e < NEL; ++e) {
            const int idx = tid + e * 256;
            const int ic = idx < NZF * R * W ? idx : NZF * R * W - 1;
            const int c = fcb + ic / (R * W), pix = ic % (R * W);
            const int rr = r0 + pix / W < H ? r0 + pix / W : H - 1;
            const unsigned gi = 4u * (unsigned)(c * HW + gpix(rr, pix & (W - 1)));
            const int cm = (c >> 4) * 32 + (c & 15);
            fb[e][0] = p.bias[DEPTH][cm]; fb[e][1] = p.bias[DEPTH][cm + 16];
            fz[e] = 0.f;
            if (p.mode == MODE_IAF || p.mode == MODE_INVERSE) fz[e] = ldf(p.zin + img_z, gi);
            if (p.mode == MODE_POSTERIOR) {
                fq[e][0] = ldf(p.qm + img_z, gi) + ldf(p.rm + img_z, gi); fq[e][1] = ldf(p.ql + img_z, gi) + ldf(p.rl + img_z, gi);
                fq[e][2] = ldf(p.eps + img_z, gi); fq[e][3] = ldf(p.pm + img_z, gi); fq[e][4] = ldf(p.pl + img_z, gi);
            }
        }
    };
    // ---- XCH: halo rows through memory ---------------------------------------------------------------------------------
    // p.xh [layer][B * nrb][W slots x H16 x 16 bytes]: row 0 of block (b, k)'s hidden layer, as it sits in LDS.
    // THE DATA IS THE FLAG: between launches every dword of the buffer holds XSENT (two SIGNALLING bf16 NaNs: no arithmetic
    // produces).  The producer copies the row from LDS -- where its epilogue has just put it, behind the epilogue's barrier -- with
    // 16-byte stores, consecutive lanes to consecutive addresses (whole 128-byte lines leave the CU; the 8-byte pieces round 3
    // stored straight from the MFMA layout, 320 bytes apart, needed ~4 us to land) and is done: no acknowledgement to wait for,
    // no flag to raise.  The block above loads the whole row, takes every 16-byte unit none of whose dwords is XSENT for what it
    // is, asks again for the others, and -- the row in its registers -- puts XSENT back.  One trip to memory per hand-over instead
    // of the three of round 3's flag words (stores acknowledged -> flag -> poll -> row loads; measured this round with flags:
    // poll 1.6 k + row 1.2 k cycles on the consumer, 0.3 - 0.7 k on the producer, and no difference between flags in the XCD's L2
    // and in memory -- sc1 loads are served by the memory side either way:
    // profiles/r04/experiments/stamps_l2_vs_memory_import_breakdown.txt).
    // Every access carries sc1 (agent scope: coherent per access across the XCDs' L2s and past this CU's L1, which may hold the
    // previous launch's row).  Agent-scope FENCES were measured in round 3: buffer_wbl2 / buffer_inv sc1 write back and
    // invalidate the whole L2 -- with it the weight packs every workgroup streams -- and the launch took 78 k instead of 54 k cycles.
    // Giving up: every wait is bounded.  A wave whose wait ends with units missing fills its share of the imported row with NaN
    // (which then flows through the remaining layers to this block's outputs and, where an exported row depends on it, on to the
    // blocks above), raises the sticky word p.xctl[IAF_XCTL_STICKY] -- the buffer can no longer be trusted to be all XSENT, so every later
    // launch on it imports NaN without looking -- and the host-visible p.xerr, which the next call on the stack returns as
    // IAF_ERR_EXCHANGE; iaf_stack_set_halo_exchange re-arms the buffers.  Wrong numbers never leave silently.
    constexpr unsigned XSENT = F16 ? IAF_XSENT_F16 : IAF_XSENT;
    constexpr unsigned XNAN = F16 ? 0x7e007e00u : 0x7fc07fc0u;                       // a pair of quiet NaNs of the planes' type
    constexpr int XSC1 = 16;                                                         // aux bits of the buffer instructions: sc1
    constexpr int XNU = W * H16, XNL = (XNU + 255) / 256;                            // 16-byte units of a row; per lane
    auto xch_rsrc = [&](int l, int slot) -> __amdgpu_buffer_rsrc_t {                  // one row as a buffer: accesses past its end are dropped
        // (the base through readfirstlane: the descriptor must be SEEN to be wave-uniform, or every access is wrapped in a
        // readfirstlane / exec-mask "waterfall" loop -- cdna_hip_programming.md T20)
        const unsigned long long a = (unsigned long long)(p.xh + ((size_t)l * p.B * p.nrb + slot) * G::xrow_bytes());
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)G::xrow_bytes(), 0x00020000);
    };
    // Both run in the four HELPER waves (units htid + 256 u).
    constexpr int XNLH = (XNU + 255) / 256;
    // behind the barrier after hidden layer l's epilogue: row 0 of its region -> the row buffer of this block
    auto xch_export = [&](int l, int reg) {
        if constexpr (XCH) {
            if (rbk > 0 && !((p.xknob & 8u) && b == 0 && rbk == p.nrb - 1 && l == 0)) {      // (test knob 8: one row is never handed over)
                const __amdgpu_buffer_rsrc_t r = xch_rsrc(l, xslot);
                const f32x4* src = smem4 + reg + H16;                                 // row 0, slots 1 .. W
                f32x4 t[XNLH];
#pragma unroll
                for (int u = 0; u < XNLH; ++u) { const int i = htid + 256 * u; t[u] = src[i < XNU ? i : XNU - 1]; }
                // (no data dword can equal the "not there yet" pattern -- a pair of signalling bf16 NaNs, IAF_XSENT -- so the row leaves as it is)
#pragma unroll
                for (int u = 0; u < XNLH; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t[u]), r, 16 * (htid + 256 * u), 0, XSC1);
                if (l == 0 && p.dbg && htid == 0) {
                    p.dbg[(size_t)blockIdx.x * 32 + 27] = __builtin_readcyclecounter();
                    p.dbg[(size_t)blockIdx.x * 32 + 26] = (unsigned long long)xslot + 1;
                }
            }
        }
    };
    // row R of the region <- row 0 of the block below (zeros past the image); the compute waves meet it at their next barrier
    auto xch_import = [&](int l, int reg) {
        if constexpr (XCH) {
            f32x4* dst = smem4 + reg + (R * RS + 1) * H16;
            if (r0 + R < H) {
                const __amdgpu_buffer_rsrc_t r = xch_rsrc(l, xslot + 1);
                u32x4 t[XNLH];
                unsigned pad = 0;                                                    // units nobody's epilogue writes: the two 16-byte units of slot padding
#pragma unroll
                for (int u = 0; u < XNLH; ++u) {
                    const int i = htid + 256 * u;
                    if (i >= XNU || (i % H16) >= NPL * H8) pad |= 1u << u;
                }
                const int tmo = (p.xknob & 8u) ? (1 << 10) : (1 << 20);              // a bounded wait: a lost neighbour must not hang the GPU
                int it = xdead ? tmo : 0;
                if (p.dbg && htid == 0) p.dbg[(size_t)blockIdx.x * 32 + (l == 0 ? 14 : 18)] = __builtin_readcyclecounter();
                while (it < tmo) {
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < XNLH; ++u) t[u] = __builtin_amdgcn_raw_buffer_load_b128(r, 16 * (htid + 256 * u), 0, XSC1);
#pragma unroll
                    for (int u = 0; u < XNLH; ++u)
                        ok = ok && (((pad >> u) & 1u) || (t[u][0] != XSENT && t[u][1] != XSENT && t[u][2] != XSENT && t[u][3] != XSENT));
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(8);                                     // (the row is not there yet: it is the only thing this wave waits for)
                    ++it;
                }
                if (p.dbg && htid == 0) {
                    p.dbg[(size_t)blockIdx.x * 32 + (l == 0 ? 16 : 19)] = __builtin_readcyclecounter();
                    p.dbg[(size_t)blockIdx.x * 32 + (l == 0 ? 24 : 25)] = (unsigned long long)it + 1;
                    p.dbg[(size_t)blockIdx.x * 32 + 26] = (unsigned long long)xslot + 1;
                }
                if (it < tmo) {                                                      // taken: the buffer is all XSENT again for the next launch
#pragma unroll
                    for (int u = 0; u < XNLH; ++u)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{XSENT, XSENT, XSENT, XSENT}, r, 16 * (htid + 256 * u), 0, XSC1);
                } else {                                                             // gave up (or the buffer is marked dead): NaN, loudly
#pragma unroll
                    for (int u = 0; u < XNLH; ++u) t[u] = u32x4{XNAN, XNAN, XNAN, XNAN};
                    if (lane == 0 && !xdead) {
                        if (p.xerr) __hip_atomic_store(p.xerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(p.xctl + IAF_XCTL_STICKY, 1ull, __ATOMIC_RELAXED, XSCOPE);
                    }
                }
#pragma unroll
                for (int u = 0; u < XNLH; ++u) { const int i = htid + 256 * u; if (i < XNU) dst[i] = __builtin_bit_cast(f32x4, t[u]); }
                if (p.dbg && htid == 0) p.dbg[(size_t)blockIdx.x * 32 + (l == 0 ? 17 : 20)] = __builtin_readcyclecounter();
            } else {
                for (int i = htid; i < XNU; i += 256) dst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    // ---- PAIR: the two workgroups of an item hand each other their channel half of the last hidden layer's region (all R + 1 rows it
    // holds: the output pair reads every one).  p.xh [B * nrb][half][G::PUNITS x 16 bytes], unit i = ((row W + col) 3 + plane) H8/2 + u;
    // the same hand-over as the rows above: all-ones between launches, sc1 stores by the producer's helper waves, the consumer's take
    // every unit without an all-ones dword and put the pattern back.
    constexpr int PNL = PAIR ? (G::PUNITS + 255) / 256 : 1;
    constexpr int LASTH = ((DEPTH - 1) & 1) ? G::HREG1 : G::HREG0;
    auto pair_rsrc = [&](int hf) -> __amdgpu_buffer_rsrc_t {
        const unsigned long long a = (unsigned long long)(p.xh + ((size_t)xslot * 2 + hf) * G::prow_bytes());
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)G::prow_bytes(), 0x00020000);
    };
    auto pair_lds = [&](int i, int hf) -> int {                  // unit i of half hf inside the last hidden region (16-byte units)
        const int sl = i / G::PUS, rem = i - sl * G::PUS;
        const int pn = rem / (H8 / 2), u = rem - pn * (H8 / 2);
        const int row = sl / W, col = sl - row * W;
        return LASTH + (row * RS + col + 1) * H16 + pn * H8 + hf * (H8 / 2) + u;
    };
    auto pair_export = [&]() {
        if constexpr (PAIR) {
            const __amdgpu_buffer_rsrc_t r = pair_rsrc(half);
            u32x4 t[PNL];
#pragma unroll
            for (int u = 0; u < PNL; ++u) {
                const int i = htid + 256 * u;
                t[u] = __builtin_bit_cast(u32x4, smem4[pair_lds(i < G::PUNITS ? i : G::PUNITS - 1, half)]);
            }
#pragma unroll
            for (int u = 0; u < PNL; ++u)
                __builtin_amdgcn_raw_buffer_store_b128(t[u], r, 16 * (htid + 256 * u), 0, XSC1);       // (past the end: dropped)
            if (p.dbg && htid == 0) p.dbg[(size_t)blockIdx.x * 32 + 27] = __builtin_readcyclecounter();
        }
    };
    auto pair_import = [&]() {
        if constexpr (PAIR) {
            const __amdgpu_buffer_rsrc_t r = pair_rsrc(half ^ 1);
            u32x4 t[PNL];
            unsigned pad = 0;
#pragma unroll
            for (int u = 0; u < PNL; ++u)
                if (htid + 256 * u >= G::PUNITS) pad |= 1u << u;
            const int tmo = (p.xknob & 8u) ? (1 << 10) : (1 << 20);
            int it = xdead ? tmo : 0;
            if (p.dbg && htid == 0) p.dbg[(size_t)blockIdx.x * 32 + 18] = __builtin_readcyclecounter();
            while (it < tmo) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < PNL; ++u) t[u] = __builtin_amdgcn_raw_buffer_load_b128(r, 16 * (htid + 256 * u), 0, XSC1);
#pragma unroll
                for (int u = 0; u < PNL; ++u)
                    ok = ok && (((pad >> u) & 1u) || (t[u][0] != XSENT && t[u][1] != XSENT && t[u][2] != XSENT && t[u][3] != XSENT));
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
                ++it;
            }
            if (p.dbg && htid == 0) {
                p.dbg[(size_t)blockIdx.x * 32 + 19] = __builtin_readcyclecounter();
                p.dbg[(size_t)blockIdx.x * 32 + 25] = (unsigned long long)it + 1;
                p.dbg[(size_t)blockIdx.x * 32 + 26] = (unsigned long long)(2 * xslot + half) + 1;
            }
            const bool taken = it < tmo;
            if (!taken) {
#pragma unroll
                for (int u = 0; u < PNL; ++u) t[u] = u32x4{XNAN, XNAN, XNAN, XNAN};
            }
#pragma unroll
            for (int u = 0; u < PNL; ++u) { const int i = htid + 256 * u; if (i < G::PUNITS) smem4[pair_lds(i, half ^ 1)] = __builtin_bit_cast(f32x4, t[u]); }
            // the compute waves are waiting for exactly this: the barrier as soon as the LDS stores are done (s_barrier counts waves: it
            // pairs with the compute waves' __syncthreads), the re-arming -- sc1 stores whose acknowledgement a __syncthreads in front of
            // them would wait for: 1.5 k cycles of the compute waves' wait -- behind it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (p.dbg && htid == 0) p.dbg[(size_t)blockIdx.x * 32 + 20] = __builtin_readcyclecounter();
            if (taken) {
#pragma unroll
                for (int u = 0; u < PNL; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{XSENT, XSENT, XSENT, XSENT}, r, 16 * (htid + 256 * u), 0, XSC1);
            } else if (lane == 0 && !xdead) {
                if (p.xerr) __hip_atomic_store(p.xerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(p.xctl + IAF_XCTL_STICKY, 1ull, __ATOMIC_RELAXED, XSCOPE);
            }
        }
    };
    // The helper's whole life: one barrier for each of the compute waves' (same order, same count -- s_barrier counts waves),
    // its traffic in between.  Behind the barrier that ends layer l's epilogue it sends row 0 of that layer to the block above and
    // starts asking for the row below: the row is in LDS by the time the compute waves have multiplied the taps of their own rows
    // and arrive at the barrier in front of the taps that read it.
    if constexpr (HELP) {
        if (is_helper) {
            __builtin_amdgcn_s_setprio(IAF_HELPER_PRIO);          // (0: they take the issue slots the compute wave of their SIMD leaves)
            {   // the context rows of the first epilogue, summed and staged [channel][pixel] (what load_ctx + store_ctx do in the
                // recomputing kernels, where they hold 40-80 registers of every compute wave across the first conv)
                constexpr int NCIH = (NCIT + 255) / 256;
                float* creg = (float*)(smem + (size_t)G::CTX_OFF * 16);
                const int vpx = (H - r0) * W < CPX ? (H - r0) * W : CPX;
                // HL0: this helper's left-over unit of the first hidden layer, between issuing the context loads and staging what they brought.
                // (Measured on one box, profiles/r05/experiments/ab_hl0_*.txt: context loads first, then the unit's fragments with a 4-step
                //  look-ahead: 8x8 step 16.56 -> 16.10 us, 16x16 equal; the fragments FIRST -- so that the unit's first MFMA need not wait for
                //  the context's round trip -- was slower at either ring depth: 16.5 - 16.8 us.)
                constexpr int RDL0 = PartL0::NSTEP - 1;
                [[maybe_unused]] f32x4 accu0[1][1], biu0[1], wrl0[HL0 ? RDL0 + 1 : 1][1][3];
                [[maybe_unused]] int lt0[1] = {NW * NFULL + (wave - NW_COMPUTE) % (NX ? NX : 1)};
                [[maybe_unused]] int q0 = -1;
                if constexpr (HL0) {
                    static_for<GN>([&](auto g_c) {
                        constexpr int GI = decltype(g_c)::value;
                        constexpr int EM = fused_extra_mask(NPT0, GN, GI);
                        static_assert(fused_popcount(EM) <= 1, "HL0: one pixel tile per left-over unit");
                        if constexpr (EM != 0) {
                            if ((wave - NW_COMPUTE) / NX == GI) q0 = EM == 1 ? 0 : EM == 2 ? 1 : EM == 4 ? 2 : 3;
                        }
                    });
                }
                f32x4 v[NCIH], v2[NCIH];
                unsigned cval = 0;
#pragma unroll
                for (int u = 0; u < NCIH; ++u) {
                    const int idx = htid + 256 * u, ic = idx < NCIT ? idx : NCIT - 1;
                    const int c = ic / CG, g4 = (ic - c * CG) * 4;
                    const bool inside = g4 < vpx;
                    const int g4c = inside ? g4 : 0, row = g4c / W, col4 = g4c - row * W;
                    const unsigned gi = 4u * (unsigned)(c * HW + gpix(r0 + row, col4 + (FLIP ? 3 : 0)));
                    v[u] = ldf4(p.ctx + img_h, gi);
                    if (p.ctx2) v2[u] = ldf4(p.ctx2 + img_h, gi);
                    cval |= (inside ? 1u : 0u) << u;
                }
                if constexpr (HL0) {
                    if (q0 >= 0) {
                        static_for<RDL0>([&](auto i) {
                            ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NPL>{}, wrl0[decltype(i)::value], wb0, NHT, lt0,
                                      PartL0{}, decltype(i)::value, std::integral_constant<int, -1>{}, 0);
                        });
                        load_bias(std::integral_constant<int, 1>{}, lt0, p.bias[0], biu0);
                        qbase = q0;
                        conv_phase(std::integral_constant<int, RDL0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{},
                                   std::integral_constant<int, G::rows_h(0)>{}, std::integral_constant<int, 1>{}, G::ZREG, Z16, Z8, wb0, NHT, lt0,
                                   wrl0, accu0, PartL0{}, std::integral_constant<int, 300>{}, std::integral_constant<bool, false>{}, 0);
                        qbase = 0;
                    }
                }
#pragma unroll
                for (int u = 0; u < NCIH; ++u) {
                    const int idx = htid + 256 * u;
                    if (idx < NCIT) {
                        const int c = idx / CG, g4 = (idx - c * CG) * 4;
                        f32x4 t = v[u];
                        if (p.ctx2) t += v2[u];                  // up_context + down_context (tf_train.py:58)
                        if (!((cval >> u) & 1u)) t = f32x4{0.f, 0.f, 0.f, 0.f};
                        *(f32x4*)(creg + c * CSTR + g4) = FLIP ? f32x4{t[3], t[2], t[1], t[0]} : t;
                    }
                }
                __syncthreads();                                  // first conv done, context staged
                if constexpr (HL0) {
                    if (q0 >= 0) {
                        qbase = q0;
                        hidden_epilogue(std::integral_constant<int, 1>{}, std::integral_constant<int, G::rows_h(0)>{}, std::integral_constant<int, 1>{},
                                        std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, lt0, accu0, biu0, G::HREG0, p.hsave[0],
                                        p.border[0], false);
                        qbase = 0;
                    }
                }
            }
            __syncthreads();                                      // first epilogue done: h_0 complete
            xch_export(0, G::HREG0);
            // HOUT8: this helper's half of the output pair (below) -- its first fragments are requested while the last hidden layer is still
            // being multiplied (the helper is idle there, or done with its left-over unit)
            typedef PairPart<NH / 32, 1> PartO8H;
            constexpr int RDOH8 = 3;
            [[maybe_unused]] f32x4 wroh8[G::HOUT8 ? RDOH8 + 1 : 1][NTWO][3];
            [[maybe_unused]] int ot8[NTWO];
#pragma unroll
            for (int j = 0; j < NTWO; ++j) ot8[j] = (wave - NW_COMPUTE) * NTWO + j;
            static_for<DEPTH - 1>([&](auto lm_c) {
                constexpr int l = decltype(lm_c)::value + 1;
                constexpr int IN_REG = ((l - 1) & 1) ? G::HREG1 : G::HREG0, OUT_REG = (l & 1) ? G::HREG1 : G::HREG0;
                if constexpr (XCH) {
                    xch_import(l - 1, IN_REG);
                    __syncthreads();                              // row R of h_{l-1} is there
                }
                if constexpr (HLEFT) {
                    // this helper's left-over unit of layer l: the tile its compute wave would have held in its last slot, the pixel tile of
                    // that wave's group -- all of the layer's taps (the row below is in LDS), bias + ELU + split -> the layer's region
                    const int hw = wave - NW_COMPUTE;
                    constexpr int NPTL = (G::rows_h(l) * W + 15) / 16;
                    static_for<GN>([&](auto g_c) {
                        constexpr int GI = decltype(g_c)::value;
                        constexpr int EM = fused_extra_mask(NPTL, GN, GI);
                        static_assert(fused_popcount(EM) <= 1, "HLEFT: one pixel tile per left-over unit");
                        if constexpr (EM != 0) {
                            if (hw / NX != GI) return;
                            int lt[1] = {NW * NFULL + hw % NX};
                            qbase = EM == 1 ? 0 : EM == 2 ? 1 : EM == 4 ? 2 : 3;
                            const f32x4* wbl = (const f32x4*)p.wp3[l];
                            // (the unit is 6 MFMAs per step: a look-ahead of RDL steps = ~800 cycles of cover for its fragments; the steps in
                            //  the order of the kernels without helpers: PartHid)
                            constexpr int RDL = 6;
                            f32x4 wrl[RDL + 1][1][3], accu[1][1], biu[1];
                            static_for<RDL>([&](auto i) {
                                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NPL>{}, wrl[decltype(i)::value], wbl, NHT, lt,
                                          PartHid{}, decltype(i)::value, std::integral_constant<int, -1>{}, 0);
                            });
                            load_bias(std::integral_constant<int, 1>{}, lt, p.bias[l], biu);
                            conv_phase(std::integral_constant<int, RDL>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{},
                                       std::integral_constant<int, G::rows_h(l)>{}, std::integral_constant<int, 1>{}, IN_REG, H16, H8, wbl, NHT, lt,
                                       wrl, accu, PartHid{}, std::integral_constant<int, 300>{}, std::integral_constant<bool, false>{}, 0);
                            hidden_epilogue(std::integral_constant<int, 1>{}, std::integral_constant<int, G::rows_h(l)>{},
                                            std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, lt,
                                            accu, biu, OUT_REG, p.hsave[l], p.border[l], false);
                            qbase = 0;
                        }
                    });
                }
                if constexpr (G::HOUT8 && l == DEPTH - 1) {
                    static_for<RDOH8>([&](auto i) {
                        ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWO * NPL>{}, wroh8[decltype(i)::value],
                                  (const f32x4*)p.wp3[DEPTH], 2 * NZT, ot8, PartO8H{}, decltype(i)::value, std::integral_constant<int, -1>{}, 0);
                    });
                }
                __syncthreads();                                  // layer l's epilogue done
                xch_export(l, OUT_REG);
            });
            if constexpr (XCH) {
                constexpr int LAST_REG = ((DEPTH - 1) & 1) ? G::HREG1 : G::HREG0;
                xch_import(DEPTH - 1, LAST_REG);
                if constexpr (G::HOUT) {
                    // the output pair's taps of the row below, odd steps: this helper's compute wave's tiles, both pixel tiles
                    constexpr int NPTO_ = (R * W + 15) / 16;
                    const int hw = wave - NW_COMPUTE;
                    int ot[NTWO];
#pragma unroll
                    for (int j = 0; j < NTWO; ++j) ot[j] = hw * NTWO + j;
                    typedef BelowPar<NH / 32, 1> PartOBH;
                    constexpr int RDOH = 3;
                    const f32x4* wboh = (const f32x4*)p.wp3[DEPTH];
                    f32x4 wroh[RDOH + 1][NTWO][3], accoh[NPTO_][NTWO];
                    static_for<RDOH>([&](auto i) {
                        ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWO * NPL>{}, wroh[decltype(i)::value], wboh, 2 * NZT,
                                  ot, PartOBH{}, decltype(i)::value, std::integral_constant<int, -1>{}, 0);
                    });
                    __syncthreads();                              // row R of the last hidden layer is there
                    conv_phase(std::integral_constant<int, RDOH>{}, std::integral_constant<int, NPTO_>{}, std::integral_constant<int, NTWO>{},
                               std::integral_constant<int, R>{}, std::integral_constant<int, (1 << NPTO_) - 1>{}, LAST_REG,
                               H16, H8, wboh, 2 * NZT, ot, wroh, accoh, PartOBH{}, std::integral_constant<int, 0>{},
                               std::integral_constant<bool, false>{}, 0);
                    float* part1 = (float*)(smem + (size_t)G::XB_OFF * 16) + R * W * G::XB_STRIDE;
#pragma unroll
                    for (int j = 0; j < NTWO; ++j)
#pragma unroll
                        for (int q = 0; q < NPTO_; ++q) {
                            const int pix = q * 16 + pl;
                            if (pix >= R * W) continue;
#pragma unroll
                            for (int r = 0; r < 4; ++r) part1[pix * G::XB_STRIDE + ot[j] * 16 + 4 * kk + r] = accoh[q][j][r];
                        }
                } else {
                    __syncthreads();                              // row R of the last hidden layer is there
                }
            }
            if constexpr (PAIR) {
                if (!((p.xknob & 8u) && b == 0 && rbk == 0 && half == 1)) pair_export();     // (test knob 8: one half is never handed over)
                pair_import();                                    // (ends with the barrier: the partner's half of the last hidden layer is there)
            }
            if constexpr (G::HOUT8) {
                // the output pair, odd K steps: this helper's compute wave's tiles
                constexpr int NPTO_ = (R * W + 15) / 16, LAST_REG = ((DEPTH - 1) & 1) ? G::HREG1 : G::HREG0;
                const f32x4* wboh = (const f32x4*)p.wp3[DEPTH];
                f32x4 accoh[NPTO_][NTWO];
                conv_phase(std::integral_constant<int, RDOH8>{}, std::integral_constant<int, NPTO_>{}, std::integral_constant<int, NTWO>{},
                           std::integral_constant<int, R>{}, std::integral_constant<int, (1 << NPTO_) - 1>{}, LAST_REG,
                           H16, H8, wboh, 2 * NZT, ot8, wroh8, accoh, PartO8H{}, std::integral_constant<int, 0>{},
                           std::integral_constant<bool, false>{}, 0);
                float* part1 = (float*)(smem + (size_t)G::XB_OFF * 16) + R * W * G::XB_STRIDE;
#pragma unroll
                for (int j = 0; j < NTWO; ++j)
#pragma unroll
                    for (int q = 0; q < NPTO_; ++q) {
                        const int pix = q * 16 + pl;
                        if (pix >= R * W) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) part1[pix * G::XB_STRIDE + ot8[j] * 16 + 4 * kk + r] = accoh[q][j][r];
                    }
            }
            raise_range();                                        // (the helpers' own units went through the split too)
            __syncthreads();                                      // output pair in the exchange buffer
            // ---- the block's free-bits reductions, by the helper waves of the workgroup that arrives last (StepP::fin_*) ----
            if (p.fin_ctl && p.mode == MODE_POSTERIOR) {
                unsigned* fmail = (unsigned*)smem;                // (every region is dead once the compute waves are through the final loop)
                __syncthreads();                                  // every compute wave's partial sums are in memory
                if (htid == 0) {
                    const unsigned g = blockIdx.x & 7u, ng = (gridDim.x - g + 7u) >> 3, ngroups = gridDim.x < 8u ? gridDim.x : 8u;
                    unsigned last = 0;
                    if ((unsigned)__hip_atomic_fetch_add(p.fin_ctl + 32 * g, 1ull, __ATOMIC_RELAXED, XSCOPE) == ng - 1u) {
                        __hip_atomic_store(p.fin_ctl + 32 * g, 0ull, __ATOMIC_RELAXED, XSCOPE);
                        if ((unsigned)__hip_atomic_fetch_add(p.fin_ctl + 256, 1ull, __ATOMIC_RELAXED, XSCOPE) == ngroups - 1u) {
                            __hip_atomic_store(p.fin_ctl + 256, 0ull, __ATOMIC_RELAXED, XSCOPE);
                            last = 1;
                        }
                    }
                    fmail[0] = last;
                }
                __syncthreads();
                if (!fmail[0]) return;
                // S[b][c] = sum over the row blocks, in row order, eight 16-byte loads in flight (as iaf_kl_finish_kernel)
                float* S = (float*)smem + 64;                     // (behind the mail word; B * n_z <= 8192 floats: checked by the host)
                {
                    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p.kl_part, 0, p.B * p.nrb * NZ * 4, 0x00020000);
                    constexpr int Z4 = NZ / 4;
                    for (int i = htid; i < p.B * Z4; i += 256) {
                        const int bb = i / Z4, c4 = i - bb * Z4;
                        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                        for (int k0 = 0; k0 < p.nrb; k0 += 8) {
                            u32x4 v8[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const int rr = k0 + k < p.nrb ? k0 + k : p.nrb - 1;
                                v8[k] = __builtin_amdgcn_raw_buffer_load_b128(r, 4 * ((bb * p.nrb + rr) * NZ + 4 * c4), 0, XSC1);
                            }
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                if (k0 + k < p.nrb) a += __builtin_bit_cast(f32x4, v8[k]);
                        }
                        *(f32x4*)(S + (size_t)bb * NZ + 4 * c4) = a;
                    }
                }
                __syncthreads();
                if (htid < 64) {                                  // one wave: n_z <= 64 channels, then B images in rounds of 64
                    float a = 0.f;
                    if (htid < NZ) {
                        float m = 0.f;
                        for (int bb = 0; bb < p.B; ++bb) m += S[(size_t)bb * NZ + htid];
                        a = fmaxf(m / (float)p.B, p.fin_kl_min);  // kl_ave[c] = max(mean_b S[b,c], kl_min)          (tf_train.py:79-80)
                        if (p.fin_gate && p.fin_kl_min > 0.f) p.fin_gate[htid] = (m / (float)p.B > p.fin_kl_min) ? 1.f : 0.f;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);      // (the finish kernel's tree: part[t] += part[t + o])
                    const float fb = __shfl(a, 0, 64);
                    for (int bb = htid; bb < p.B; bb += 64) {
                        float c = 0.f;
                        for (int cc = 0; cc < NZ; ++cc) c += S[(size_t)bb * NZ + cc];
                        p.fin_cost[bb] = c;                                           // tf_train.py:85
                        p.fin_obj[bb] = p.fin_kl_min > 0.f ? fb : c;                  // tf_train.py:82 / 84
                    }
                }
            }
            return;
        }
    }

    // ---- hidden layers ---------------------------------------------------------------------------------------------
    // the first (or only) part of a hidden layer l >= 1 and of the output pair; XCH: the taps of the own rows, the imported row's
    // taps follow as a second part behind the import
    typedef std::conditional_t<XCH != 0, std::conditional_t<VAR == 0, PartOwnTri, PartOwn>, PartHid> PartH1;
    constexpr bool HO8 = G::HOUT8 && HELP;                       // the output pair's K steps alternate between a compute wave and its helper
    constexpr bool XK2 = PAIR || G::HOUT || HO8;                 // ... their sums meet in the exchange buffer's two parts
    typedef std::conditional_t<XCH != 0, PartOwn, std::conditional_t<HO8, PairPart<NPAIR_H, 0>, PartFull>> PartO1;
    int otile[NTWO];
#pragma unroll
    for (int j = 0; j < NTWO; ++j) otile[j] = wave * NTWO + j;
    // PAIR: the tiles of this workgroup's half of the last hidden layer: tiles half NHT/2 + (0 .. NHT/2 - 1), dealt out over the waves as a
    // layer of NHT/2 tiles would be (rounds of 4 + one left-over tile shared per pixel tile)
    constexpr int NHTP = NHT / 2, NFULLP = NHTP / NW, NTWP = NFULLP + 1;
    int ptile[NTWP];
    f32x4 wrp[PAIR ? UB : 1][NTWP][3];
    if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < NTWP; ++j)
            ptile[j] = NHTP * half + ((j < NFULLP) ? ((j & 1) ? NW * (j + 1) - 1 - wave : wave + NW * j) : NW * NFULLP);
    }
    // PAIR's output pair: this workgroup's NZT tiles (m and s of its z channels) over four waves = two waves per tile, which take the K
    // steps of a part alternately (PairPart); first the part over the input channels of the own half, then -- behind the import -- the rest.
    // Own half 0: input pairs 0 .. NPO-1 are entirely its own; half 1: the last NPO pairs.
    const int opar = wave & 1;
    constexpr int NPO = (NH / 2) / 32, NPR = NPAIR_H - NPO;
    const int opair_own = half ? NPAIR_H - NPO : 0, opair_rest = half ? 0 : NPO;
    if constexpr (PAIR) otile[0] = NZT * half + (wave >> 1);
    f32x4 wr1[UB][NTWH][3];          // hidden layer l uses ring l & 1 (wr0 / wr1): the other one receives layer l + 1's first steps
    f32x4 wrh1[HLEFT ? UB : 1][NTW1][3];      // HLEFT: layer 1's ring of NFULL slots (wr1 is then unused)
    f32x4 wro[UO][NTWO][3];          // ring of the output pair
    const f32x4* wbo = (const f32x4*)p.wp3[DEPTH];
    auto preload_out = [&]() __attribute__((always_inline)) {
        if constexpr (PAIR) {
            static_for<2>([&](auto par_c) {
                if (opar != decltype(par_c)::value) return;
                static_for<RDO>([&](auto i) {
                    ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWO * NPL>{}, wro[decltype(i)::value], wbo, 2 * NZT,
                              otile, PairPart<NPO, decltype(par_c)::value>{}, decltype(i)::value, ALL, opair_own);
                });
            });
            return;
        }
        static_for<RDO>([&](auto i) {
            ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWO * NPL>{}, wro[decltype(i)::value], wbo, 2 * NZT,
                      otile, PartO1{}, decltype(i)::value, ALL, 0);
        });
    };
    // the weights of the phase after hidden layer l -- the next hidden layer's ring, or the output pair's -- start travelling
    // while layer l's epilogue runs
    auto preload_after = [&](auto l_c) __attribute__((always_inline)) {
        constexpr int l = decltype(l_c)::value;
        if constexpr (PAIR && l + 1 < DEPTH) {
            const f32x4* wbn = (const f32x4*)p.wp3[l + 1];
            static_for<RDH>([&](auto i) {
                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWP * NPL>{}, wrp[decltype(i)::value], wbn, NHT, ptile,
                          PartHid{}, decltype(i)::value, ALL, 0);
            });
        } else if constexpr (l + 1 < DEPTH && HLEFT) {
            const f32x4* wbn = (const f32x4*)p.wp3[l + 1];
            static_for<RDH>([&](auto i) {
                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTW1 * NPL>{}, wrh1[decltype(i)::value], wbn, NHT, htile,
                          PartH1{}, decltype(i)::value, ALL, 0);
            });
        } else if constexpr (l + 1 < DEPTH) {
            const f32x4* wbn = (const f32x4*)p.wp3[l + 1];
            static_for<RDH>([&](auto i) {
                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWH * NPL>{},
                          ((l + 1) & 1) ? wr1[decltype(i)::value] : wr0[decltype(i)::value], wbn, NHT, htile, PartH1{}, decltype(i)::value,
                          ALL, 0);
            });
        } else {
            preload_out();
        }
    };
    // every wave group runs its own instantiation of a hidden phase (the left-over tile's pixel tiles are compile time)
    static_for<GN>([&](auto g_c) {
        constexpr int GI = decltype(g_c)::value;
        if (xg != GI) return;
        constexpr int EM0 = HL0 ? (1 << NPT0) - 1 : (NX == 0 || !XSPLIT) ? (1 << NPT0) - 1 : fused_extra_mask(NPT0, GN, GI);
        f32x4 acc0[NPT0][NTW0], bi0[NTW0];
        load_bias(std::integral_constant<int, NTW0>{}, htile, p.bias[0], bi0);
        if constexpr (HL0)
            conv_phase(std::integral_constant<int, RD0>{}, std::integral_constant<int, NPT0>{}, std::integral_constant<int, NTW0>{},
                       std::integral_constant<int, G::rows_h(0)>{}, std::integral_constant<int, EM0>{}, G::ZREG, Z16, Z8, wb0, NHT, htile,
                       wrh0, acc0, PartL0{}, std::integral_constant<int, 400 + GI>{}, SET, 0);
        else
            conv_phase(std::integral_constant<int, RD0>{}, std::integral_constant<int, NPT0>{}, std::integral_constant<int, NTW0>{},
                       std::integral_constant<int, G::rows_h(0)>{}, std::integral_constant<int, EM0>{}, G::ZREG, Z16, Z8, wb0, NHT, htile,
                       wr0, acc0, PartL0{}, g_c, SET, 0);
        IAF_FSTAMP(6);
        xch_next_epoch_a();
        if constexpr (!HELP) store_ctx();
        if constexpr (DEPTH == 1) load_final_operands();
        preload_after(std::integral_constant<int, 0>{});
        IAF_FSTAMP(10);
        __syncthreads();                                         // (every wave runs exactly one of the GN instantiations)
        IAF_FSTAMP(11);
        hidden_epilogue(std::integral_constant<int, NPT0>{}, std::integral_constant<int, G::rows_h(0)>{},
                        std::integral_constant<int, EM0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, NTW0>{}, htile, acc0, bi0,
                        G::HREG0, p.hsave[0], p.border[0], PAIR != 0);
    });
    __syncthreads();
    IAF_FSTAMP(2);
    if constexpr (PAIR) {
        constexpr int l = 1, IN_REG = G::HREG0, OUT_REG = G::HREG1;
        {   // the staged context sat in the h_odd region: its zero columns again
            constexpr int H1ROWS = G::rows_reg(1);
            for (int i = tid; i < H1ROWS * 2 * H16; i += 256) {
                const int rs = i / H16, u = i - rs * H16;
                smem4[G::HREG1 + ((rs >> 1) * RS + (rs & 1) * (W + 1)) * H16 + u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        constexpr int NPTL = (G::rows_h(l) * W + 15) / 16;
        const f32x4* wbl = (const f32x4*)p.wp3[l];
        // every wave is a group of its own (the left-over tile's pixel tiles); the centre tap's dead blocks depend on the half
        static_for<2 * NW>([&](auto g_c) {
            constexpr int GI = decltype(g_c)::value % NW, HF = decltype(g_c)::value / NW;
            if (wave != GI || half != HF) return;
            constexpr int EML = fused_extra_mask(NPTL, NW, GI);
            f32x4 accl[NPTL][NTWP], bil[NTWP];
            load_bias(std::integral_constant<int, NTWP>{}, ptile, p.bias[l], bil);
            conv_phase(std::integral_constant<int, RDH>{}, std::integral_constant<int, NPTL>{}, std::integral_constant<int, NTWP>{},
                       std::integral_constant<int, G::rows_h(l)>{}, std::integral_constant<int, EML>{}, IN_REG, H16, H8, wbl, NHT, ptile,
                       wrp, accl, PartHid{}, std::integral_constant<int, 100 * (1 + HF) + GI>{}, SET, 0);
            IAF_FSTAMP(7);
            load_final_operands();
            preload_out();
            hidden_epilogue(std::integral_constant<int, NPTL>{}, std::integral_constant<int, G::rows_h(l)>{},
                            std::integral_constant<int, EML>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NTWP>{}, ptile, accl,
                            bil, OUT_REG, p.hsave[l], p.border[l], false);
            IAF_FSTAMP(12);
        });
        __syncthreads();
    }
    if constexpr (HLEFT) {
        // the second hidden layer with its left-over units in the helper waves: every compute wave multiplies NFULL whole tiles
        constexpr int l = 1, IN_REG = G::HREG0, OUT_REG = G::HREG1;
        {   // the staged context sat in the h_odd region: its zero columns again, before the epilogues fill the rest
            constexpr int H1ROWS = G::rows_reg(1);
            for (int i = tid; i < H1ROWS * 2 * H16; i += 256) {
                const int rs = i / H16, u = i - rs * H16;
                smem4[G::HREG1 + ((rs >> 1) * RS + (rs & 1) * (W + 1)) * H16 + u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        constexpr int NPTL = (G::rows_h(l) * W + 15) / 16, EALL = (1 << NPTL) - 1;
        static_for<GN>([&](auto g_c) {                            // (the groups differ in the centre tap's dead blocks only)
            constexpr int GI = decltype(g_c)::value;
            if (xg != GI) return;
            f32x4 accl[NPTL][NTW1], bil[NTW1];
            load_bias(std::integral_constant<int, NTW1>{}, htile, p.bias[l], bil);
            const f32x4* wbl = (const f32x4*)p.wp3[l];
            conv_phase(std::integral_constant<int, RDH>{}, std::integral_constant<int, NPTL>{}, std::integral_constant<int, NTW1>{},
                       std::integral_constant<int, G::rows_h(l)>{}, std::integral_constant<int, EALL>{}, IN_REG, H16, H8, wbl, NHT, htile,
                       wrh1, accl, PartH1{}, std::integral_constant<int, 400 + GI>{}, SET, 0);
            if constexpr (XCH) {
                static_for<RDH>([&](auto i) {
                    ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTW1 * NPL>{}, wrh1[decltype(i)::value], wbl, NHT, htile,
                              PartBelow{}, decltype(i)::value, ALL, 0);
                });
                IAF_FSTAMP(28);
                __syncthreads();                                  // the helper waves have put the row below into row R of IN_REG
                IAF_FSTAMP(15);
                conv_phase(std::integral_constant<int, RDH>{}, std::integral_constant<int, NPTL>{}, std::integral_constant<int, NTW1>{},
                           std::integral_constant<int, G::rows_h(l)>{}, std::integral_constant<int, EALL>{}, IN_REG, H16, H8, wbl, NHT, htile,
                           wrh1, accl, PartBelow{}, std::integral_constant<int, 400 + GI>{}, ADD, 0);
            }
            IAF_FSTAMP(7);
            load_final_operands();
            preload_after(std::integral_constant<int, l>{});
            hidden_epilogue(std::integral_constant<int, NPTL>{}, std::integral_constant<int, G::rows_h(l)>{},
                            std::integral_constant<int, EALL>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NTW1>{}, htile, accl,
                            bil, OUT_REG, p.hsave[l], p.border[l], false);
            IAF_FSTAMP(12);
        });
        __syncthreads();
    }
    static_for<(PAIR || HLEFT) ? 0 : DEPTH - 1>([&](auto lm_c) {
        constexpr int l = decltype(lm_c)::value + 1;              // hidden layer l reads h_{l-1}, writes h_l into the other region
        constexpr int IN_REG = ((l - 1) & 1) ? G::HREG1 : G::HREG0, OUT_REG = (l & 1) ? G::HREG1 : G::HREG0;
        if constexpr (l == 1) {   // the staged context sat in the h_odd region: its zero columns again, before the epilogue fills the rest
            constexpr int H1ROWS = G::rows_reg(1);
            for (int i = tid; i < H1ROWS * 2 * H16; i += 256) {
                const int rs = i / H16, u = i - rs * H16;
                smem4[G::HREG1 + ((rs >> 1) * RS + (rs & 1) * (W + 1)) * H16 + u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        constexpr int NPTL = (G::rows_h(l) * W + 15) / 16;
        static_for<GN>([&](auto g_c) {
            constexpr int GI = decltype(g_c)::value;
            if (xg != GI) return;
            constexpr int EML = (NX == 0 || !XSPLIT) ? (1 << NPTL) - 1 : fused_extra_mask(NPTL, GN, GI);
            f32x4 accl[NPTL][NTWH], bil[NTWH];
            load_bias(std::integral_constant<int, NTWH>{}, htile, p.bias[l], bil);
                const f32x4* wbl = (const f32x4*)p.wp3[l];
            conv_phase(std::integral_constant<int, RDH>{}, std::integral_constant<int, NPTL>{}, std::integral_constant<int, NTWH>{},
                       std::integral_constant<int, G::rows_h(l)>{}, std::integral_constant<int, EML>{}, IN_REG, H16, H8, wbl, NHT, htile,
                       (l & 1) ? wr1 : wr0, accl, PartH1{}, g_c, SET, 0);
            if constexpr (XCH) {     // the row below arrives while the taps of the own rows were multiplied: its taps now
                static_for<RDH>([&](auto i) {
                    ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWH * NPL>{},
                              (l & 1) ? wr1[decltype(i)::value] : wr0[decltype(i)::value], wbl, NHT, htile, PartBelow{}, decltype(i)::value, ALL, 0);
                });
                if constexpr (l == 1) IAF_FSTAMP(28);
                __syncthreads();                                  // the helper waves have put the row below into row R of IN_REG
                if constexpr (l == 1) IAF_FSTAMP(15);
                conv_phase(std::integral_constant<int, RDH>{}, std::integral_constant<int, NPTL>{}, std::integral_constant<int, NTWH>{},
                           std::integral_constant<int, G::rows_h(l)>{}, std::integral_constant<int, EML>{}, IN_REG, H16, H8, wbl, NHT, htile,
                           (l & 1) ? wr1 : wr0, accl, PartBelow{}, g_c, ADD, 0);
            }
            if constexpr (l == 1) IAF_FSTAMP(7);
            if constexpr (l == DEPTH - 1) load_final_operands();
                preload_after(std::integral_constant<int, l>{});
            hidden_epilogue(std::integral_constant<int, NPTL>{}, std::integral_constant<int, G::rows_h(l)>{},
                            std::integral_constant<int, EML>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, NTWH>{}, htile, accl,
                            bil, OUT_REG, p.hsave[l], p.border[l], false);
            if constexpr (l == 1) IAF_FSTAMP(12);
        });
        __syncthreads();
    });
    IAF_FSTAMP(3);
    raise_range();

    // ---- output pair: packed tiles (m_0, s_0, m_1, s_1, ...), partial sums -> exchange buffer [K part][pixel][2 n_z] -------
    // (the operands of the final transform are fetched first: they travel while the output pair is multiplied)
    float* xbuf = (float*)(smem + (size_t)G::XB_OFF * 16);
    xch_next_epoch_b();
    if constexpr (PAIR) {
        f32x4 acco[NPTO][NTWO];
        constexpr int LAST_REG = ((DEPTH - 1) & 1) ? G::HREG1 : G::HREG0;
        static_for<2>([&](auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            if (opar != PAR) return;
            // the input channels of the own half (they are in LDS: this workgroup's epilogue wrote them) ...
            conv_phase(std::integral_constant<int, RDO>{}, std::integral_constant<int, NPTO>{}, std::integral_constant<int, NTWO>{},
                       std::integral_constant<int, R>{}, std::integral_constant<int, (1 << NPTO) - 1>{}, LAST_REG,
                       H16, H8, wbo, 2 * NZT, otile, wro, acco, PairPart<NPO, PAR>{}, std::integral_constant<int, 0>{}, SET, opair_own);
            static_for<RDO>([&](auto i) {
                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWO * NPL>{}, wro[decltype(i)::value], wbo, 2 * NZT,
                          otile, PairPart<NPR, PAR>{}, decltype(i)::value, ALL, opair_rest);
            });
            IAF_FSTAMP(29);
            __syncthreads();                                      // ... then the partner's, which its helper waves sent and ours put into LDS
            IAF_FSTAMP(21);
            conv_phase(std::integral_constant<int, RDO>{}, std::integral_constant<int, NPTO>{}, std::integral_constant<int, NTWO>{},
                       std::integral_constant<int, R>{}, std::integral_constant<int, (1 << NPTO) - 1>{}, LAST_REG,
                       H16, H8, wbo, 2 * NZT, otile, wro, acco, PairPart<NPR, PAR>{}, std::integral_constant<int, 0>{}, ADD, opair_rest);
        });
        IAF_FSTAMP(4);
        float* mine = xbuf + opar * (R * W * G::XB_STRIDE);       // the two waves of a tile leave their K parts side by side
#pragma unroll
        for (int q = 0; q < NPTO; ++q) {
            const int pix = q * 16 + pl;
            if (pix >= R * W) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[pix * G::XB_STRIDE + otile[0] * 16 + 4 * kk + r] = acco[q][0][r];
        }
    } else {
        f32x4 acco[NPTO][NTWO];
        constexpr int LAST_REG = ((DEPTH - 1) & 1) ? G::HREG1 : G::HREG0;
        conv_phase(std::integral_constant<int, RDO>{}, std::integral_constant<int, NPTO>{}, std::integral_constant<int, NTWO>{},
                   std::integral_constant<int, R>{}, std::integral_constant<int, (1 << NPTO) - 1>{}, LAST_REG,
                   H16, H8, wbo, 2 * NZT, otile, wro, acco, PartO1{}, std::integral_constant<int, 0>{}, SET, 0);
        if constexpr (XCH) {
            // (HOUT: the helper takes the odd steps; 10 steps here and tap (+1,+1)'s 5 there was no faster: 48.8 k against 48.5 k cycles)
            typedef std::conditional_t<G::HOUT, BelowPar<NPAIR_H, 0>, PartBelow> PartOB;
            static_for<RDO>([&](auto i) {
                ring_load(std::integral_constant<int, 0>{}, std::integral_constant<int, NTWO * NPL>{}, wro[decltype(i)::value], wbo, 2 * NZT,
                          otile, PartOB{}, decltype(i)::value, ALL, 0);
            });
            IAF_FSTAMP(29);
            __syncthreads();                                      // ... and the row below of the last hidden layer
            IAF_FSTAMP(21);
            conv_phase(std::integral_constant<int, RDO>{}, std::integral_constant<int, NPTO>{}, std::integral_constant<int, NTWO>{},
                       std::integral_constant<int, R>{}, std::integral_constant<int, (1 << NPTO) - 1>{}, LAST_REG,
                       H16, H8, wbo, 2 * NZT, otile, wro, acco, PartOB{}, std::integral_constant<int, 0>{}, ADD, 0);
        }
        IAF_FSTAMP(4);
        float* mine = xbuf;
#pragma unroll
        for (int j = 0; j < NTWO; ++j)
#pragma unroll
            for (int q = 0; q < NPTO; ++q) {
                const int pix = q * 16 + pl;
                if (pix >= R * W) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[pix * G::XB_STRIDE + otile[j] * 16 + 4 * kk + r] = acco[q][j][r];
            }
    }
    __syncthreads();
    IAF_FSTAMP(13);

    // ---- affine transform, log-det term, KL elements (tf_train.py:56-75), NCHW stores coalesced along the rows -------
    float klv[NEL];              // (0 for rows past the image bottom and surplus lanes)
#pragma unroll
    for (int e = 0; e < NEL; ++e) klv[e] = 0.f;
    const int fcb = PAIR ? half * NZF : 0;
#pragma unroll
    for (int e = 0; e < NEL; ++e) {
        const int idx = tid + e * 256;
        if (idx >= NZF * R * W) continue;
        const int c = fcb + idx / (R * W), pix = idx % (R * W);
        const int row = pix / W;
        if (r0 + row >= H) continue;
        const unsigned gi = 4u * (unsigned)(c * HW + gpix(r0 + row, pix - row * W));     // byte offset inside image b
        const int cm = (c >> 4) * 32 + (c & 15);
        float m_raw = fb[e][0], s_raw = fb[e][1];
        if constexpr (BORDER) {                                  // the border channel of the output pair's own input
            const float* bt = p.border[DEPTH];
            const int col = pix - row * W;
            const bool last_row = r0 + row == H - 1, c0 = col == 0, cl = col == W - 1;
            const float w1 = cl ? 1.f : 0.f, w2 = (last_row || c0) ? 1.f : 0.f, w3 = last_row ? 1.f : 0.f, w4 = (last_row || cl) ? 1.f : 0.f;
            m_raw += w1 * bt[cm] + w2 * bt[2 * NZ + cm] + w3 * bt[4 * NZ + cm] + w4 * bt[6 * NZ + cm];
            s_raw += w1 * bt[cm + 16] + w2 * bt[2 * NZ + cm + 16] + w3 * bt[4 * NZ + cm + 16] + w4 * bt[6 * NZ + cm + 16];
        }
        m_raw += xbuf[pix * G::XB_STRIDE + cm];
        s_raw += xbuf[pix * G::XB_STRIDE + cm + 16];
        if constexpr (XK2) {                                     // (the other K part)
            m_raw += xbuf[(R * W + pix) * G::XB_STRIDE + cm];
            s_raw += xbuf[(R * W + pix) * G::XB_STRIDE + cm + 16];
        }
        if (p.mode == MODE_RAW) {
            stf(p.out0 + img_z, gi, m_raw);
            stf(p.out1 + img_z, gi, s_raw);
        } else if (p.mode == MODE_IAF) {
            const float m = m_raw * 0.1f, s = s_raw * 0.1f;        // tf_train.py:70
            stf(p.out0 + img_z, gi, (fz[e] - m) * __expf(-s));     // tf_train.py:71: (z - m) / exp(s); x * exp(-s) instead of the
                                                                   // 10-instruction IEEE division: <= 2 ulp from it, 1e-7 relative
            stf(p.out1 + img_z, gi, s);                            // tf_train.py:72
        } else if (p.mode == MODE_INVERSE) {
            const float m = m_raw * 0.1f, s = s_raw * 0.1f;
            stf(p.out0 + img_z, gi, fz[e] * __expf(s) + m);
            stf(p.out1 + img_z, gi, s);
        } else {
            const float m = m_raw * 0.1f, s = s_raw * 0.1f;
            const float mean = fq[e][0];                            // tf_train.py:57
            const float logvar = 2.f * fq[e][1];
            const float z0 = mean + __expf(0.5f * logvar) * fq[e][2];                             // :63
            const float d0 = z0 - mean;
            float logqs = -0.5f * (1.8378770664093453f + logvar + d0 * d0 * __expf(-logvar));    // :68 (x / exp(v) as x * exp(-v))
            const float zz = (z0 - m) * __expf(-s);                                               // :71
            logqs += s;                                                                           // :72
            const float plv = 2.f * fq[e][4];                                                     // :56
            const float d1 = zz - fq[e][3];
            const float logps = -0.5f * (1.8378770664093453f + plv + d1 * d1 * __expf(-plv));     // :73
            stf(p.out0 + img_z, gi, zz);
            if (p.out1) stf(p.out1 + img_z, gi, s);
            klv[e] = logqs - logps;                                                               // :75
            if (p.kl_elem) stf(p.kl_elem + img_z, gi, klv[e]);
        }
    }
    // ---- the block's KL reductions start here (tf_train.py:77, sum over H, W): a channel's R*W pixels are R*W consecutive
    // lanes of one wave -> butterfly sum, one 4-byte store per (row block, channel); see StepP::kl_part
    if (p.mode == MODE_POSTERIOR && p.kl_part) {
        constexpr int RW = R * W;
        static_assert((RW & (RW - 1)) == 0 && RW <= 64 && 256 % RW == 0, "a channel's pixels are RW consecutive lanes of one wave");
#pragma unroll
        for (int e = 0; e < NEL; ++e) {
            float a = klv[e];
#pragma unroll
            for (int o = RW / 2; o > 0; o >>= 1) a += __shfl_xor(a, o);
            const int idx = tid + e * 256;
            if (idx < NZF * RW && (idx & (RW - 1)) == 0) {
                float* q = p.kl_part + ((size_t)b * p.nrb + rbk) * NZ + fcb + idx / RW;
                if (HELP && p.fin_ctl) __hip_atomic_store(q, a, __ATOMIC_RELAXED, XSCOPE);       // (read by another workgroup, below)
                else *q = a;
            }
        }
    }
    IAF_FSTAMP(5);
    if constexpr (HELP) {
        // the in-launch free-bits reductions (fin_helper): the compute waves only keep the barriers company
        if (p.fin_ctl && p.mode == MODE_POSTERIOR) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // this wave's partial sums are in memory
            __syncthreads();
            __syncthreads();
            if (!*(volatile unsigned*)smem) return;                                   // not the last workgroup
            __syncthreads();
        }
    }
#undef IAF_FSTAMP
}
