// pbd_kernels_conv.hip -- filter-bank correlation over every pyramid level (gfx950).
//
// Replaces SpatialConvolutionEngine::pdf / convolve (reference src/SpatialConvolutionEngine.cpp:
// 70-124) whose arithmetic is cv::Filter2D (src/filter.cpp:3879-3924): per channel the taps are
// accumulated in raster order from 0 with separately rounded multiply and add, then the channel
// sums are added in channel order onto a zero response.  EXACT mode reproduces that sequence per
// output element bit for bit (skipped zero-weight taps and the `0 +` of the first tap only affect
// the sign of an intermediate zero, which cannot reach the response: it starts at +0 and x + (+-0)
// == x).  FMA mode fuses multiply and add (scores within 1e-4, not bit-identical).
//
// Two kernels: k_conv3 (float, 5 x 5 filters -- every known model: strip-sequence tiles staged channel-planar in LDS,
// weights through the scalar unit, packed multiply / add) and k_conv_generic (any filter size 1..7, T = float or double).
// Out-of-image cells are materialised with the reference's constant border (0, but 1 for the last channel:
// src/SpatialConvolutionEngine.cpp:147-156).  Compiled with -ffp-contract=off.
#include "pbd_internal.h"

#include <algorithm>
#include <type_traits>

namespace pbd {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------------------------
// The exact / FMA 5 x 5 convolution (k_conv3).
//
// TILE.  Every level is cut into STRIPS of four rows (a lane computes four vertically adjacent outputs: its 8 x 5 input
// window of one channel sits in registers).  The strips of all levels of all frames of the launch, each taken left to
// right, form ONE sequence of positions; a tile is 64 consecutive positions of it -- up to three SEGMENTS, each a run
// inside one strip (pbd_internal.h: ConvSegTile, built on the host).  A tile therefore runs over the right edge of a
// level into its next strip, over the last strip of a level into the next level, and over the last level of a frame into
// the next frame, instead of leaving lanes idle at every edge: 569 tiles per 640 x 480 frame, where rectangular 256-cell
// tiles of three shapes needed 617 and strips wrapping inside one level 597 (549.7 would be no waste at all; 563 is what
// rounding every level to whole strips costs).  The segments' haloed patches lie side by side in the same 8 LDS rows
// (76 columns = 64 + 3 x 4), so the row pitch is uniform and the channel loop knows nothing about segments.
//
// WEIGHTS come through the scalar unit: s_load into SGPRs, used as the scalar operand of v_pk_mul_f32 -- no LDS slices,
// no broadcast reads into 80 VGPRs (round 2).  128 VGPRs, four waves per SIMD (two workgroups of eight waves per CU).
// Per unit the weights are [32 channels][25 taps][QL] floats (QL = filters of the unit: 2, 4, 6 or 8), so the 5 x QL
// weights of a tap row are contiguous and arrive in two to four wide scalar loads whatever QL is (with a fixed stride of
// 8 a six-filter unit needed ten narrow ones per tap row, and a wave may only have 15 LDS / scalar requests in flight:
// 37.4 -> 41.7 ms).
//
// WORK ITEMS.  With eight waves the 19.5 groups of eight filters of the person model do not divide evenly, so the host
// cuts the bank into UNITS of 8 or 6 filters (156 = 6 x 8 + 18 x 6: a wave takes 8 + 6 + 6 or 6 + 6 + 6) handed out
// largest first through an LDS counter.
//
// ARITHMETIC: as the reference -- see the head of this file; the first tap of a channel starts the channel sum without
// the reference's `0 +` (the two differ only when the product is -0, a -0 can only survive as a channel sum of -0, and
// the response it is added to starts at +0: x + (+-0) == x).
typedef const float __attribute__((address_space(4))) cfloat1;
typedef const v2f __attribute__((address_space(4))) cfloat2;

constexpr int kSegPW = 64 + 4 * kConvMaxSeg, kSegPH = 8, kSegPlane = (kSegPH * kSegPW) | 1;   // 76 x 8 cells, odd plane stride

template <bool FMA, int NW>
__device__ __forceinline__ void conv_tile3(const ConvParams &p, const float *__restrict__ wts, const float *__restrict__ featp,
                                           float *__restrict__ respp, float *sm, int *next_u, int *geo)
{
    constexpr int K = 5, P = 4, a = K / 2, PW = kSegPW, PH = kSegPH, PLANE = kSegPlane;
    const int t = threadIdx.x;
    // per-segment geometry (wave-uniform: SGPRs): level size, first cell of (frame, level), LDS column / lane where the
    // segment starts
    struct Geom {
        int nseg, nlanes;
        int len[kConvMaxSeg], W[kConvMaxSeg], H[kConvMaxSeg], x0[kConvMaxSeg], y0[kConvMaxSeg], cb[kConvMaxSeg], lb[kConvMaxSeg];
        long long cell[kConvMaxSeg];
    };
    auto load_geom = [&]() {
        const ConvSegTile T = p.segtiles[blockIdx.x];
        Geom G;
        G.nseg = T.nseg;
        int cb = 0, lb = 0;
#pragma unroll
        for (int k = 0; k < kConvMaxSeg; ++k) {
            const bool live = k < T.nseg;
            const LevelDesc d = p.lv[live ? T.seg[k].level : 0];
            G.len[k] = live ? T.len[k] : 0;
            G.W[k] = live ? d.cols : 0; G.H[k] = live ? d.rows : 0;
            G.x0[k] = T.seg[k].x0; G.y0[k] = 4 * T.seg[k].strip;
            G.cell[k] = (long long)(p.frame0 + T.seg[k].frame) * p.cell_per_frame + d.cell_off;
            G.cb[k] = cb; G.lb[k] = lb;
            cb += live ? T.len[k] + K - 1 : 0;
            lb += live ? T.len[k] : 0;
        }
        G.nlanes = lb;
        return G;
    };
    // Channel 31.  This library's HOG writes +0 there in every cell, and the convolution's border value for that channel is
    // 1 (src/SpatialConvolutionEngine.cpp:147-156): a tap contributes w * 1 = w when it falls outside the image and +-0 inside,
    // so the channel's sum is the ORDERED sum of the weights of the out-of-image taps -- a function of how many rows / columns
    // of the window stick out on each side (81 cases, tabulated per filter on the host in the reference's tap order) -- and it
    // is the LAST term added to the response (:85-93).  Exact mode therefore runs 31 channels and, for lanes whose window
    // leaves the image, adds the tabulated sum afterwards: the same rounding sequence, 1/32 less work.  (With features the
    // caller supplied, c31_zero is false and the channel is computed like the others; so it is in FMA mode, kept literal.)
    const bool skip31 = !FMA && p.c31_zero;
    bool interior = true;                                    // every window of the tile lies inside its level
    const float *sp0;
    const Geom G = load_geom();
    {
    const int lane0 = t & 63;
#pragma unroll
    for (int k = 0; k < kConvMaxSeg; ++k)
        if (k < G.nseg) interior = interior && G.x0[k] >= a && G.x0[k] + G.len[k] + a <= G.W[k] && G.y0[k] >= a && G.y0[k] + P + a <= G.H[k];
    {   // the lane's LDS window
        const int k = (lane0 >= G.lb[1] && G.nseg > 1 ? 1 : 0) + (lane0 >= G.lb[2] && G.nseg > 2 ? 1 : 0);
        sp0 = sm + (k == 0 ? G.cb[0] - G.lb[0] : k == 1 ? G.cb[1] - G.lb[1] : G.cb[2] - G.lb[2]) + lane0;
    }
    // what the stores need after the channel loop goes through LDS (8 ints per segment + the two lane boundaries): kept in
    // SGPRs it would have to live through a loop that already holds 80 of them for the weights
    if (t < kConvMaxSeg) {
        int *g = geo + 8 * t;
        g[0] = G.W[t]; g[1] = G.H[t]; g[2] = G.x0[t] - G.lb[t]; g[3] = G.y0[t];
        g[4] = (int)(unsigned)(G.cell[t] & 0xffffffffll); g[5] = (int)(G.cell[t] >> 32);
        g[6] = t + 1 < G.nseg ? G.lb[t + 1] : 64;          // first lane of the NEXT segment (64: none)
        g[7] = G.nlanes;
    }
    {   // stage the haloed patches channel-planar: 8 lanes read the 128-byte cell of a position as 4 channels each, NW*8
        // cells per pass, in batches of UB independent loads; each lane then scatters its 4 channels to their planes
        constexpr int CPP = NW * 8, NIT = (PH * PW + CPP - 1) / CPP, UB = 5;
        const int c4 = t & 7, cell0 = t >> 3;
        const v4f border = (c4 == 7) ? v4f{0.0f, 0.0f, 0.0f, 1.0f} : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        for (int it0 = 0; it0 < NIT; it0 += UB) {
            v4f v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int ci = (it0 + u) * CPP + cell0;
                const int cy = ci / PW, cx = ci - cy * PW;
                const int k = (cx >= G.cb[1] && G.nseg > 1 ? 1 : 0) + (cx >= G.cb[2] && G.nseg > 2 ? 1 : 0);
                const int W = k == 0 ? G.W[0] : k == 1 ? G.W[1] : G.W[2];
                const int H = k == 0 ? G.H[0] : k == 1 ? G.H[1] : G.H[2];
                const int gx = (k == 0 ? G.x0[0] - G.cb[0] : k == 1 ? G.x0[1] - G.cb[1] : G.x0[2] - G.cb[2]) + cx - a;
                const int gy = (k == 0 ? G.y0[0] : k == 1 ? G.y0[1] : G.y0[2]) + cy - a;
                const long long cell = k == 0 ? G.cell[0] : k == 1 ? G.cell[1] : G.cell[2];
                const int len = k == 0 ? G.len[0] : k == 1 ? G.len[1] : G.len[2];
                const int lx = cx - (k == 0 ? G.cb[0] : k == 1 ? G.cb[1] : G.cb[2]);
                v[u] = border;
                if (it0 + u < NIT && ci < PH * PW && lx < len + K - 1 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v[u] = *reinterpret_cast<const v4f *>(featp + ((size_t)cell + (size_t)gy * W + gx) * 32 + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int ci = (it0 + u) * CPP + cell0;
                if (it0 + u < NIT && ci < PH * PW) {
                    float *dst = sm + (c4 * 4) * PLANE + ci;
                    dst[0] = v[u].x; dst[PLANE] = v[u].y; dst[2 * PLANE] = v[u].z; dst[3 * PLANE] = v[u].w;
                }
            }
        }
    }
    }
    const int u0 = blockIdx.y * p.units_per_block;
    const int u1 = min(u0 + p.units_per_block, p.nunits);
    if (t == 0) *next_u = u0;
    __syncthreads();

    const int lane = t & 63;

    auto run_unit = [&](auto ql_tag, const int u, const int f0) {
        constexpr int QL = decltype(ql_tag)::value, QH = QL / 2;
        // the unit's weights: wave-uniform addresses in the constant address space -> s_load
        cfloat1 *wc = (cfloat1 *)(wts + p.unit_woff[u]);
        v2f r[P][QH];
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int q = 0; q < QH; ++q) r[pp][q] = v2f{0.0f, 0.0f};
        // Software-pipelined sliding window.  A lane's 4 output rows see input rows 0..7 of its column; tap row i uses rows
        // i..i+3.  Every input row is read from LDS ONCE per channel, and the reads of a stage (one new row of this channel,
        // or the next channel's first four; the 5 x QL weights of the next tap row) are issued before the packed operations
        // of the stage, into registers the stage does not touch -- a wave never waits for its own reads.
        float Ft[P + K - 1][K];
        // SGPRs.  Taps 0..2 of a tap row are fetched one stage ahead (two buffers); taps 3..4 at the start of the stage that
        // uses them, three taps (about 100 packed operations per wave, four waves interleaved) before their first use -- 64
        // scalar registers instead of 80, which is what lets the loop keep its other scalars out of spill lanes
        constexpr int KA = 3;
        v2f Wa[2][KA][QH], Wb[K - KA][QH];
        // LDS addressing: ONE running base (the lane's window in the plane of the pair's first channel, advanced by two planes
        // per iteration); everything else is an immediate offset -- the planes span 76 KB, more than an offset field reaches,
        // and per-channel bases computed from the channel index cost a dozen address registers the loop does not have
        const float *spc = sp0;
        auto load_row = [&](const float *base, int rr) {
            const float *sp = base + rr * PW;
#pragma unroll
            for (int j = 0; j < K; ++j) Ft[rr][j] = sp[j];
        };
        auto load_w = [&](int buf, int c, int i) {          // taps 0..2 of tap row (c, i): for the NEXT stage
#pragma unroll
            for (int j = 0; j < KA; ++j)
#pragma unroll
                for (int q = 0; q < QH; ++q) Wa[buf][j][q] = *(cfloat2 *)(wc + ((c * K + i) * K + j) * QL + 2 * q);
        };
        auto load_wb = [&](int c, int i) {                  // taps 3..4 of tap row (c, i): for THIS stage
#pragma unroll
            for (int j = KA; j < K; ++j)
#pragma unroll
                for (int q = 0; q < QH; ++q) Wb[j - KA][q] = *(cfloat2 *)(wc + ((c * K + i) * K + j) * QL + 2 * q);
        };
        v2f s[P][QH];
        auto zero_s = [&]() {
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int q = 0; q < QH; ++q) s[pp][q] = v2f{0.0f, 0.0f};
        };
        auto add_s = [&]() {
#pragma unroll
            for (int pp = 0; pp < P; ++pp)
#pragma unroll
                for (int q = 0; q < QH; ++q) r[pp][q] = r[pp][q] + s[pp][q];
        };
        auto comp = [&](int wbi, int i) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                v2f w[QH];
#pragma unroll
                for (int q = 0; q < QH; ++q) w[q] = j < KA ? Wa[wbi][j][q] : Wb[j - KA][q];
                // products, then their additions, PB pairs at a time: with four waves on a SIMD the other waves' instructions
                // sit between a product and its addition, so two pairs are enough distance -- and two pairs of temporaries
                // (not four) keep the loop inside 128 registers without a spill
                constexpr int PB = 2;
                const bool first = (i == 0 && j == 0);
                const v2f zero2 = v2f{0.0f, 0.0f};
#pragma unroll
                for (int pp = 0; pp < P; ++pp) {
                    const v2f f = v2f{Ft[pp + i][j], Ft[pp + i][j]};
                    if (FMA) {
#pragma unroll
                        for (int q = 0; q < QH; ++q) s[pp][q] = __builtin_elementwise_fma(w[q], f, first ? zero2 : s[pp][q]);
                    } else {
#pragma unroll
                        for (int q0 = 0; q0 < QH; q0 += PB) {
                            v2f tq[PB];
#pragma unroll
                            for (int q = q0; q < q0 + PB && q < QH; ++q) tq[q - q0] = w[q] * f;
#pragma unroll
                            for (int q = q0; q < q0 + PB && q < QH; ++q) s[pp][q] = first ? tq[q - q0] : s[pp][q] + tq[q - q0];
                        }
                    }
                }
            }
        };
        // The stages live in one basic block; to keep the compiler from sinking the packed operations below the loads of
        // later stages, the accumulators pass through an empty asm with a memory clobber at both ends of every stage (the
        // loads cannot cross it, the operations are tied to it through their operands).
#define PBD_PIN3()                                                                                                         \
    do {                                                                                                                   \
        if constexpr (QL == 8)                                                                                             \
            asm volatile("" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]),    \
                         "+v"(s[1][2]), "+v"(s[1][3]), "+v"(s[2][0]), "+v"(s[2][1]), "+v"(s[2][2]), "+v"(s[2][3]),         \
                         "+v"(s[3][0]), "+v"(s[3][1]), "+v"(s[3][2]), "+v"(s[3][3])::"memory");                            \
        else if constexpr (QL == 6)                                                                                        \
            asm volatile("" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[1][2]),    \
                         "+v"(s[2][0]), "+v"(s[2][1]), "+v"(s[2][2]), "+v"(s[3][0]), "+v"(s[3][1]), "+v"(s[3][2])::"memory"); \
        else if constexpr (QL == 4)                                                                                        \
            asm volatile("" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[2][0]), "+v"(s[2][1]),    \
                         "+v"(s[3][0]), "+v"(s[3][1])::"memory");                                                          \
        else                                                                                                               \
            asm volatile("" : "+v"(s[0][0]), "+v"(s[1][0]), "+v"(s[2][0]), "+v"(s[3][0])::"memory");                       \
    } while (0)
#define PBD_STAGE3(LOADS, WB, I)                 \
    do {                                         \
        LOADS;                                   \
        PBD_PIN3();                              \
        __builtin_amdgcn_sched_barrier(0);       \
        comp(WB, I);                             \
        PBD_PIN3();                              \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#pragma unroll
        for (int rr = 0; rr < P; ++rr) load_row(spc, rr);
        load_w(0, 0, 0);
        zero_s();                      // defined values for the first pin; every channel starts its own sum
        // one channel: rows 4..7 arrive during stages 0..3, the next channel's rows 0..3 during stage 4 (rows 0..3 are dead
        // by then); the tap rows alternate between the two weight buffers, so two channels make the pattern repeat.  Nothing
        // is conditional (the last iteration re-reads channel 31)
#define PBD_CHANNEL3(c, cn, SPC, SPN, B0, B1)                                                                                      \
    do {                                                                                                                          \
        PBD_STAGE3(load_row((SPC), 4); load_wb((c), 0); load_w(B1, (c), 1), B0, 0);                                               \
        PBD_STAGE3(load_row((SPC), 5); load_wb((c), 1); load_w(B0, (c), 2), B1, 1);                                               \
        PBD_STAGE3(load_row((SPC), 6); load_wb((c), 2); load_w(B1, (c), 3), B0, 2);                                               \
        PBD_STAGE3(load_row((SPC), 7); load_wb((c), 3); load_w(B0, (c), 4), B1, 3);                                               \
        PBD_STAGE3(load_row((SPN), 0); load_row((SPN), 1); load_row((SPN), 2); load_row((SPN), 3); load_wb((c), 4); load_w(B1, (cn), 0), B0, 4); \
        add_s();                                                                                                                  \
    } while (0)
        // skip31: 15 pairs and channel 30 alone -- as an epilogue, not as a break inside the loop body, which stays one
        // basic block
        const int cpairs = skip31 ? 30 : 32;
#pragma clang loop unroll(disable)
        for (int c = 0; c < cpairs; c += 2) {
            const int c2 = min(c + 2, 31);
            const float *spn = spc + (c2 - c) * PLANE;       // c2 - c: 2, or 1 in the last iteration (wave-uniform)
            PBD_CHANNEL3(c, c + 1, spc, spc + PLANE, 0, 1);
            PBD_CHANNEL3(c + 1, c2, spc + PLANE, spn, 1, 0);
            spc += 2 * PLANE;
        }
        if (skip31) PBD_CHANNEL3(30, 31, spc, spc + PLANE, 0, 1);
#undef PBD_CHANNEL3
#undef PBD_STAGE3
#undef PBD_PIN3
        // the lane's outputs: (segment, column) -> level coordinates and the response pointer, formed here, after the loop
        int ln = threadIdx.x & 63;
        asm volatile("" : "+v"(ln));
        const int k = (ln >= geo[6] ? 1 : 0) + (ln >= geo[14] ? 1 : 0);
        const int *g = geo + 8 * k;
        const int W = g[0], H = g[1];
        const int x = g[2] + ln, y = g[3];
        const long long cell = (long long)(unsigned)g[4] | ((long long)g[5] << 32);
        const size_t HW = (size_t)H * W;
        float *respg = respp + (size_t)cell * p.F + (size_t)y * W + x;
        const bool live = ln < g[7];
        if (skip31 && !interior && live) {
            // channel 31's term for the windows that leave the image: rows / columns outside on each side, clipped to 0..2
            const int left = min(max(a - x, 0), a), right = min(max(x + a - (W - 1), 0), a);
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                const int yy = y + pp;
                const int top = min(max(a - yy, 0), a), bot = min(max(yy + a - (H - 1), 0), a);
                const int cs = ((top * 3 + bot) * 3 + left) * 3 + right;
                if (cs != 0 && yy < H) {
                    const float *tab = p.c31tab + (size_t)cs * p.c31stride + f0;
#pragma unroll
                    for (int q = 0; q < QL; ++q) r[pp][q / 2][q & 1] += tab[q];
                }
            }
        }
        if (f0 + QL <= p.nf && p.fmap == nullptr) {
            // a full unit stores without per-filter branches: one block of QL independent stores per row
            float *rg = respg + (size_t)f0 * HW;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                if (live && y + pp < H) {
#pragma unroll
                    for (int q = 0; q < QL; ++q) rg[(size_t)q * HW + (size_t)pp * W] = r[pp][q / 2][q & 1];
                }
            }
        } else if (live) {
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                if (y + pp < H) {
#pragma unroll
                    for (int q = 0; q < QL; ++q) {
                        const int f = f0 + q;
                        if (f < p.nf) respg[(size_t)(p.fmap ? p.fmap[f] : f) * HW + (size_t)pp * W] = r[pp][q / 2][q & 1];
                    }
                }
            }
        }
    };
    for (;;) {
        int u = 0;
        if (lane == 0) u = atomicAdd(next_u, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= u1) break;
        const int f0 = p.unit_f0[u], ql = p.unit_ql[u];       // wave-uniform (scalar loads)
        if (ql == 8) run_unit(std::integral_constant<int, 8>{}, u, f0);
        else if (ql == 6) run_unit(std::integral_constant<int, 6>{}, u, f0);
        else if (ql == 4) run_unit(std::integral_constant<int, 4>{}, u, f0);
        else run_unit(std::integral_constant<int, 2>{}, u, f0);
    }
}

template <bool FMA, int NW>
__global__ __launch_bounds__(NW * 64, (2 * NW + 3) / 4) void k_conv3(ConvParams p, const float *__restrict__ wts, const float *__restrict__ featp,
                                                   float *__restrict__ respp)
{
    __shared__ __attribute__((aligned(16))) float sm[32 * kSegPlane + 3];
    __shared__ int next_u;
    __shared__ __attribute__((aligned(16))) int geo[8 * kConvMaxSeg];
    conv_tile3<FMA, NW>(p, wts, featp, respp, sm, &next_u, geo);
}

// generic kernel: any filter size, any real type R (the reference's T=double instantiation runs here);
// one output pixel per thread, weights [channel][tap][Fpad] read with wave-uniform vector loads
// KT: the filter side at compile time (5: the tap loops unroll, so the 25 wave-uniform weight loads and LDS reads of a channel are
// issued in batches instead of one wait per tap) or 0 for any size
template <typename R, bool FMA, int KT>
__global__ __launch_bounds__(256) void k_conv_generic(ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    R *smd = reinterpret_cast<R *>(smraw);
    constexpr int TW = kConvTW, TH = kConvTH, Q = kConvQ;
    const int K = KT ? KT : p.ksize;
    const int PW = TW + K - 1, PH = TH + K - 1;
    const int PLANE = (PH * PW) | 1;
    const ConvTile tile = p.tiles[blockIdx.x];
    const int frame = p.frame0 + blockIdx.z;
    const LevelDesc d = p.lv[tile.level];
    const int H = d.rows, W = d.cols;
    const int a = K / 2;
    const int t = threadIdx.x;
    const R *feat = static_cast<const R *>(p.feat) + ((size_t)frame * p.cell_per_frame + d.cell_off) * 32;
    const R *wts = static_cast<const R *>(p.wts);
    // CB channels of the haloed tile at a time: all 32 when they fit LDS (staged once for every filter group), fewer for
    // large filters / T = double (then every group restages its channel blocks: a few loads per thread against CB * K * K * 8
    // multiply-adds).  The per-filter order of the sums does not depend on CB.
    const int CB = p.cblock;
    auto stage = [&](int c0) {
        for (int idx = t; idx < PH * PW * CB; idx += 256) {
            const int ci = idx / CB, c = c0 + idx - ci * CB;
            const int cy = ci / PW, cx = ci - cy * PW;
            const int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
            R v = (c == 31) ? (R)1 : (R)0;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = feat[((size_t)gy * W + gx) * 32 + c];
            smd[(c - c0) * PLANE + ci] = v;
        }
    };
    if (CB == 32) { stage(0); __syncthreads(); }
    const int px = t & 31, py = t >> 5;
    const int x = tile.x0 + px, y = tile.y0 + py;
    const bool valid = (x < W) && (y < H);
    const int ngroups = p.Fpad / Q;
    const int g0 = blockIdx.y * p.groups_per_block;
    const int g1 = min(g0 + p.groups_per_block, ngroups);
    const size_t HW = (size_t)H * W;
    R *resp = static_cast<R *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + (size_t)y * W + x;
    for (int g = g0; g < g1; ++g) {
        R r[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) r[q] = (R)0;
        for (int c = 0; c < 32; ++c) {
            if (CB != 32 && c % CB == 0) { __syncthreads(); stage(c); __syncthreads(); }
            const R *sp = smd + (c % CB) * PLANE + py * PW + px;
            const R *wp = wts + (size_t)c * (K * K) * p.Fpad + g * Q;
            R s[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) s[q] = (R)0;
            // the weights of a tap are wave-uniform and read-only: through the constant address space they come in on the scalar
            // unit (s_load into SGPRs, used as the scalar operand of the multiply) instead of as 64-lane vector loads of one address
            typedef const R __attribute__((address_space(4))) cR;
            auto tap = [&](int i, int j) {
                const R f = sp[i * PW + j];
                cR *w = (cR *)(wp + (size_t)(i * K + j) * p.Fpad);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    if (FMA) s[q] = __builtin_fma(w[q], f, s[q]);
                    else s[q] = s[q] + w[q] * f;
                }
            };
            if (KT) {
#pragma unroll
                for (int i = 0; i < (KT ? KT : 1); ++i)
#pragma unroll
                    for (int j = 0; j < (KT ? KT : 1); ++j) tap(i, j);
            } else {
                for (int i = 0; i < K; ++i)
                    for (int j = 0; j < K; ++j) tap(i, j);
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) r[q] = r[q] + s[q];
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int f = g * Q + q;
                if (f < p.nf) resp[(size_t)(p.fmap ? p.fmap[f] : f) * HW] = r[q];
            }
        }
    }
}

int conv_occupancy(int nw)
{   // resident workgroups per CU of the exact 5x5 kernel (diagnostics)
    int n = -1;
    if (nw == kConv3NW || nw == 4) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv3<false, kConv3NW>, kConv3NW * 64, 0);
    return n;
}

template <typename R, bool FMA>
static void launch_generic(const ConvParams &p, dim3 grid, hipStream_t s)
{
    const int PW = kConvTW + p.ksize - 1, PH = kConvTH + p.ksize - 1;
    const size_t lds = (size_t)p.cblock * ((PH * PW) | 1) * sizeof(R);
    if (p.ksize == 5) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_generic<R, FMA, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PBD_LAUNCH((k_conv_generic<R, FMA, 5>), grid, dim3(256), lds, s, p);
        return;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_generic<R, FMA, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PBD_LAUNCH((k_conv_generic<R, FMA, 0>), grid, dim3(256), lds, s, p);
}

template <bool FMA, int NW>
static void launch_conv3(const ConvParams &p, hipStream_t s)
{
    const int gy = (p.nunits + p.units_per_block - 1) / p.units_per_block;
    PBD_LAUNCH((k_conv3<FMA, NW>), dim3(p.nsegtiles, gy, 1), dim3(NW * 64), 0, s, p, static_cast<const float *>(p.wts3),
               static_cast<const float *>(p.feat), static_cast<float *>(p.resp));
}

void launch_conv(const ConvParams &p, int nframes, bool f64, hipStream_t s)
{
    if (p.ntiles == 0 || p.nf == 0) return;
    const int ngroups = p.Fpad / kConvQ;
    const int gy = (ngroups + p.groups_per_block - 1) / p.groups_per_block;
    dim3 grid(p.ntiles, gy, nframes);
    if (!f64 && p.ksize == 5) {
        // float, 5 x 5: strip-sequence tiles (frames are inside the tile list), weights through the scalar unit
        if (p.nsegtiles == 0) return;
        if (p.fma) launch_conv3<true, kConv3NW>(p, s); else launch_conv3<false, kConv3NW>(p, s);
    } else if (f64) {
        if (p.fma) launch_generic<double, true>(p, grid, s); else launch_generic<double, false>(p, grid, s);
    } else {
        if (p.fma) launch_generic<float, true>(p, grid, s); else launch_generic<float, false>(p, grid, s);
    }
}

}  // namespace pbd
