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
// Mapping: one workgroup = one 256-cell tile (32 x 8, 16 x 16 or 8 x 32: the host covers each level with
// the mix that wastes the fewest lanes) of one level of one frame.  The (TW+k-1) x (TH+k-1) x 32
// channel input tile is staged once in LDS, re-laid out channel-planar so that a wave's lanes
// (consecutive x) read consecutive LDS words; out-of-image cells are materialised with the
// reference's constant border (0, but 1 for the last channel: :147-156).  Each thread owns one
// output pixel and sweeps the filters in groups of 8; the 8 weights of a (channel, tap) are
// wave-uniform and come through the scalar cache, so the inner loop is 16 VALU ops per LDS read.
// Compiled with -ffp-contract=off.
#include "pbd_internal.h"

#include <algorithm>
#include <type_traits>

namespace pbd {

// Weights are read-only for the whole launch and every address is wave-uniform: reading them
// through the constant address space lets the compiler keep them in SGPRs (s_load_dwordx8).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const v4f __attribute__((address_space(4))) cfloat4;

// K x K filters, NW waves per workgroup, tile shape S (TW = 32 >> S wide, TH = 8 << S high).  Every wave
// covers the whole tile: lane = (x, row group), 4 consecutive rows per lane, whose 8 x K input window of one channel sits in registers; the waves
// share the filter groups (handed out through an LDS counter).
// Weights: the K*K*8 weights of a (group, channel) are 800 contiguous bytes in HBM ([g][c][tap][8]).
// Each wave stages them into its own double-buffered LDS slice with one 16-byte load per lane, one
// channel ahead of use, and reads them back with broadcast ds_read_b128 (all lanes, same address), so
// the inner loop has no scalar-cache traffic and LDS waits can be counted (lgkmcnt(N)).
// Per (channel, tap): 2 broadcast LDS reads and 32 multiply-adds per lane (16 packed mul + 16 packed add).
template <int K, bool FMA, int NW, int S>
__device__ __forceinline__ void conv_tile(const ConvParams &p, const float *__restrict__ wts, const float *__restrict__ featp,
                                          float *__restrict__ respp, float *sm, int *next_g, const ConvTile tile)
{
    // S = 3: a WRAPPED tile -- 64 consecutive positions of the level's strips of four rows, i.e. up to two segments: the rest
    // of strip `tile.y0` from column `tile.x0`, then the beginning of the next strip.  Their haloed patches lie side by side
    // in the same PH rows of LDS (PW = 64 + 2 (K - 1)), so the row pitch is uniform and the channel loop does not change.
    constexpr bool WRAP = S == 3;
    constexpr int TW = WRAP ? 64 : kConvTW >> S, TH = WRAP ? 4 : kConvTH << S, Q = kConvQ, P = 4;
    static_assert(TW * TH == 256 && TH % P == 0, "a wave of 64 lanes x 4 rows covers the tile");
    constexpr int PW = WRAP ? TW + 2 * (K - 1) : TW + K - 1, PH = TH + K - 1;
    constexpr int PLANE = (PH * PW) | 1;   // odd plane stride: conflict-free staging writes
    constexpr int WCH = K * K * Q;         // weights of one (group, channel)
    constexpr int WLANES = (WCH + 3) / 4;  // lanes that stage 16 bytes each
    static_assert(WLANES <= 64, "one staging instruction per wave");
    static_assert(32 * PLANE + 3 + NW * 2 * WLANES * 4 <= 32 * 577 + 3 + NW * 2 * WLANES * 4, "LDS sized for the largest shape");

    const int frame = p.frame0 + blockIdx.z;
    const LevelDesc d = p.lv[tile.level];
    const int H = d.rows, W = d.cols;
    constexpr int a = K / 2;
    const int t = threadIdx.x;
    const float *feat = featp + ((size_t)frame * p.cell_per_frame + d.cell_off) * 32;
    // wrapped tile: `la` positions in the first segment; its patch occupies LDS columns [0, la + K - 1), the second segment's
    // the columns from there on
    const int la = WRAP ? min(W - tile.x0, 64) : 0;

    {   // stage: 8 lanes read the 128-byte cell of a position as 4 channels each, NW*8 cells per pass, in
        // batches of UB independent loads; each lane then scatters its 4 channels to their planes
        // (bank = 4*c4 + cell + 17*k mod 32: two lanes per bank, the minimum for 64 lanes)
        constexpr int CPP = NW * 8, NIT = (PH * PW + CPP - 1) / CPP, UB = 7;
        const int c4 = t & 7, cell0 = t >> 3;
        const v4f border = (c4 == 7) ? v4f{0.0f, 0.0f, 0.0f, 1.0f} : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        for (int it0 = 0; it0 < NIT; it0 += UB) {
            v4f v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int ci = (it0 + u) * CPP + cell0;
                const int cy = ci / PW, cx = ci - cy * PW;
                int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
                if (WRAP) {
                    const bool second = cx >= la + K - 1;
                    gy = 4 * (tile.y0 + (second ? 1 : 0)) + cy - a;
                    gx = second ? cx - (la + K - 1) - a : tile.x0 + cx - a;
                }
                v[u] = border;
                if (it0 + u < NIT && ci < PH * PW && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v[u] = *reinterpret_cast<const v4f *>(feat + ((size_t)gy * W + gx) * 32 + c4 * 4);
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
    const int g0 = blockIdx.y * p.groups_per_block;
    if (t == 0) *next_g = g0;
    __syncthreads();

    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool second = WRAP && lane >= la;
    const int px = WRAP ? lane + (second ? K - 1 : 0) : lane & (TW - 1);                  // LDS column of the lane's window
    const int py = WRAP ? 0 : (lane >> (WRAP ? 0 : 5 - S)) * P;
    const int x = WRAP ? (second ? lane - la : tile.x0 + lane) : tile.x0 + px;
    const int y = WRAP ? 4 * (tile.y0 + (second ? 1 : 0)) : tile.y0 + py;
    const int ngroups = p.Fpad / Q;
    const int g1 = min(g0 + p.groups_per_block, ngroups);
    const size_t HW = (size_t)H * W;
    float *resp = respp + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + (size_t)y * W + x;
    const float *sp0 = sm + py * PW + px;
    // HOG channel 31 is 0 inside the image (1 only in the constant border): a tile whose haloed patch lies inside the level
    // skips that channel (about a third of the tiles; exact mode only -- a fused multiply-add of zeros changes nothing either,
    // but the FMA mode is kept literal)
    const bool skip31 = WRAP ? (!FMA && p.c31_zero && la >= 64 && 4 * tile.y0 >= a && tile.x0 >= a && 4 * tile.y0 + TH + a <= H && tile.x0 + TW + a <= W)
                             : (!FMA && p.c31_zero && tile.y0 >= a && tile.x0 >= a && tile.y0 + TH + a <= H && tile.x0 + TW + a <= W);
    // wave-private weight slice, 16-byte aligned
    float *wbuf = sm + ((32 * PLANE + 3) & ~3) + wave * (2 * WLANES * 4);
    // lanes past the last 16-byte piece of the 800-byte weight block repeat the last piece (same address, same data): the
    // staging is then branch-free and the channel loop stays ONE basic block -- values that cross a block boundary reach the
    // packed multiplies as lone 32-bit registers and are copied into both halves of a fresh pair first
    const int wlane = lane < WLANES ? lane : WLANES - 1;

    // filter groups are handed out dynamically: the 2 x NW waves of the resident workgroups do not spread
    // evenly over the 4 SIMDs, so waves on the less loaded SIMDs take more groups
    // QL = live filters of the group: 8, or 4 for a last group whose upper half is padding (156 filters = 19 groups of
    // 8 + 4: the half group runs half the packed operations instead of multiplying zeros -- 2 % of the launch)
    auto run_group = [&](auto ql_tag, const int g) {
        constexpr int QL = decltype(ql_tag)::value, QH = QL / 2, NWV = QL / 4;
        const v4f *wsrc = reinterpret_cast<const v4f *>(wts + (size_t)g * 32 * WCH) + wlane;
        v4f wreg = wsrc[0];
        *reinterpret_cast<v4f *>(wbuf + wlane * 4) = wreg;
        v2f r[P][QH];
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int q = 0; q < QH; ++q) r[pp][q] = v2f{0.0f, 0.0f};
        // Software-pipelined sliding window.  A lane's 4 output rows see input rows 0..7 of the tile column; tap
        // row i uses rows i..i+3.  Every input row is read from LDS ONCE per channel, and the LDS reads of a stage
        // (one new row of this channel, one row of the next channel's first four, the 5 x 8 weights of the next
        // tap row: 15 instructions) are issued before the 160 packed operations of the stage, into registers the
        // stage does not touch -- a wave never waits for its own LDS reads.  Two channels per loop iteration keep
        // the buffer parity static (weights: 10 stages A B A B A | B A B A B; rows: even / odd channel).
        float Ft[P + K - 1][K];
        v4f Wt[2][K][NWV];
        auto load_row = [&](int c, int r) {
            const float *sp = sp0 + c * PLANE + r * PW;
#pragma unroll
            for (int j = 0; j < K; ++j) Ft[r][j] = sp[j];
        };
        auto load_w = [&](int buf, int c, int i) {
            const float *wcur = wbuf + (c & 1) * (WLANES * 4) + i * K * Q;
#pragma unroll
            for (int j = 0; j < K; ++j) {
#pragma unroll
                for (int h = 0; h < NWV; ++h) Wt[buf][j][h] = *reinterpret_cast<const v4f *>(wcur + j * Q + 4 * h);
            }
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
                for (int h = 0; h < NWV; ++h) {
                    const v4f wv = Wt[wbi][j][h];
                    w[2 * h] = __builtin_shufflevector(wv, wv, 0, 1);
                    w[2 * h + 1] = __builtin_shufflevector(wv, wv, 2, 3);
                }
#pragma unroll
                for (int pp = 0; pp < P; ++pp) {
                    const v2f f = v2f{Ft[pp + i][j], Ft[pp + i][j]};
                    // the first tap of a channel starts the sum.  The reference computes 0 + w*f (src/filter.cpp:3916); the
                    // exact mode takes w*f itself: the two differ only when the product is -0 (0 + -0 = +0), a -0 can only
                    // survive as a channel sum of -0, and the response it is added to starts at +0 and x + (+-0) == x
                    const bool first = (i == 0 && j == 0);
                    const v2f zero2 = v2f{0.0f, 0.0f};
                    if (FMA) {
#pragma unroll
                        for (int q = 0; q < QH; ++q) s[pp][q] = __builtin_elementwise_fma(w[q], f, first ? zero2 : s[pp][q]);
                    } else {
                        // four products, then their four additions: an addition issues 16 cycles after its product
                        v2f tq[QH];
#pragma unroll
                        for (int q = 0; q < QH; ++q) tq[q] = w[q] * f;
#pragma unroll
                        for (int q = 0; q < QH; ++q) s[pp][q] = first ? tq[q] : s[pp][q] + tq[q];
                    }
                }
            }
        };
        // stage = tap row i of a channel: rows 4..7 arrive during stages 0..3, the next channel's rows 0..3 during
        // stage 4 (rows 0..3 are dead by then); nothing is conditional (the last iteration re-reads channel 31)
        // The stages live in one basic block; to keep the compiler from sinking the packed operations below the
        // loads of later stages, the 16 accumulators pass through an empty asm with a memory clobber at both ends
        // of every stage (the loads cannot cross it, the operations are tied to it through their operands).
#define PBD_PIN()                                                                                                          \
    do {                                                                                                                   \
        if constexpr (QL == 8)                                                                                             \
            asm volatile("" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][QH - 2]), "+v"(s[0][QH - 1]), "+v"(s[1][0]), "+v"(s[1][1]),      \
                         "+v"(s[1][QH - 2]), "+v"(s[1][QH - 1]), "+v"(s[2][0]), "+v"(s[2][1]), "+v"(s[2][QH - 2]), "+v"(s[2][QH - 1]), \
                         "+v"(s[3][0]), "+v"(s[3][1]), "+v"(s[3][QH - 2]), "+v"(s[3][QH - 1])::"memory");                    \
        else                                                                                                               \
            asm volatile("" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[2][0]), "+v"(s[2][1]),    \
                         "+v"(s[3][0]), "+v"(s[3][1])::"memory");                                                          \
    } while (0)
#define PBD_STAGE(LOADS, WB, I)                  \
    do {                                         \
        LOADS;                                   \
        PBD_PIN();                               \
        __builtin_amdgcn_sched_barrier(0);       \
        comp(WB, I);                             \
        PBD_PIN();                               \
        __builtin_amdgcn_sched_barrier(0);       \
    } while (0)
#pragma unroll
        for (int rr = 0; rr < P; ++rr) load_row(0, rr);
        load_w(0, 0, 0);
        zero_s();                      // defined values for the first pin; every channel starts its own sum
        // the first channel of a pair (weights slice 0; channel c+1's weights travel HBM -> wreg -> slice 1) ...
#define PBD_EVEN_CHANNEL(c)                                                                                                       \
    do {                                                                                                                          \
        wreg = wsrc[(size_t)((c) + 1) * (WCH / 4)];                                                                               \
        PBD_STAGE(load_row((c), 4); load_w(1, (c), 1), 0, 0);                                                                     \
        PBD_STAGE(load_row((c), 5); load_w(0, (c), 2), 1, 1);                                                                     \
        PBD_STAGE(load_row((c), 6); load_w(1, (c), 3), 0, 2);                                                                     \
        *reinterpret_cast<v4f *>(wbuf + WLANES * 4 + wlane * 4) = wreg;                                                           \
        PBD_STAGE(load_row((c), 7); load_w(0, (c), 4), 1, 3);                                                                     \
        PBD_STAGE(load_row((c) + 1, 0); load_row((c) + 1, 1); load_row((c) + 1, 2); load_row((c) + 1, 3); load_w(1, (c) + 1, 0), 0, 4); \
        add_s();                                                                                                                  \
    } while (0)
        // ... and the second (slice 1; channel c+2 -> slice 0)
#define PBD_ODD_CHANNEL(c, c2)                                                                                                    \
    do {                                                                                                                          \
        wreg = wsrc[(size_t)(c2) * (WCH / 4)];                                                                                    \
        PBD_STAGE(load_row((c) + 1, 4); load_w(0, (c) + 1, 1), 1, 0);                                                             \
        PBD_STAGE(load_row((c) + 1, 5); load_w(1, (c) + 1, 2), 0, 1);                                                             \
        PBD_STAGE(load_row((c) + 1, 6); load_w(0, (c) + 1, 3), 1, 2);                                                             \
        *reinterpret_cast<v4f *>(wbuf + wlane * 4) = wreg;                                                                        \
        PBD_STAGE(load_row((c) + 1, 7); load_w(1, (c) + 1, 4), 0, 3);                                                             \
        PBD_STAGE(load_row((c2), 0); load_row((c2), 1); load_row((c2), 2); load_row((c2), 3); load_w(0, (c2), 0), 1, 4);          \
        add_s();                                                                                                                  \
    } while (0)
        // channel 31 is zero over the whole patch of an interior tile (skip31): its sum is +-0 and r + (+-0) == r, so those
        // tiles run 15 pairs and channel 30 alone -- as an epilogue, not as a break inside the loop body, which has to stay
        // one basic block (see wlane)
        const int cpairs = skip31 ? 30 : 32;
#pragma clang loop unroll(disable)
        for (int c = 0; c < cpairs; c += 2) {
            const int c2 = min(c + 2, 31);
            PBD_EVEN_CHANNEL(c);
            PBD_ODD_CHANNEL(c, c2);
        }
        if (skip31) PBD_EVEN_CHANNEL(30);
#undef PBD_EVEN_CHANNEL
#undef PBD_ODD_CHANNEL
#undef PBD_STAGE
#undef PBD_PIN
        // a full group (all but possibly the last) stores without per-filter branches: one block of 8
        // independent stores per row
        // the store addresses are formed here, after the channel loop: hoisted above it they would be carried through the
        // loop in scratch (13 spilled 64-bit pointers per lane, 1 GB of scratch writes per 64-frame launch)
        float *respg = resp;
        asm volatile("" : "+v"(respg));
        if (g * Q + QL <= p.nf && p.fmap == nullptr) {
            float *rg = respg + (size_t)(g * Q) * HW;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                if (x < W && y + pp < H) {
#pragma unroll
                    for (int q = 0; q < QL; ++q) rg[(size_t)q * HW + (size_t)pp * W] = r[pp][q / 2][q & 1];
                }
            }
        } else if (x < W) {
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                if (y + pp < H) {
#pragma unroll
                    for (int q = 0; q < QL; ++q) {
                        const int f = g * Q + q;
                        if (f < p.nf) respg[(size_t)(p.fmap ? p.fmap[f] : f) * HW + (size_t)pp * W] = r[pp][q / 2][q & 1];
                    }
                }
            }
        }
    };
    const int half_g = (p.nf % Q != 0 && p.nf % Q <= Q / 2) ? ngroups - 1 : -1;   // the group whose upper half is padding
    for (;;) {
        int g = 0;
        if (lane == 0) g = atomicAdd(next_g, 1);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g >= g1) break;
        if (g == half_g) run_group(std::integral_constant<int, Q / 2>{}, g);
        else run_group(std::integral_constant<int, Q>{}, g);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the same tile, the same arithmetic and the same stage pipeline, but the WEIGHTS COME THROUGH THE SCALAR UNIT
// (s_load_dwordx8 into SGPRs, used as the scalar operand of v_pk_mul_f32) instead of through wave-private LDS slices and
// 80 VGPRs of double-buffered broadcast reads.  That takes the kernel from 237 to <= 168 VGPRs, i.e. from two to THREE
// waves per SIMD (two workgroups of six waves per CU): a register-only stream of the packed pairs runs at 4.63 cycles per
// instruction with two waves per SIMD, 4.45 with three (tools/probes/vgpr_banks.hip), and a third wave fills the issue
// slots the other two leave while they wait (VALU was busy 89 % of the cycles with two).
// Work items: with six waves the 19.5 groups of eight filters no longer divide evenly, so the host cuts the bank into
// UNITS of 8 or 6 filters (156 = 6 x 8 + 18 x 6: every wave gets 8 + 6 + 6 + 6 when the SIMDs are evenly loaded, and the
// hand-out through the LDS counter -- larger units first -- evens it out when they are not).  Weights: [unit][32][25][8]
// floats, a unit's unused upper lanes zero and never multiplied (QL = live filters of the unit: 2, 4, 6 or 8).
typedef float v8f __attribute__((ext_vector_type(8)));
typedef const v8f __attribute__((address_space(4))) cfloat8;

template <int K, bool FMA, int NW, int S>
__device__ __forceinline__ void conv_tile3(const ConvParams &p, const float *__restrict__ wts, const float *__restrict__ featp,
                                           float *__restrict__ respp, float *sm, int *next_u, const ConvTile tile)
{
    constexpr bool WRAP = S == 3;
    constexpr int TW = WRAP ? 64 : kConvTW >> S, TH = WRAP ? 4 : kConvTH << S, P = 4;
    static_assert(TW * TH == 256 && TH % P == 0, "a wave of 64 lanes x 4 rows covers the tile");
    static_assert(K == 5, "the stage pipeline below is written for 5 x 5 filters");
    constexpr int PW = WRAP ? TW + 2 * (K - 1) : TW + K - 1, PH = TH + K - 1;
    constexpr int PLANE = (PH * PW) | 1;   // odd plane stride: conflict-free staging writes

    const int frame = p.frame0 + blockIdx.z;
    const LevelDesc d = p.lv[tile.level];
    const int H = d.rows, W = d.cols;
    constexpr int a = K / 2;
    const int t = threadIdx.x;
    const float *feat = featp + ((size_t)frame * p.cell_per_frame + d.cell_off) * 32;
    const int la = WRAP ? min(W - tile.x0, 64) : 0;

    {   // stage the haloed tile channel-planar (as conv_tile)
        constexpr int CPP = NW * 8, NIT = (PH * PW + CPP - 1) / CPP, UB = 6;
        const int c4 = t & 7, cell0 = t >> 3;
        const v4f border = (c4 == 7) ? v4f{0.0f, 0.0f, 0.0f, 1.0f} : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        for (int it0 = 0; it0 < NIT; it0 += UB) {
            v4f v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int ci = (it0 + u) * CPP + cell0;
                const int cy = ci / PW, cx = ci - cy * PW;
                int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
                if (WRAP) {
                    const bool second = cx >= la + K - 1;
                    gy = 4 * (tile.y0 + (second ? 1 : 0)) + cy - a;
                    gx = second ? cx - (la + K - 1) - a : tile.x0 + cx - a;
                }
                v[u] = border;
                if (it0 + u < NIT && ci < PH * PW && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v[u] = *reinterpret_cast<const v4f *>(feat + ((size_t)gy * W + gx) * 32 + c4 * 4);
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
    const int u0 = blockIdx.y * p.units_per_block;
    const int u1 = min(u0 + p.units_per_block, p.nunits);
    if (t == 0) *next_u = u0;
    __syncthreads();

    const int lane = t & 63;
    // the lane's place in the tile: LDS column / row of its window, level coordinates of its first output.  Evaluated again
    // after the channel loop (a handful of integer operations) instead of being carried through it in registers: at four
    // waves per SIMD the loop has 128 of them
    auto lane_geom = [&](int &px, int &py, int &x, int &y) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bool second = WRAP && ln >= la;
        px = WRAP ? ln + (second ? K - 1 : 0) : ln & (TW - 1);
        py = WRAP ? 0 : (ln >> (WRAP ? 0 : 5 - S)) * P;
        x = WRAP ? (second ? ln - la : tile.x0 + ln) : tile.x0 + px;
        y = WRAP ? 4 * (tile.y0 + (second ? 1 : 0)) : tile.y0 + py;
    };
    const size_t HW = (size_t)H * W;
    const float *sp0;
    {
        int px, py, x, y;
        lane_geom(px, py, x, y);
        sp0 = sm + py * PW + px;
    }
    const bool skip31 = WRAP ? (!FMA && p.c31_zero && la >= 64 && 4 * tile.y0 >= a && tile.x0 >= a && 4 * tile.y0 + TH + a <= H && tile.x0 + TW + a <= W)
                             : (!FMA && p.c31_zero && tile.y0 >= a && tile.x0 >= a && tile.y0 + TH + a <= H && tile.x0 + TW + a <= W);

    auto run_unit = [&](auto ql_tag, const int u, const int f0) {
        constexpr int QL = decltype(ql_tag)::value, QH = QL / 2;
        // the unit's weights, 8 floats per (channel, tap): wave-uniform addresses in the constant address space -> s_load
        cfloat8 *wc = (cfloat8 *)(wts + (size_t)u * (32 * K * K * 8));
        v2f r[P][QH];
#pragma unroll
        for (int pp = 0; pp < P; ++pp)
#pragma unroll
            for (int q = 0; q < QH; ++q) r[pp][q] = v2f{0.0f, 0.0f};
        float Ft[P + K - 1][K];
        v8f Wt[2][K];                  // SGPRs: tap row in use | tap row being fetched
        auto load_row = [&](int c, int rr) {
            const float *sp = sp0 + c * PLANE + rr * PW;
#pragma unroll
            for (int j = 0; j < K; ++j) Ft[rr][j] = sp[j];
        };
        auto load_w = [&](int buf, int c, int i) {
#pragma unroll
            for (int j = 0; j < K; ++j) Wt[buf][j] = wc[(c * K + i) * K + j];
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
                for (int q = 0; q < QH; ++q) w[q] = v2f{Wt[wbi][j][2 * q], Wt[wbi][j][2 * q + 1]};
                // RB rows at a time: at least four products, then their additions -- an addition issues >= 16 cycles after its
                // product (with three pairs per row a row-by-row order would put it 12 cycles after, and stall)
                constexpr int RB = QH >= 4 ? 1 : QH == 3 ? 2 : QH == 2 ? 2 : 4;
#pragma unroll
                for (int p0 = 0; p0 < P; p0 += RB) {
                    const bool first = (i == 0 && j == 0);      // see conv_tile: the `0 +` of a channel's first tap cannot reach the response
                    const v2f zero2 = v2f{0.0f, 0.0f};
                    if (FMA) {
#pragma unroll
                        for (int pr = 0; pr < RB; ++pr) {
                            const v2f f = v2f{Ft[p0 + pr + i][j], Ft[p0 + pr + i][j]};
#pragma unroll
                            for (int q = 0; q < QH; ++q) s[p0 + pr][q] = __builtin_elementwise_fma(w[q], f, first ? zero2 : s[p0 + pr][q]);
                        }
                    } else {
                        v2f tq[RB][QH];
#pragma unroll
                        for (int pr = 0; pr < RB; ++pr) {
                            const v2f f = v2f{Ft[p0 + pr + i][j], Ft[p0 + pr + i][j]};
#pragma unroll
                            for (int q = 0; q < QH; ++q) tq[pr][q] = w[q] * f;
                        }
#pragma unroll
                        for (int pr = 0; pr < RB; ++pr)
#pragma unroll
                            for (int q = 0; q < QH; ++q) s[p0 + pr][q] = first ? tq[pr][q] : s[p0 + pr][q] + tq[pr][q];
                    }
                }
            }
        };
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
        for (int rr = 0; rr < P; ++rr) load_row(0, rr);
        load_w(0, 0, 0);
        zero_s();
        // one channel: the tap rows alternate between the two weight buffers, so two channels make the pattern repeat
#define PBD_CHANNEL3(c, cn, B0, B1)                                                                                                \
    do {                                                                                                                          \
        PBD_STAGE3(load_row((c), 4); load_w(B1, (c), 1), B0, 0);                                                                  \
        PBD_STAGE3(load_row((c), 5); load_w(B0, (c), 2), B1, 1);                                                                  \
        PBD_STAGE3(load_row((c), 6); load_w(B1, (c), 3), B0, 2);                                                                  \
        PBD_STAGE3(load_row((c), 7); load_w(B0, (c), 4), B1, 3);                                                                  \
        PBD_STAGE3(load_row((cn), 0); load_row((cn), 1); load_row((cn), 2); load_row((cn), 3); load_w(B1, (cn), 0), B0, 4);       \
        add_s();                                                                                                                  \
    } while (0)
        const int cpairs = skip31 ? 30 : 32;
#pragma clang loop unroll(disable)
        for (int c = 0; c < cpairs; c += 2) {
            const int c2 = min(c + 2, 31);
            PBD_CHANNEL3(c, c + 1, 0, 1);
            PBD_CHANNEL3(c + 1, c2, 1, 0);
        }
        if (skip31) PBD_CHANNEL3(30, 31, 0, 1);
#undef PBD_CHANNEL3
#undef PBD_STAGE3
#undef PBD_PIN3
        int px, py, x, y;
        lane_geom(px, py, x, y);
        float *respg = respp + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + (size_t)y * W + x;
        if (f0 + QL <= p.nf && p.fmap == nullptr) {
            float *rg = respg + (size_t)f0 * HW;
#pragma unroll
            for (int pp = 0; pp < P; ++pp) {
                if (x < W && y + pp < H) {
#pragma unroll
                    for (int q = 0; q < QL; ++q) rg[(size_t)q * HW + (size_t)pp * W] = r[pp][q / 2][q & 1];
                }
            }
        } else if (x < W) {
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

template <int K, bool FMA, int NW>
__global__ __launch_bounds__(NW * 64, (2 * NW + 3) / 4) void k_conv3(ConvParams p, const float *__restrict__ wts, const float *__restrict__ featp,
                                                   float *__restrict__ respp)
{
    __shared__ __attribute__((aligned(16))) float sm[32 * 577 + 3];
    __shared__ int next_u;
    const int b = blockIdx.x;
    const ConvTile tile = p.shaped4[b];
    if (b < p.nshaped4[0]) conv_tile3<K, FMA, NW, 0>(p, wts, featp, respp, sm, &next_u, tile);
    else if (b < p.nshaped4[0] + p.nshaped4[1]) conv_tile3<K, FMA, NW, 1>(p, wts, featp, respp, sm, &next_u, tile);
    else if (b < p.nshaped4[0] + p.nshaped4[1] + p.nshaped4[2]) conv_tile3<K, FMA, NW, 2>(p, wts, featp, respp, sm, &next_u, tile);
    else conv_tile3<K, FMA, NW, 3>(p, wts, featp, respp, sm, &next_u, tile);
}

// one launch covers all three tile shapes: the shape is uniform per workgroup (tiles are sorted by shape)
template <int K, bool FMA, int NW>
__global__ __launch_bounds__(NW * 64, (2 * NW + 3) / 4) void k_conv(ConvParams p, const float *__restrict__ wts, const float *__restrict__ featp,
                                                  float *__restrict__ respp)
{
    __shared__ __attribute__((aligned(16))) float sm[32 * 577 + 3 + NW * 2 * ((K * K * kConvQ + 3) / 4) * 4];
    __shared__ int next_g;
    const int b = blockIdx.x;
    const ConvTile tile = p.shaped4[b];
    if (b < p.nshaped4[0]) conv_tile<K, FMA, NW, 0>(p, wts, featp, respp, sm, &next_g, tile);
    else if (b < p.nshaped4[0] + p.nshaped4[1]) conv_tile<K, FMA, NW, 1>(p, wts, featp, respp, sm, &next_g, tile);
    else if (b < p.nshaped4[0] + p.nshaped4[1] + p.nshaped4[2]) conv_tile<K, FMA, NW, 2>(p, wts, featp, respp, sm, &next_g, tile);
    else conv_tile<K, FMA, NW, 3>(p, wts, featp, respp, sm, &next_g, tile);
}

// generic kernel: any filter size, any real type R (the reference's T=double instantiation runs here);
// one output pixel per thread, weights [channel][tap][Fpad] read with wave-uniform vector loads
template <typename R, bool FMA>
__global__ __launch_bounds__(256) void k_conv_generic(ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    R *smd = reinterpret_cast<R *>(smraw);
    constexpr int TW = kConvTW, TH = kConvTH, Q = kConvQ;
    const int K = p.ksize;
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
    {
        const int c = t & 31;
        const R border = (c == 31) ? (R)1 : (R)0;
        for (int ci = t >> 5; ci < PH * PW; ci += 8) {
            const int cy = ci / PW, cx = ci - cy * PW;
            const int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
            R v = border;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = feat[((size_t)gy * W + gx) * 32 + c];
            smd[c * PLANE + ci] = v;
        }
    }
    __syncthreads();
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
            const R *sp = smd + c * PLANE + py * PW + px;
            const R *wp = wts + (size_t)c * (K * K) * p.Fpad + g * Q;
            R s[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) s[q] = (R)0;
            for (int i = 0; i < K; ++i) {
                for (int j = 0; j < K; ++j) {
                    const R f = sp[i * PW + j];
                    const R *w = wp + (size_t)(i * K + j) * p.Fpad;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if (FMA) s[q] = __builtin_fma(w[q], f, s[q]);
                        else s[q] = s[q] + w[q] * f;
                    }
                }
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
    if (nw == 4) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv<5, false, 4>, 256, 0);
    return n;
}

template <typename R, bool FMA>
static void launch_generic(const ConvParams &p, dim3 grid, hipStream_t s)
{
    const int PW = kConvTW + p.ksize - 1, PH = kConvTH + p.ksize - 1;
    const size_t lds = (size_t)32 * ((PH * PW) | 1) * sizeof(R);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_generic<R, FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PBD_LAUNCH((k_conv_generic<R, FMA>), grid, dim3(256), lds, s, p);
}

template <bool FMA, int NW>
static void launch_shapes3(const ConvParams &p, int nframes, hipStream_t s)
{
    const int nt = p.nshaped4[0] + p.nshaped4[1] + p.nshaped4[2] + p.nshaped4[3];
    const int gy = (p.nunits + p.units_per_block - 1) / p.units_per_block;
    PBD_LAUNCH((k_conv3<5, FMA, NW>), dim3(nt, gy, nframes), dim3(NW * 64), 0, s, p, static_cast<const float *>(p.wts3),
               static_cast<const float *>(p.feat), static_cast<float *>(p.resp));
}

template <bool FMA, int NW>
static void launch_shapes(const ConvParams &p, int gy, int nframes, hipStream_t s)
{
    const int nt = p.nshaped4[0] + p.nshaped4[1] + p.nshaped4[2] + p.nshaped4[3];
    PBD_LAUNCH((k_conv<5, FMA, NW>), dim3(nt, gy, nframes), dim3(NW * 64), 0, s, p, static_cast<const float *>(p.wts),
                       static_cast<const float *>(p.feat), static_cast<float *>(p.resp));
}

void launch_conv(const ConvParams &p, int nframes, bool f64, hipStream_t s)
{
    if (p.ntiles == 0 || p.nf == 0) return;
    const int ngroups = p.Fpad / kConvQ;
    const int gy = (ngroups + p.groups_per_block - 1) / p.groups_per_block;
    dim3 grid(p.ntiles, gy, nframes);
    if (!f64 && p.ksize == 5 && p.wts3 != nullptr) {
        // round 3: weights through the scalar unit, six waves per workgroup (three per SIMD)
        if (p.fma) launch_shapes3<true, kConv3NW>(p, nframes, s); else launch_shapes3<false, kConv3NW>(p, nframes, s);
    } else if (!f64 && p.ksize == 5) {
        // 4 waves per workgroup, so that the two resident workgroups put 2 waves on every SIMD (5 or 6 leave the
        // SIMDs unevenly loaded: 54.7 / 50.9 ms vs 46.0 ms per 64-frame step when this was measured)
        if (p.fma) launch_shapes<true, 4>(p, gy, nframes, s); else launch_shapes<false, 4>(p, gy, nframes, s);
    } else if (f64) {
        if (p.fma) launch_generic<double, true>(p, grid, s); else launch_generic<double, false>(p, grid, s);
    } else {
        if (p.fma) launch_generic<float, true>(p, grid, s); else launch_generic<float, false>(p, grid, s);
    }
}

}  // namespace pbd
