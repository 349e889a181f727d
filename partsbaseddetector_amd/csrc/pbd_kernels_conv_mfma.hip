// pbd_kernels_conv_mfma.hip -- the filter bank as a dense contraction on the matrix cores (gfx950).
//
// Opt-in mode PBD_CONV_MFMA (BASELINE.json north_star: "MFMA only if the 32-channel filter dot-products
// are cast as a dense contraction").  Responses agree with the reference to ~1e-6 (bar: 1e-4), but are
// not bit-identical, so the default stays the EXACT VALU kernel (pbd_kernels_conv.hip).
//
//   out[filter][pixel] = sum_{tap, channel} W[filter][tap][channel] * F[pixel + tap][channel]      (K = 800)
//
// Every fp32 operand is split into two bf16 terms, x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), and
// hi*hi + hi*lo + lo*hi is accumulated in fp32 by v_mfma_f32_32x32x16_bf16: 16 significant bits per
// operand, relative product error ~2^-16, three MFMAs per fp32 MAC tile instead of one.
//
// Mapping: see conv_mfma_tile below.
//
// PBD_CONV_MFMA_F16 (BASELINE.json configs[4], SURVEY.md section 8(d) "Config 5"): the same contraction with
// every operand rounded once to fp16 and ONE v_mfma_f32_32x32x16_f16 per product tile, fp32 accumulation.
// Records are 80 bytes (32 fp16 + 16 pad, again conflict-free for the 16-byte fragment reads), a third of
// the matrix-core work and 35 KB of LDS per workgroup.
// The 1e-4 score bar does NOT hold in this mode; tests report the error and the detection agreement.
#include "pbd_internal.h"

#include <type_traits>

namespace pbd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaFB = 160;           // filters per pass (5 M-tiles of 32)

// Workgroup = one 256-pixel tile (32 x 8, 16 x 16 or 8 x 32: the mixed cover of the exact kernel) of one level / frame
// x 160 filters, FIVE waves: wave m owns the 32 filters of M-tile m and all 8 N-tiles (32 pixels each) of the tile.
//   A operand (weights of one tap, 16 channels): straight from global memory / L2 in fragment order
//     ([pass][tap][k-step][M-tile][hi|lo][lane] x 16 bytes, built once on the host) -- one coalesced 1 KB load per
//     fragment, fetched one step ahead; every wave of every workgroup reads the same 800 KB, so they stay in L2 / L1.
//   B operand (features): the haloed tile sits in LDS as [cell][hi 32ch | lo 32ch | pad] records (144 B, or 80 B for
//     fp16: the 16-byte fragment reads of 32 consecutive cells are bank-conflict free), converted once per workgroup.
//   No barrier after the tile is staged: the waves do not share anything else, so two workgroups per CU (62 KB of LDS
//   each) keep the matrix pipes busy while others stage or store.
//   D[filter][pixel]: lanes = pixels, registers = filters -> a store instruction writes runs of consecutive x.
template <bool F16> struct MfmaRec { static constexpr int kBytes = F16 ? kMfmaRecBytesF16 : kMfmaRecBytes; };

template <int K, bool F16, int S>
__device__ __forceinline__ void conv_mfma_tile(const ConvParams &p, const u32x4 *__restrict__ wfrag, const float *__restrict__ featp,
                                               float *__restrict__ respp, unsigned char *sm_f, const ConvTile tile)
{
    constexpr int kRec = MfmaRec<F16>::kBytes;
    constexpr int TW = kConvTW >> S, TH = kConvTH << S;
    constexpr int PW = TW + K - 1, PH = TH + K - 1;
    constexpr int NCELL = PW * PH;
    constexpr int MT = kMfmaFB / 32, NT = 8, NV = F16 ? 1 : 2;
    static_assert(MT == 5 && NT == 8, "the static split over four waves below is written for 5 x 8 tile products");
    const int frame = p.frame0 + blockIdx.z;
    const int pass = blockIdx.y;                       // block of 160 filters
    const LevelDesc d = p.lv[tile.level];
    const int H = d.rows, W = d.cols;
    constexpr int a = K / 2;
    const int t = threadIdx.x;
    const float *feat = featp + ((size_t)frame * p.cell_per_frame + d.cell_off) * 32;

    // ---- stage the feature tile: fp32 -> (hi, lo) bf16 or fp16; thread = (cell, 8-channel group); the loads of a
    // batch of UB tasks are issued together (one exposed memory latency per batch instead of one per task)
    {
        constexpr int NTASK = NCELL * 4, NTHR = 256, UB = 4;
        for (int base = 0; base < NTASK; base += NTHR * UB) {
            float4 va[UB], vb[UB];
            bool inside[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * NTHR + t;
                const int ci = idx >> 2, cg = idx & 3;
                const int cy = ci / PW, cx = ci - cy * PW;
                const int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
                inside[u] = idx < NTASK && gy >= 0 && gy < H && gx >= 0 && gx < W;
                va[u] = vb[u] = float4{0.f, 0.f, 0.f, 0.f};
                if (inside[u]) {
                    const float4 *src = reinterpret_cast<const float4 *>(feat + ((size_t)gy * W + gx) * 32 + cg * 8);
                    va[u] = src[0]; vb[u] = src[1];
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * NTHR + t;
                if (idx >= NTASK) continue;
                const int ci = idx >> 2, cg = idx & 3;
                float v[8] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
                if (!inside[u] && cg == 3) v[7] = 1.0f;        // constant border: 1 on channel 31 (SpatialConvolutionEngine.cpp:153-156)
                if constexpr (F16) {
                    f16x8 hv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) hv[j] = (_Float16)v[j];
                    *reinterpret_cast<f16x8 *>(sm_f + ci * kRec + cg * 16) = hv;
                } else {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        hi[j] = (__bf16)v[j];
                        lo[j] = (__bf16)(v[j] - (float)hi[j]);
                    }
                    *reinterpret_cast<bf16x8 *>(sm_f + ci * kRec + cg * 16) = hi;
                    *reinterpret_cast<bf16x8 *>(sm_f + ci * kRec + 64 + cg * 16) = lo;
                }
            }
        }
    }
    __syncthreads();

    const int lane = t & 63;
    const int r = lane & 31, hh = lane >> 5;
    // pixel q = n * 32 + r of the tile -> (py, px) = (q / TW, q % TW); byte offset of its cell record in the haloed LDS
    // tile: N-tile n starts (32 / TW) rows below N-tile n - 1, so the offsets are cell0 + n * kNStep
    const int cell0 = ((r / TW) * PW + (r % TW)) * kRec + hh * 16;
    constexpr int kNStep = (32 / TW) * PW * kRec;
    constexpr size_t kStep = (size_t)MT * NV * 64;
    constexpr int NSTEP = K * K * 2;
    const size_t HW = (size_t)H * W;
    using RT = typename std::conditional<F16, _Float16, float>::type;

    // One block of MB M-tiles (32 filters each) x NB N-tiles (32 pixels each) by one wave.  Per step (one tap, 16 channels):
    // MB x NV weight fragments from L2 (fragment order, PF steps ahead, a ring of registers with static indices), NB x NV
    // feature fragments from LDS, MB x NB x (F16 ? 1 : 3) MFMAs.  A feature fragment is single-buffered: the fragment of the
    // NEXT step is requested right after the MFMAs that read the current one have been issued, and is needed a whole step later.
    auto block = [&](auto mb_c, auto nb_c, auto pf_c, int m0, int n0) {
        constexpr int MB = decltype(mb_c)::value, NB = decltype(nb_c)::value, PF = decltype(pf_c)::value;
        static_assert(NSTEP % PF == 0, "the step loop runs in whole groups of PF");
        const u32x4 *wsrc = wfrag + ((size_t)pass * (K * K * 2) * MT + m0) * NV * 64 + lane;
        const int celln = cell0 + n0 * kNStep;
        f32x16 acc[MB][NB];
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][n][e] = 0.0f;
        u32x4 wq[PF][MB][NV];
#pragma unroll
        for (int j = 0; j < PF; ++j)
#pragma unroll
            for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                for (int v = 0; v < NV; ++v) wq[j][mi][v] = wsrc[(size_t)j * kStep + (mi * NV + v) * 64];
        auto b_addr = [&](int sidx) {
            const int tp = sidx >> 1, kh = sidx & 1;
            const int ti = tp / K, tj = tp - ti * K;
            return sm_f + (ti * PW + tj) * kRec + kh * 32 + celln;
        };
        u32x4 bq[NB][NV];
        {
            const unsigned char *bb = b_addr(0);
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int v = 0; v < NV; ++v) bq[n][v] = *reinterpret_cast<const u32x4 *>(bb + n * kNStep + v * 64);
        }
#pragma unroll 1
        for (int s0 = 0; s0 < NSTEP; s0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int sidx = s0 + j;
                const unsigned char *bnext = b_addr(min(sidx + 1, NSTEP - 1));
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    if constexpr (F16) {
                        const f16x8 bf = __builtin_bit_cast(f16x8, bq[n][0]);
#pragma unroll
                        for (int mi = 0; mi < MB; ++mi)
                            acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wq[j][mi][0]), bf, acc[mi][n], 0, 0, 0);
                    } else {
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[n][0]), bl = __builtin_bit_cast(bf16x8, bq[n][1]);
#pragma unroll
                        for (int mi = 0; mi < MB; ++mi)
                            acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[j][mi][0]), bh, acc[mi][n], 0, 0, 0);
#pragma unroll
                        for (int mi = 0; mi < MB; ++mi)
                            acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[j][mi][0]), bl, acc[mi][n], 0, 0, 0);
#pragma unroll
                        for (int mi = 0; mi < MB; ++mi)
                            acc[mi][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wq[j][mi][1]), bh, acc[mi][n], 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < NV; ++v) bq[n][v] = *reinterpret_cast<const u32x4 *>(bnext + n * kNStep + v * 64);
                }
                const int snext = min(sidx + PF, NSTEP - 1);
#pragma unroll
                for (int mi = 0; mi < MB; ++mi)
#pragma unroll
                    for (int v = 0; v < NV; ++v) wq[j][mi][v] = wsrc[(size_t)snext * kStep + (mi * NV + v) * 64];
            }
        }
        // D layout (32x32): column = lane & 31 (pixel), row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5) (filter).
        // The 16 plane pointers of a lane are formed by additions from one base (the products f * HW were two full-width
        // multiplies per store, and `f < F` a divergent region per store: more instructions than the fp16 mode's matrix work);
        // the filter test is uniform for every M-tile but the last (m_full).
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) {
            const int f0 = pass * kMfmaFB + (m0 + mi) * 32 + 4 * hh;
            const bool m_full = pass * kMfmaFB + (m0 + mi) * 32 + 32 <= p.F;
            RT *rbase = reinterpret_cast<RT *>(respp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + (size_t)f0 * HW;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int q = (n0 + n) * 32 + r;
                const int y = tile.y0 + q / TW, x = tile.x0 + q % TW;
                if (x < W && y < H) {
                    RT *rp = rbase + (size_t)y * W + x;
                    if (m_full) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            RT *rg = rp + (size_t)(8 * g4) * HW;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { *rg = (RT)acc[mi][n][4 * g4 + e]; rg += HW; }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int fo = (e & 3) + 8 * (e >> 2);
                            if (f0 + fo < p.F) rp[(size_t)fo * HW] = (RT)acc[mi][n][e];
                        }
                    }
                }
            }
        }
    };
    // Static, balanced split of the 5 M-tiles x 8 N-tiles of the workgroup over its four waves, ten tile products each:
    // wave (h, u) owns N-tiles 4h .. 4h+3; M-tiles 2u, 2u+1 against all four of them (a 2 x 4 block: every feature fragment
    // read from LDS feeds six MFMAs, every weight fragment twelve), then M-tile 4 against N-tiles 4h+2u, 4h+2u+1 (1 x 2).
    // Round 2 handed out ten 1 x 4 items through an LDS counter (4 + 4 + 2 items over four waves, twice the LDS reads):
    // bf16 7.6 -> 6.75 ms, fp16 2.95 -> 2.83 ms per 64-frame step.  All ten products in ONE step loop (254 registers) was
    // no faster in bf16 and slower in fp16 (3.4 ms); see profiles/r03_conv/README.md for where the rest of the time is.
    auto mtile_used = [&](int m) { return pass * kMfmaFB + m * 32 < p.F; };
    const int wv = t >> 6, h = wv >> 1, u = wv & 1;
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    if (mtile_used(2 * u + 1)) block(I2{}, I4{}, I2{}, 2 * u, 4 * h);
    else if (mtile_used(2 * u)) block(I1{}, I4{}, I5{}, 2 * u, 4 * h);
    if (mtile_used(4)) block(I1{}, I2{}, I5{}, 4, 4 * h + 2 * u);
}

template <int K, bool F16>
// two workgroups of four waves per CU (LDS: 62 KB each in bf16 mode): two waves per SIMD is what the kernel gets, so that is
// what it asks for -- a request of three made the compiler squeeze to 168 registers it could not reach (220) for nothing
__global__ __launch_bounds__(256, 2) void k_conv_mfma(ConvParams p, const u32x4 *__restrict__ wfrag,
                                                                      const float *__restrict__ featp, float *__restrict__ respp)
{
    __shared__ __attribute__((aligned(16))) unsigned char sm_f[433 * MfmaRec<F16>::kBytes];
    const int b = blockIdx.x;
    const ConvTile tile = p.shaped[b];
    if (b < p.nshaped[0]) conv_mfma_tile<K, F16, 0>(p, wfrag, featp, respp, sm_f, tile);
    else if (b < p.nshaped[0] + p.nshaped[1]) conv_mfma_tile<K, F16, 1>(p, wfrag, featp, respp, sm_f, tile);
    else conv_mfma_tile<K, F16, 2>(p, wfrag, featp, respp, sm_f, tile);
}

int conv_mfma_occupancy(bool f16)
{   // resident workgroups per CU (diagnostics)
    int n = -1;
    if (f16) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_mfma<5, true>, 256, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_conv_mfma<5, false>, 256, 0);
    return n;
}

void launch_conv_mfma(const ConvParams &p, const void *wrec, bool f16, int nframes, hipStream_t s)
{
    const int nt = p.nshaped[0] + p.nshaped[1] + p.nshaped[2];
    if (nt == 0 || p.F == 0) return;
    const int passes = (p.F + kMfmaFB - 1) / kMfmaFB;
    dim3 grid(nt, passes, nframes);
    if (f16)
        PBD_LAUNCH((k_conv_mfma<5, true>), grid, dim3(256), 0, s, p, static_cast<const u32x4 *>(wrec),
                           static_cast<const float *>(p.feat), static_cast<float *>(p.resp));
    else
        PBD_LAUNCH((k_conv_mfma<5, false>), grid, dim3(256), 0, s, p, static_cast<const u32x4 *>(wrec),
                           static_cast<const float *>(p.feat), static_cast<float *>(p.resp));
}

}  // namespace pbd
