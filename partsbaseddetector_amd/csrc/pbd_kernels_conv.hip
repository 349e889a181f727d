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
// Mapping: one workgroup = one 32 x 8 tile of one level of one frame.  The (32+k-1) x (8+k-1) x 32
// channel input tile is staged once in LDS, re-laid out channel-planar so that a wave's lanes
// (consecutive x) read consecutive LDS words; out-of-image cells are materialised with the
// reference's constant border (0, but 1 for the last channel: :147-156).  Each thread owns one
// output pixel and sweeps the filters in groups of 8; the 8 weights of a (channel, tap) are
// wave-uniform and come through the scalar cache, so the inner loop is 16 VALU ops per LDS read.
// Compiled with -ffp-contract=off.
#include "pbd_internal.h"

namespace pbd {

// Weights are read-only for the whole launch and every address is wave-uniform: reading them
// through the constant address space lets the compiler keep them in SGPRs (s_load_dwordx8).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(4))) cfloat4;

template <int K, bool FMA>
__global__ __launch_bounds__(256, 2) void k_conv(ConvParams p, const float *__restrict__ wts, const float *__restrict__ featp,
                                                 float *__restrict__ respp)
{
    constexpr int TW = kConvTW, TH = kConvTH, Q = kConvQ;
    constexpr int PW = TW + K - 1, PH = TH + K - 1;
    constexpr int PLANE = (PH * PW) | 1;   // odd plane stride: conflict-free staging writes
    __shared__ float sm[32 * PLANE];

    const ConvTile tile = p.tiles[blockIdx.x];
    const int frame = blockIdx.z;
    const LevelDesc d = p.lv[tile.level];
    const int H = d.rows, W = d.cols;
    constexpr int a = K / 2;
    const int t = threadIdx.x;
    const float *feat = featp + ((size_t)frame * p.cell_per_frame + d.cell_off) * 32;

    {   // stage: lane = channel, 8 cells per pass; global reads are 128-byte cells
        const int c = t & 31;
        const float border = (c == 31) ? 1.0f : 0.0f;
        for (int ci = t >> 5; ci < PH * PW; ci += 8) {
            const int cy = ci / PW, cx = ci - cy * PW;
            const int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
            float v = border;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = feat[((size_t)gy * W + gx) * 32 + c];
            sm[c * PLANE + ci] = v;
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
    float *resp = respp + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + (size_t)y * W + x;
    const float *sp0 = sm + py * PW + px;

    for (int g = g0; g < g1; ++g) {
        float r[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) r[q] = 0.0f;
        for (int c = 0; c < 32; ++c) {
            const float *sp = sp0 + c * PLANE;
            cfloat4 *wp = (cfloat4 *)(wts + (size_t)c * (K * K) * p.Fpad + g * Q);
            const int wstride = p.Fpad / 4;
            float s[Q];
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float f = sp[i * PW + j];
                    const v4f wa = wp[(size_t)(i * K + j) * wstride];
                    const v4f wb = wp[(size_t)(i * K + j) * wstride + 1];
                    const float w[Q] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if (i == 0 && j == 0) s[q] = w[q] * f;
                        else if (FMA) s[q] = __fmaf_rn(w[q], f, s[q]);
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
                if (f < p.F) resp[(size_t)f * HW] = r[q];
            }
        }
    }
}

// generic filter size (no unrolling); used for models whose filters are not 5x5
template <bool FMA>
__global__ __launch_bounds__(256) void k_conv_generic(ConvParams p)
{
    extern __shared__ float smd[];
    constexpr int TW = kConvTW, TH = kConvTH, Q = kConvQ;
    const int K = p.ksize;
    const int PW = TW + K - 1, PH = TH + K - 1;
    const int PLANE = (PH * PW) | 1;
    const ConvTile tile = p.tiles[blockIdx.x];
    const int frame = blockIdx.z;
    const LevelDesc d = p.lv[tile.level];
    const int H = d.rows, W = d.cols;
    const int a = K / 2;
    const int t = threadIdx.x;
    const float *feat = p.feat + ((size_t)frame * p.cell_per_frame + d.cell_off) * 32;
    {
        const int c = t & 31;
        const float border = (c == 31) ? 1.0f : 0.0f;
        for (int ci = t >> 5; ci < PH * PW; ci += 8) {
            const int cy = ci / PW, cx = ci - cy * PW;
            const int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
            float v = border;
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
    float *resp = p.resp + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + (size_t)y * W + x;
    for (int g = g0; g < g1; ++g) {
        float r[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) r[q] = 0.0f;
        for (int c = 0; c < 32; ++c) {
            const float *sp = smd + c * PLANE + py * PW + px;
            const float *wp = p.wts + (size_t)c * (K * K) * p.Fpad + g * Q;
            float s[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) s[q] = 0.0f;
            for (int i = 0; i < K; ++i) {
                for (int j = 0; j < K; ++j) {
                    const float f = sp[i * PW + j];
                    const float *w = wp + (size_t)(i * K + j) * p.Fpad;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if (FMA) s[q] = __fmaf_rn(w[q], f, s[q]);
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
                if (f < p.F) resp[(size_t)f * HW] = r[q];
            }
        }
    }
}

void launch_conv(const ConvParams &p, int nframes, hipStream_t s)
{
    if (p.ntiles == 0 || p.F == 0) return;
    const int ngroups = p.Fpad / kConvQ;
    const int gy = (ngroups + p.groups_per_block - 1) / p.groups_per_block;
    dim3 grid(p.ntiles, gy, nframes);
    if (p.ksize == 5) {
        if (p.fma) hipLaunchKernelGGL((k_conv<5, true>), grid, dim3(256), 0, s, p, p.wts, p.feat, p.resp);
        else hipLaunchKernelGGL((k_conv<5, false>), grid, dim3(256), 0, s, p, p.wts, p.feat, p.resp);
    } else {
        const int PW = kConvTW + p.ksize - 1, PH = kConvTH + p.ksize - 1;
        const size_t lds = (size_t)32 * ((PH * PW) | 1) * sizeof(float);
        if (p.fma) hipLaunchKernelGGL((k_conv_generic<true>), grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL((k_conv_generic<false>), grid, dim3(256), lds, s, p);
    }
}

}  // namespace pbd
