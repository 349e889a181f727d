// pbd_kernels_features.hip -- pyramid resampling and HOG feature kernels (gfx950).
//
// Replaces HOGFeatures<T>::pyramid / features<uint8_t> (reference src/HOGFeatures.cpp:95-341) for
// T=float.  All arithmetic is ordered exactly as the reference's; the file is compiled with
// -ffp-contract=off so no multiply-add is fused.
#include "pbd_internal.h"

#include <stdlib.h>

namespace pbd {

// level containing flat element `idx` for the offsets selected by OFF (0 img, 1 blk, 2 cell)
template <int OFF>
__device__ __forceinline__ long long lv_off(const LevelDesc &d)
{
    return OFF == 0 ? d.img_off : (OFF == 1 ? d.blk_off : d.cell_off);
}
template <int OFF>
__device__ __forceinline__ int find_level(const LevelDesc *lv, int lo, int hi, long long idx)
{   // largest l in [lo, hi) with off(l) <= idx  (offsets are non-decreasing)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (lv_off<OFF>(lv[mid]) <= idx) lo = mid; else hi = mid;
    }
    return lo;
}

// The three bytes of a BGR pixel in one (unaligned) 32-bit load: byte loads cost a full memory instruction
// each, and these kernels are bound by the number of them.  The fourth byte belongs to the next pixel (the
// pyramid buffer carries 4 bytes of slack; callers' frames use the byte path for their very last pixel).
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t load_px3(const uint8_t *q) { return *reinterpret_cast<const u32_unaligned *>(q); }
__device__ __forceinline__ uint32_t load_px3_bytes(const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16); }
__device__ __forceinline__ int px_ch(uint32_t v, int c) { return (int)((v >> (8 * c)) & 0xffu); }

// Block-cooperative variant: the offsets of the candidate levels go to LDS in ONE memory round trip and the
// search runs there.  (The per-thread search above is six DEPENDENT global loads -- ~4000 cycles before a wave's
// first useful instruction -- and was what bounded these short kernels.)  Every thread of the block must call it.
template <int OFF>
__device__ __forceinline__ int find_level_blk(const LevelDesc *lv, int lo, int hi, long long idx, long long *s_off)
{
    for (int i = lo + (int)threadIdx.x; i < hi; i += (int)blockDim.x) s_off[i] = lv_off<OFF>(lv[i]);
    __syncthreads();
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (s_off[mid] <= idx) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// cv::resize, INTER_LINEAR, 8-bit (call site src/HOGFeatures.cpp:116).  Coefficient tables are
// built on the host (pbd_plan.cpp); here: horizontal pass in int, vertical pass
// ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2 >> 2.  One thread per destination pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resize(PyrParams p, long long npix)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = find_level_blk<0>(p.lv, 0, p.interval, idx, s_off);
    if (idx >= npix) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.img_off);
    const int dy = local / d.img_cols, dx = local - dy * d.img_cols;
    const ResizeTabX tx = p.tabx[d.tab_x + dx];
    const ResizeTabY ty = p.taby[d.tab_y + dy];
    const int cn = p.cn;
    const uint8_t *src = p.frames + (size_t)frame * p.rows * p.cols * cn;
    const uint8_t *S0 = src + (size_t)ty.y0 * p.cols * cn, *S1 = src + (size_t)ty.y1 * p.cols * cn;
    const int sx = tx.sx, sx1 = sx + 1 < p.cols ? sx + 1 : sx;
    uint8_t *D = p.pyr + ((size_t)frame * p.pix_per_frame + d.img_off + local) * cn;
    if (cn == 3) {
        // the last pixel of the caller's frame is read bytewise (no 4th byte to touch)
        auto ld = [&](const uint8_t *row, int yy, int xx) {
            return (yy == p.rows - 1 && xx == p.cols - 1) ? load_px3_bytes(row + xx * 3) : load_px3(row + xx * 3);
        };
        const uint32_t p00 = ld(S0, ty.y0, sx), p10 = ld(S1, ty.y1, sx), p01 = ld(S0, ty.y0, sx1), p11 = ld(S1, ty.y1, sx1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int r0 = px_ch(p00, c) * tx.a0 + px_ch(p01, c) * tx.a1;
            const int r1 = px_ch(p10, c) * tx.a0 + px_ch(p11, c) * tx.a1;
            D[c] = (uint8_t)((((ty.b0 * (r0 >> 4)) >> 16) + ((ty.b1 * (r1 >> 4)) >> 16) + 2) >> 2);
        }
        return;
    }
    for (int c = 0; c < cn; ++c) {
        const int r0 = S0[sx * cn + c] * tx.a0 + S0[sx1 * cn + c] * tx.a1;
        const int r1 = S1[sx * cn + c] * tx.a0 + S1[sx1 * cn + c] * tx.a1;
        D[c] = (uint8_t)((((ty.b0 * (r0 >> 4)) >> 16) + ((ty.b1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
}

// Four consecutive destination pixels per thread (8-bit BGR, the hot case): the 12 result bytes leave as three 4-byte
// stores instead of twelve 1-byte ones and the row coefficients are fetched once.  Quads that run over the end of a
// level row fall back to pixel-by-pixel byte stores.  Same arithmetic per pixel as k_resize.
__device__ __forceinline__ uint32_t resize_px3(const PyrParams &p, const uint8_t *S0, const uint8_t *S1, const ResizeTabY &ty, const ResizeTabX &tx)
{
    auto ld = [&](const uint8_t *row, int yy, int xx) {     // the last pixel of the caller's frame is read bytewise
        return (yy == p.rows - 1 && xx == p.cols - 1) ? load_px3_bytes(row + xx * 3) : load_px3(row + xx * 3);
    };
    const int sx = tx.sx, sx1 = sx + 1 < p.cols ? sx + 1 : sx;
    const uint32_t p00 = ld(S0, ty.y0, sx), p10 = ld(S1, ty.y1, sx), p01 = ld(S0, ty.y0, sx1), p11 = ld(S1, ty.y1, sx1);
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int r0 = px_ch(p00, c) * tx.a0 + px_ch(p01, c) * tx.a1;
        const int r1 = px_ch(p10, c) * tx.a0 + px_ch(p11, c) * tx.a1;
        out |= (uint32_t)(uint8_t)((((ty.b0 * (r0 >> 4)) >> 16) + ((ty.b1 * (r1 >> 4)) >> 16) + 2) >> 2) << (8 * c);
    }
    return out;
}

__global__ __launch_bounds__(256) void k_resize4(PyrParams p, long long npix)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int l = find_level_blk<0>(p.lv, 0, p.interval, min(idx, npix - 1), s_off);
    if (idx >= npix) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.img_off);
    const int dy = local / d.img_cols, dx = local - dy * d.img_cols;
    const uint8_t *src = p.frames + (size_t)frame * p.rows * p.cols * 3;
    uint8_t *D = p.pyr + ((size_t)frame * p.pix_per_frame + idx) * 3;
    if (dx + 3 < d.img_cols) {
        const ResizeTabY ty = p.taby[d.tab_y + dy];
        const uint8_t *S0 = src + (size_t)ty.y0 * p.cols * 3, *S1 = src + (size_t)ty.y1 * p.cols * 3;
        uint32_t q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = resize_px3(p, S0, S1, ty, p.tabx[d.tab_x + dx + i]);
        u32_unaligned *o = reinterpret_cast<u32_unaligned *>(D);
        o[0] = q[0] | (q[1] << 24);
        o[1] = (q[1] >> 8) | (q[2] << 16);
        o[2] = (q[2] >> 16) | (q[3] << 8);
        return;
    }
    for (int i = 0; i < 4; ++i) {     // the quad wraps to the next row or runs into the next level
        const long long pi = idx + i;
        if (pi >= npix) return;
        int li = l;
        while (li + 1 < p.interval && p.lv[li + 1].img_off <= pi) ++li;
        const LevelDesc di = p.lv[li];
        const int loc = (int)(pi - di.img_off);
        const int y = loc / di.img_cols, x = loc - y * di.img_cols;
        const ResizeTabY ty = p.taby[di.tab_y + y];
        const uint32_t v = resize_px3(p, src + (size_t)ty.y0 * p.cols * 3, src + (size_t)ty.y1 * p.cols * 3, ty, p.tabx[di.tab_x + x]);
        D[3 * i] = (uint8_t)v; D[3 * i + 1] = (uint8_t)(v >> 8); D[3 * i + 2] = (uint8_t)(v >> 16);
    }
}

// The other depths (16U, 32F, 64F): cv::resize keeps float coefficients and works in float (double for 64F);
// D = S[sx]*a0 + S[sx+1]*a1 (exactly S[sx] at the last column), dst = cast(R0*b0 + R1*b1), cast = cvRound + clamp for
// 16U (third-party arithmetic restated from OpenCV's generic code path; unpinned, as for 8-bit).
template <typename PT, typename WT> __device__ __forceinline__ PT resize_cast(WT v);
template <> __device__ __forceinline__ uint16_t resize_cast<uint16_t, float>(float v)
{
    const int iv = __float2int_rn(v);
    return (uint16_t)(iv < 0 ? 0 : iv > 65535 ? 65535 : iv);
}
template <> __device__ __forceinline__ float resize_cast<float, float>(float v) { return v; }
template <> __device__ __forceinline__ double resize_cast<double, double>(double v) { return v; }

template <typename PT, typename WT>
__global__ __launch_bounds__(256) void k_resize_t(PyrParams p, long long npix)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = find_level_blk<0>(p.lv, 0, p.interval, idx, s_off);
    if (idx >= npix) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.img_off);
    const int dy = local / d.img_cols, dx = local - dy * d.img_cols;
    const ResizeTabXf tx = p.tabxf[d.tab_x + dx];
    const ResizeTabYf ty = p.tabyf[d.tab_y + dy];
    const int cn = p.cn;
    const PT *src = reinterpret_cast<const PT *>(p.frames) + (size_t)frame * p.rows * p.cols * cn;
    const PT *S0 = src + (size_t)ty.y0 * p.cols * cn, *S1 = src + (size_t)ty.y1 * p.cols * cn;
    PT *D = reinterpret_cast<PT *>(p.pyr) + ((size_t)frame * p.pix_per_frame + d.img_off + local) * cn;
    for (int c = 0; c < cn; ++c) {
        WT r0, r1;
        if (tx.last) {
            r0 = (WT)S0[tx.sx * cn + c] * (WT)1; r1 = (WT)S1[tx.sx * cn + c] * (WT)1;
        } else {
            r0 = (WT)S0[tx.sx * cn + c] * (WT)tx.a0 + (WT)S0[(tx.sx + 1) * cn + c] * (WT)tx.a1;
            r1 = (WT)S1[tx.sx * cn + c] * (WT)tx.a0 + (WT)S1[(tx.sx + 1) * cn + c] * (WT)tx.a1;
        }
        D[c] = resize_cast<PT, WT>(r0 * (WT)ty.b0 + r1 * (WT)ty.b1);
    }
}

void launch_resize(const PyrParams &p, int nframes, long long npix, hipStream_t s)
{
    dim3 grid((unsigned)((npix + 255) / 256), nframes);
    if (p.depth == kDepth16U) PBD_LAUNCH((k_resize_t<uint16_t, float>), grid, dim3(256), 0, s, p, npix);
    else if (p.depth == kDepth32F) PBD_LAUNCH((k_resize_t<float, float>), grid, dim3(256), 0, s, p, npix);
    else if (p.depth == kDepth64F) PBD_LAUNCH((k_resize_t<double, double>), grid, dim3(256), 0, s, p, npix);
    else if (p.cn == 3) {
        dim3 grid4((unsigned)((npix + 1023) / 1024), nframes);
        PBD_LAUNCH(k_resize4, grid4, dim3(256), 0, s, p, npix);
    } else PBD_LAUNCH(k_resize, grid, dim3(256), 0, s, p, npix);
}

// ------------------------------------------------------------------------------------------------
// cv::pyrDown, 8-bit (call site src/HOGFeatures.cpp:122): [1 4 6 4 1] x [1 4 6 4 1],
// BORDER_REFLECT_101, (sum + 128) >> 8.  One thread per destination pixel of the levels
// [first_level, last_level), whose sources are the levels `interval` below.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len)
{
    // the taps reach at most two positions outside [0, len): for len >= 3 one reflection each way is the whole loop, as
    // selects (the loop form cost a compare-and-branch pair per coordinate, ten per pixel)
    if (len >= 3) {
        p = p < 0 ? -p : p;
        return p >= len ? 2 * len - 2 - p : p;
    }
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void k_pyrdown(PyrParams p, int first_level, int last_level, long long base, long long npix)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = find_level_blk<0>(p.lv, first_level, last_level, idx + base, s_off);
    if (idx >= npix) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const LevelDesc sd = p.lv[d.src_level];
    const int local = (int)(idx + base - d.img_off);
    const int y = local / d.img_cols, x = local - y * d.img_cols;
    const int cn = p.cn;
    const uint8_t *S = p.pyr + ((size_t)frame * p.pix_per_frame + sd.img_off) * cn;
    uint8_t *D = p.pyr + ((size_t)frame * p.pix_per_frame + d.img_off + local) * cn;
    int xs[5], ys[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        xs[k] = reflect101(2 * x - 2 + k, sd.img_cols) * cn;
        ys[k] = reflect101(2 * y - 2 + k, sd.img_rows);
    }
    if (cn == 3) {
        int r[5][3];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint8_t *R = S + (size_t)ys[k] * sd.img_cols * 3;
            uint32_t q[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) q[i] = load_px3(R + xs[i]);      // source levels live in the pyramid buffer (slack at its end)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                r[k][c] = px_ch(q[2], c) * 6 + (px_ch(q[1], c) + px_ch(q[3], c)) * 4 + px_ch(q[0], c) + px_ch(q[4], c);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) D[c] = (uint8_t)((r[2][c] * 6 + (r[1][c] + r[3][c]) * 4 + r[0][c] + r[4][c] + 128) >> 8);
        return;
    }
    for (int c = 0; c < cn; ++c) {
        int r[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const uint8_t *R = S + (size_t)ys[k] * sd.img_cols * cn + c;
            r[k] = R[xs[2]] * 6 + (R[xs[1]] + R[xs[3]]) * 4 + R[xs[0]] + R[xs[4]];
        }
        D[c] = (uint8_t)((r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4] + 128) >> 8);
    }
}

// The other depths: 16U integer as 8-bit; 32F / 64F the same taps in the pixel type, row = s2*6 + (s1+s3)*4 + s0 + s4,
// dst = (r2*6 + (r1+r3)*4 + r0 + r4) * (1/256).
template <typename PT, typename WT> __device__ __forceinline__ PT pyr_finish(WT v);
template <> __device__ __forceinline__ uint16_t pyr_finish<uint16_t, int>(int v) { return (uint16_t)((v + 128) >> 8); }
template <> __device__ __forceinline__ float pyr_finish<float, float>(float v) { return v * (1.f / 256.f); }
template <> __device__ __forceinline__ double pyr_finish<double, double>(double v) { return v * (1. / 256.); }

template <typename PT, typename WT>
__global__ __launch_bounds__(256) void k_pyrdown_t(PyrParams p, int first_level, int last_level, long long base, long long npix)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = find_level_blk<0>(p.lv, first_level, last_level, idx + base, s_off);
    if (idx >= npix) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const LevelDesc sd = p.lv[d.src_level];
    const int local = (int)(idx + base - d.img_off);
    const int y = local / d.img_cols, x = local - y * d.img_cols;
    const int cn = p.cn;
    const PT *S = reinterpret_cast<const PT *>(p.pyr) + ((size_t)frame * p.pix_per_frame + sd.img_off) * cn;
    PT *D = reinterpret_cast<PT *>(p.pyr) + ((size_t)frame * p.pix_per_frame + d.img_off + local) * cn;
    int xs[5], ys[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        xs[k] = reflect101(2 * x - 2 + k, sd.img_cols) * cn;
        ys[k] = reflect101(2 * y - 2 + k, sd.img_rows);
    }
    for (int c = 0; c < cn; ++c) {
        WT r[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const PT *R = S + (size_t)ys[k] * sd.img_cols * cn + c;
            r[k] = (WT)R[xs[2]] * 6 + ((WT)R[xs[1]] + (WT)R[xs[3]]) * 4 + (WT)R[xs[0]] + (WT)R[xs[4]];
        }
        D[c] = pyr_finish<PT, WT>(r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4]);
    }
}

void launch_pyrdown_range(const PyrParams &p, int nframes, int first_level, int last_level, long long base,
                          long long npix, hipStream_t s)
{
    dim3 grid((unsigned)((npix + 255) / 256), nframes);
    if (p.depth == kDepth16U) PBD_LAUNCH((k_pyrdown_t<uint16_t, int>), grid, dim3(256), 0, s, p, first_level, last_level, base, npix);
    else if (p.depth == kDepth32F) PBD_LAUNCH((k_pyrdown_t<float, float>), grid, dim3(256), 0, s, p, first_level, last_level, base, npix);
    else if (p.depth == kDepth64F) PBD_LAUNCH((k_pyrdown_t<double, double>), grid, dim3(256), 0, s, p, first_level, last_level, base, npix);
    else PBD_LAUNCH(k_pyrdown, grid, dim3(256), 0, s, p, first_level, last_level, base, npix);
}

// ------------------------------------------------------------------------------------------------
// HOG cell histograms, gather form (R = reference template parameter T).  One thread per block (cell of
// the `blocks` grid): it walks the source pixels that the reference's scatter loop
// (src/HOGFeatures.cpp:202-267) adds into this block, in the same raster order, so every bin sees the
// same sequence of additions.  Bins are 18 registers; the selected bin is updated through predicated
// adds of +0, which leave a non-negative sum unchanged.  Also writes the block energy (:270-283).
// ------------------------------------------------------------------------------------------------
template <typename R> __device__ __forceinline__ R real_sqrt(R v);
template <> __device__ __forceinline__ float real_sqrt<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double real_sqrt<double>(double v) { return sqrt(v); }

// Pass 1: per image pixel, the snapped orientation (0..17) and the gradient magnitude of the strongest
// colour channel (src/HOGFeatures.cpp:205-260).  Every pixel feeds four blocks, so this part is done once
// per pixel instead of once per (pixel, block).
template <typename R>
__global__ __launch_bounds__(256) void k_hog_grad(HogParams p)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = find_level_blk<0>(p.lv, 0, p.nlevels, idx, s_off);
    if (idx >= p.pix_per_frame) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.img_off);
    const int rows = d.img_rows, cols = d.img_cols;
    const int ys = local / cols, xs = local - ys * cols;
    const size_t o = (size_t)frame * p.pix_per_frame + idx;
    if (xs < 1 || ys < 1 || xs > cols - 2 || ys > rows - 2) return;     // never sampled (clamped to cols-2 / rows-2)
    const int cn = p.cn;
    const uint8_t *im = p.pyr + ((size_t)frame * p.pix_per_frame + d.img_off) * cn;
    const size_t stride = (size_t)cols * cn;
    const R uu[9] = {(R)1.000, (R)0.9397, (R)0.7660, (R)0.5000, (R)0.1736, (R)-0.1736, (R)-0.5000, (R)-0.7660, (R)-0.9397};
    const R vv[9] = {(R)0.000, (R)0.3420, (R)0.6428, (R)0.8660, (R)0.9848, (R)0.9848, (R)0.8660, (R)0.6428, (R)0.3420};
    R dx, dy, v;
    if (cn == 1) {
        const uint8_t *s = im + xs + (size_t)ys * stride;
        dy = (R)((int)s[stride] - (int)*(s - stride));
        dx = (R)((int)s[1] - (int)s[-1]);
        v = dx * dx + dy * dy;
    } else {
        const uint8_t *s = im + 3 * xs + (size_t)ys * stride;
        const uint32_t pd = load_px3(s + stride), pu = load_px3(s - stride), pr = load_px3(s + 3), pl = load_px3(s - 3);
        const R dyb = (R)(px_ch(pd, 0) - px_ch(pu, 0));
        const R dxb = (R)(px_ch(pr, 0) - px_ch(pl, 0));
        const R vb = dxb * dxb + dyb * dyb;
        const R dyg = (R)(px_ch(pd, 1) - px_ch(pu, 1));
        const R dxg = (R)(px_ch(pr, 1) - px_ch(pl, 1));
        const R vg = dxg * dxg + dyg * dyg;
        dy = (R)(px_ch(pd, 2) - px_ch(pu, 2));
        dx = (R)(px_ch(pr, 2) - px_ch(pl, 2));
        v = dx * dx + dy * dy;
        if (vg > v) { v = vg; dx = dxg; dy = dyg; }
        if (vb > v) { v = vb; dx = dxb; dy = dyb; }
    }
    R best_dot = (R)0;
    int best_o = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const R dot = uu[k] * dx + vv[k] * dy;
        if (dot > best_dot) { best_dot = dot; best_o = k; }
        else if (-dot > best_dot) { best_dot = -dot; best_o = k + 9; }
    }
    static_cast<R *>(p.gmag)[o] = real_sqrt<R>(v);
    p.gori[o] = (uint8_t)best_o;
}

// Four consecutive pixels per thread (8-bit BGR, the hot case): when the four share an image row and are all interior,
// the rows above / below come in as one 16-byte load each and the row itself as 16 + 4 bytes (10 memory instructions
// per four pixels instead of 24), the results leave as one 16-byte and one 4-byte store.  Same arithmetic per pixel.
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(1)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ int px_byte(const u32x4_u &v, uint32_t extra, int b)
{   // byte b (compile-time) of the 20 bytes {v, extra}
    const uint32_t w = b < 16 ? v[b >> 2] : extra;
    return (int)((w >> (8 * (b & 3))) & 0xffu);
}

template <typename R>
__device__ __forceinline__ void hog_grad_pixel(R dxb, R dyb, R dxg, R dyg, R dxr, R dyr, R &mag, int &ori)
{   // src/HOGFeatures.cpp:217-260: strongest channel (third channel first, then G, then B), 18-way orientation snap
    const R uu[9] = {(R)1.000, (R)0.9397, (R)0.7660, (R)0.5000, (R)0.1736, (R)-0.1736, (R)-0.5000, (R)-0.7660, (R)-0.9397};
    const R vv[9] = {(R)0.000, (R)0.3420, (R)0.6428, (R)0.8660, (R)0.9848, (R)0.9848, (R)0.8660, (R)0.6428, (R)0.3420};
    const R vb = dxb * dxb + dyb * dyb;
    const R vg = dxg * dxg + dyg * dyg;
    R dx = dxr, dy = dyr;
    R v = dx * dx + dy * dy;
    if (vg > v) { v = vg; dx = dxg; dy = dyg; }
    if (vb > v) { v = vb; dx = dxb; dy = dyb; }
    // The reference's scan, k ascending: "if (dot > best) {best = dot; o = k} else if (-dot > best) {best = -dot; o = k + 9}".
    // best is never negative, so at most one of the two tests can pass and the pair is "|dot| > best" with the sign of dot
    // choosing k or k + 9 (a NaN fails every test in both forms).  The tables are antisymmetric / symmetric about k = 4.5
    // (uu[9-j] = -uu[j], vv[9-j] = vv[j]) and rounding is sign-symmetric, so uu[9-j]*dx + vv[9-j]*dy is, bit for bit,
    // vv[j]*dy - uu[j]*dx: eight products and eight sums instead of eighteen and nine; uu[0]*dx + vv[0]*dy is 1*dx + 0*dy.
    R dots[9];
    dots[0] = uu[0] * dx + vv[0] * dy;
#pragma unroll
    for (int j = 1; j <= 4; ++j) {
        const R pa = uu[j] * dx, pb = vv[j] * dy;
        dots[j] = pa + pb;
        dots[9 - j] = pb - pa;
    }
    R best_dot = (R)0;
    int best_o = 0;
    if constexpr (sizeof(R) == 4) {
        // the winner is carried as k | sign bit of its dot product: one and-or per candidate, the k + 9 resolved at the end
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float ad = __builtin_fabsf(dots[k]);
            const bool win = ad > best_dot;
            code = win ? ((__float_as_uint(dots[k]) & 0x80000000u) | (unsigned)k) : code;
            best_dot = win ? ad : best_dot;
        }
        best_o = (int)(code & 0xfu) + ((code >> 31) ? 9 : 0);
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const R dot = dots[k];
            const R ad = dot < (R)0 ? -dot : dot;
            if (ad > best_dot) { best_dot = ad; best_o = dot < (R)0 ? k + 9 : k; }
        }
    }
    mag = real_sqrt<R>(v);
    ori = best_o;
}

__global__ __launch_bounds__(256) void k_hog_grad4(HogParams p)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int l = find_level_blk<0>(p.lv, 0, p.nlevels, min(idx, p.pix_per_frame - 1), s_off);
    if (idx >= p.pix_per_frame) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.img_off);
    const int rows = d.img_rows, cols = d.img_cols;
    const int ys = local / cols, xs = local - ys * cols;
    const size_t o = (size_t)frame * p.pix_per_frame + idx;
    float *gmag = static_cast<float *>(p.gmag);
    const bool fast = local + 3 < rows * cols && xs >= 1 && xs + 3 <= cols - 2 && ys >= 1 && ys <= rows - 2;
    if (fast) {
        const uint8_t *im = p.pyr + ((size_t)frame * p.pix_per_frame + d.img_off) * 3;
        const size_t stride = (size_t)cols * 3;
        const uint8_t *s = im + 3 * xs + (size_t)ys * stride;
        const u32x4_u up = *reinterpret_cast<const u32x4_u *>(s - stride), dn = *reinterpret_cast<const u32x4_u *>(s + stride);
        const u32x4_u mid = *reinterpret_cast<const u32x4_u *>(s - 3);
        const uint32_t mid2 = *reinterpret_cast<const u32_unaligned *>(s + 13);       // bytes 16..19 of the row window
        f32x4_u mg;
        uint32_t og = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // pixel i: up / down at bytes 3i+c, left at window bytes 3i+c, right at 3(i+2)+c
            float m; int oo;
            hog_grad_pixel<float>((float)(px_byte(mid, mid2, 3 * (i + 2) + 0) - px_byte(mid, mid2, 3 * i + 0)),
                                  (float)(px_byte(dn, 0, 3 * i + 0) - px_byte(up, 0, 3 * i + 0)),
                                  (float)(px_byte(mid, mid2, 3 * (i + 2) + 1) - px_byte(mid, mid2, 3 * i + 1)),
                                  (float)(px_byte(dn, 0, 3 * i + 1) - px_byte(up, 0, 3 * i + 1)),
                                  (float)(px_byte(mid, mid2, 3 * (i + 2) + 2) - px_byte(mid, mid2, 3 * i + 2)),
                                  (float)(px_byte(dn, 0, 3 * i + 2) - px_byte(up, 0, 3 * i + 2)), m, oo);
            mg[i] = m;
            og |= (uint32_t)oo << (8 * i);
        }
        *reinterpret_cast<f32x4_u *>(gmag + o) = mg;
        *reinterpret_cast<u32_unaligned *>(p.gori + o) = og;
        return;
    }
    // quads that touch an image border, wrap to the next row or cross into the next level: pixel by pixel
    for (int i = 0; i < 4; ++i) {
        const long long pi = idx + i;
        if (pi >= p.pix_per_frame) return;
        int li = l;
        while (li + 1 < p.nlevels && p.lv[li + 1].img_off <= pi) ++li;
        const LevelDesc di = p.lv[li];
        const int loc = (int)(pi - di.img_off);
        const int r2 = di.img_rows, c2 = di.img_cols;
        const int y = loc / c2, x = loc - y * c2;
        if (x < 1 || y < 1 || x > c2 - 2 || y > r2 - 2) continue;     // never sampled (clamped to cols-2 / rows-2)
        const uint8_t *im = p.pyr + ((size_t)frame * p.pix_per_frame + di.img_off) * 3;
        const size_t stride = (size_t)c2 * 3;
        const uint8_t *s = im + 3 * x + (size_t)y * stride;
        const uint32_t pd = load_px3(s + stride), pu = load_px3(s - stride), pr = load_px3(s + 3), pl = load_px3(s - 3);
        float m; int oo;
        hog_grad_pixel<float>((float)(px_ch(pr, 0) - px_ch(pl, 0)), (float)(px_ch(pd, 0) - px_ch(pu, 0)),
                              (float)(px_ch(pr, 1) - px_ch(pl, 1)), (float)(px_ch(pd, 1) - px_ch(pu, 1)),
                              (float)(px_ch(pr, 2) - px_ch(pl, 2)), (float)(px_ch(pd, 2) - px_ch(pu, 2)), m, oo);
        gmag[(size_t)frame * p.pix_per_frame + pi] = m;
        p.gori[(size_t)frame * p.pix_per_frame + pi] = (uint8_t)oo;
    }
}

// The same for 16U / 32F / 64F pixels (features<uint16_t|float|double>, src/HOGFeatures.cpp:136-146): the difference is
// taken in the pixel type (integers promote to int, float / double subtract as such) and then converted to T.
template <typename R, typename PT> __device__ __forceinline__ R pix_diff(PT a, PT b);
template <> __device__ __forceinline__ float pix_diff<float, uint16_t>(uint16_t a, uint16_t b) { return (float)((int)a - (int)b); }
template <> __device__ __forceinline__ double pix_diff<double, uint16_t>(uint16_t a, uint16_t b) { return (double)((int)a - (int)b); }
template <> __device__ __forceinline__ float pix_diff<float, float>(float a, float b) { return a - b; }
template <> __device__ __forceinline__ double pix_diff<double, float>(float a, float b) { return (double)(a - b); }
template <> __device__ __forceinline__ float pix_diff<float, double>(double a, double b) { return (float)(a - b); }
template <> __device__ __forceinline__ double pix_diff<double, double>(double a, double b) { return a - b; }

template <typename R, typename PT>
__global__ __launch_bounds__(256) void k_hog_grad_t(HogParams p)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = find_level_blk<0>(p.lv, 0, p.nlevels, idx, s_off);
    if (idx >= p.pix_per_frame) return;
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.img_off);
    const int rows = d.img_rows, cols = d.img_cols;
    const int ys = local / cols, xs = local - ys * cols;
    const size_t o = (size_t)frame * p.pix_per_frame + idx;
    if (xs < 1 || ys < 1 || xs > cols - 2 || ys > rows - 2) return;
    const int cn = p.cn;
    const PT *im = reinterpret_cast<const PT *>(p.pyr) + ((size_t)frame * p.pix_per_frame + d.img_off) * cn;
    const size_t stride = (size_t)cols * cn;
    const R uu[9] = {(R)1.000, (R)0.9397, (R)0.7660, (R)0.5000, (R)0.1736, (R)-0.1736, (R)-0.5000, (R)-0.7660, (R)-0.9397};
    const R vv[9] = {(R)0.000, (R)0.3420, (R)0.6428, (R)0.8660, (R)0.9848, (R)0.9848, (R)0.8660, (R)0.6428, (R)0.3420};
    R dx, dy, v;
    if (cn == 1) {
        const PT *s = im + xs + (size_t)ys * stride;
        dy = pix_diff<R, PT>(s[stride], *(s - stride));
        dx = pix_diff<R, PT>(s[1], s[-1]);
        v = dx * dx + dy * dy;
    } else {
        const PT *s = im + 3 * xs + (size_t)ys * stride;
        const R dyb = pix_diff<R, PT>(s[stride], *(s - stride));
        const R dxb = pix_diff<R, PT>(s[3], s[-3]);
        const R vb = dxb * dxb + dyb * dyb;
        const R dyg = pix_diff<R, PT>(s[1 + stride], *(s + 1 - stride));
        const R dxg = pix_diff<R, PT>(s[4], s[-2]);
        const R vg = dxg * dxg + dyg * dyg;
        dy = pix_diff<R, PT>(s[2 + stride], *(s + 2 - stride));
        dx = pix_diff<R, PT>(s[5], s[-1]);
        v = dx * dx + dy * dy;
        if (vg > v) { v = vg; dx = dxg; dy = dyg; }
        if (vb > v) { v = vb; dx = dxb; dy = dyb; }
    }
    R best_dot = (R)0;
    int best_o = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const R dot = uu[k] * dx + vv[k] * dy;
        if (dot > best_dot) { best_dot = dot; best_o = k; }
        else if (-dot > best_dot) { best_dot = -dot; best_o = k + 9; }
    }
    static_cast<R *>(p.gmag)[o] = real_sqrt<R>(v);
    p.gori[o] = (uint8_t)best_o;
}

// Pass 2: one thread per block walks its source pixels in raster order and adds (wy*wx)*mag into the bin
// of the pixel's orientation.  The 18 bins of a thread live in LDS ([18][256], the thread always hits bank
// tid % 32), so the update is one read-add-write instead of 18 predicated register adds; the sequence of
// float additions per bin is the reference's.  Also writes the block energy (:270-283).
// SB = compile-time sbin (4, 8) or 0 for any: with a known sbin the x weights of the thread's window are fetched
// once into registers instead of once per source pixel (the coordinate-table loads were most of the kernel's
// memory requests: lanes are blocks, their table entries sbin apart).
template <typename R, int SB>
__global__ __launch_bounds__(256) void k_hog_hist(HogParams p)
{
    __shared__ R bins[18 * 256];
    const long long idx0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = idx0 < p.blk_per_frame;
    const long long idx = active ? idx0 : p.blk_per_frame - 1;
    const int frame = p.frame0 + blockIdx.y;
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const int l = find_level_blk<1>(p.lv, 0, p.nlevels, idx, s_off);
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.blk_off);
    const int by = local / d.blk_cols, bx = local - by * d.blk_cols;
    const int sbin = p.sbin;
    const int rows = d.img_rows, cols = d.img_cols;
    const int vish = d.blk_rows * sbin, visw = d.blk_cols * sbin;
    const R *gmag = static_cast<const R *>(p.gmag) + (size_t)frame * p.pix_per_frame + d.img_off;
    const uint8_t *gori = p.gori + (size_t)frame * p.pix_per_frame + d.img_off;
    const HogCoordT<R> *coord = static_cast<const HogCoordT<R> *>(p.coord);
    R *h = bins + threadIdx.x;
#pragma unroll
    for (int o = 0; o < 18; ++o) h[o * 256] = (R)0;

    // pixels with ip in {b-1, b}: (y+0.5)/sbin - 0.5 in [b-1, b+1), i.e. y in [sbin*b - sbin/2 - 0.5, sbin*b + 3*sbin/2 - 0.5);
    // one extra pixel either side, the table test below decides
    int ylo = sbin * by - (sbin + 1) / 2 - 1, yhi = sbin * by + (3 * sbin + 1) / 2 + 1;
    int xlo = sbin * bx - (sbin + 1) / 2 - 1, xhi = sbin * bx + (3 * sbin + 1) / 2 + 1;
    if (ylo < 1) ylo = 1;
    if (xlo < 1) xlo = 1;
    if (yhi > vish - 1) yhi = vish - 1;
    if (xhi > visw - 1) xhi = visw - 1;
    if (!active) yhi = ylo;

    if constexpr (SB > 0) {
        constexpr int NX = 2 * SB + 2;                         // window width before clamping
        const int x0 = SB * bx - (SB + 1) / 2 - 1;
        R wxs[NX];
        int xoff[NX];                                           // source column, -1: this x does not feed the block
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int x = x0 + i;
            xoff[i] = -1; wxs[i] = (R)0;
            if (x >= xlo && x < xhi) {
                const HogCoordT<R> cx = coord[x];
                if (cx.ip == bx) { wxs[i] = cx.v1; xoff[i] = x < cols - 2 ? x : cols - 2; }
                else if (cx.ip + 1 == bx) { wxs[i] = cx.v0; xoff[i] = x < cols - 2 ? x : cols - 2; }
            }
        }
        for (int y = ylo; y < yhi; ++y) {
            const HogCoordT<R> cy = coord[y];
            R wy;
            if (cy.ip == by) wy = cy.v1;
            else if (cy.ip + 1 == by) wy = cy.v0;
            else continue;
            const size_t rowg = (size_t)(y < rows - 2 ? y : rows - 2) * cols;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if (xoff[i] < 0) continue;
                const size_t g = rowg + xoff[i];
                const R contrib = (wy * wxs[i]) * gmag[g];
                R *bin = h + (int)gori[g] * 256;
                *bin = *bin + contrib;
            }
        }
    } else
    for (int y = ylo; y < yhi; ++y) {
        const HogCoordT<R> cy = coord[y];
        R wy;
        if (cy.ip == by) wy = cy.v1;            // this block is (iyp, .): weight vy1
        else if (cy.ip + 1 == by) wy = cy.v0;   // this block is (iyp+1, .): weight vy0
        else continue;
        const int ys = y < rows - 2 ? y : rows - 2;
        for (int x = xlo; x < xhi; ++x) {
            const HogCoordT<R> cx = coord[x];
            R wx;
            if (cx.ip == bx) wx = cx.v1;
            else if (cx.ip + 1 == bx) wx = cx.v0;
            else continue;
            const int xs = x < cols - 2 ? x : cols - 2;
            const size_t g = (size_t)ys * cols + xs;
            // the four scatter lines multiply (vy?*vx?) first, then by v; products commute
            const R contrib = (wy * wx) * gmag[g];
            R *bin = h + (int)gori[g] * 256;
            *bin = *bin + contrib;
        }
    }
    if (!active) return;
    R *hist = static_cast<R *>(p.hist) + (size_t)frame * 18 * p.blk_per_frame + idx;
    R hv[18];
#pragma unroll
    for (int o = 0; o < 18; ++o) { hv[o] = h[o * 256]; hist[(size_t)o * p.blk_per_frame] = hv[o]; }
    R e = (R)0;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const R t = hv[o] + hv[o + 9];
        e += t * t;
    }
    static_cast<R *>(p.norm)[(size_t)frame * p.blk_per_frame + idx] = e;
}

// Fused form of the two passes for the hot case (8-bit BGR frames, T = float, sbin 4 or 8): one workgroup = one tile of
// TBX x TBY blocks.  The gradient magnitude / orientation of the pixels its blocks sample ((TB-1)*SB + 2*SB + 2 per side)
// are computed straight from the level image into LDS, four pixels per lane as in k_hog_grad4, and the block threads
// then walk their windows there: the 5 bytes per pixel of pass 1 never travel to HBM and back (they were 2/3 of the
// two kernels' traffic, each pixel being re-read by four blocks), and the window reads, SB apart between lanes in
// global memory, become LDS reads.  Arithmetic and addition order per bin are those of the two kernels above.
template <int SB, int TBX, int TBY>
__global__ __launch_bounds__(TBX * TBY) void k_hog_tile(HogParams p)
{
    constexpr int NT = TBX * TBY;
    constexpr int LO = (SB + 1) / 2 + 1;                           // pixels sampled left of / above the tile's first block
    constexpr int PWX = SB * (TBX - 1) + (3 * SB + 1) / 2 + 1 + LO, PWY = SB * (TBY - 1) + (3 * SB + 1) / 2 + 1 + LO;
    constexpr int LW = (PWX + 3) & ~3, NQ = LW / 4;
    __shared__ float s_mag[PWY * LW];
    __shared__ uint8_t s_ori[PWY * LW];
    __shared__ float bins[18 * NT];
    const ConvTile tile = p.htiles[blockIdx.x];
    const int frame = p.frame0 + blockIdx.y;
    const LevelDesc d = p.lv[tile.level];
    const int rows = d.img_rows, cols = d.img_cols;
    const int t = threadIdx.x;
    const int ox = SB * tile.x0 - LO, oy = SB * tile.y0 - LO;     // image position of LDS cell (0, 0); ConvTile::{y0, x0} = first block
    const uint8_t *im = p.pyr + ((size_t)frame * p.pix_per_frame + d.img_off) * 3;
    const size_t stride = (size_t)cols * 3;
    const long long npix = (long long)rows * cols;
    for (int q = t; q < PWY * NQ; q += NT) {
        const int py = q / NQ, px = (q - py * NQ) * 4;
        const int y = oy + py, x = ox + px;
        if (y < 1 || y > rows - 2) continue;                       // never sampled (positions are clamped to rows-2 / cols-2)
        const uint8_t *s = im + 3 * x + (size_t)y * stride;
        if (x >= 1 && x + 3 <= cols - 2 && (long long)y * cols + x + 3 < npix) {
            const u32x4_u up = *reinterpret_cast<const u32x4_u *>(s - stride), dn = *reinterpret_cast<const u32x4_u *>(s + stride);
            const u32x4_u mid = *reinterpret_cast<const u32x4_u *>(s - 3);
            const uint32_t mid2 = *reinterpret_cast<const u32_unaligned *>(s + 13);
            f32x4_u mg;
            uint32_t og = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m; int oo;
                hog_grad_pixel<float>((float)(px_byte(mid, mid2, 3 * (i + 2) + 0) - px_byte(mid, mid2, 3 * i + 0)),
                                      (float)(px_byte(dn, 0, 3 * i + 0) - px_byte(up, 0, 3 * i + 0)),
                                      (float)(px_byte(mid, mid2, 3 * (i + 2) + 1) - px_byte(mid, mid2, 3 * i + 1)),
                                      (float)(px_byte(dn, 0, 3 * i + 1) - px_byte(up, 0, 3 * i + 1)),
                                      (float)(px_byte(mid, mid2, 3 * (i + 2) + 2) - px_byte(mid, mid2, 3 * i + 2)),
                                      (float)(px_byte(dn, 0, 3 * i + 2) - px_byte(up, 0, 3 * i + 2)), m, oo);
                mg[i] = m;
                og |= (uint32_t)oo << (8 * i);
            }
            *reinterpret_cast<f32x4_u *>(s_mag + py * LW + px) = mg;
            *reinterpret_cast<uint32_t *>(s_ori + py * LW + px) = og;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xi = x + i;
                if (xi < 1 || xi > cols - 2) continue;
                const uint8_t *si = s + 3 * i;
                const uint32_t pd = load_px3(si + stride), pu = load_px3(si - stride), pr = load_px3(si + 3), pl = load_px3(si - 3);
                float m; int oo;
                hog_grad_pixel<float>((float)(px_ch(pr, 0) - px_ch(pl, 0)), (float)(px_ch(pd, 0) - px_ch(pu, 0)),
                                      (float)(px_ch(pr, 1) - px_ch(pl, 1)), (float)(px_ch(pd, 1) - px_ch(pu, 1)),
                                      (float)(px_ch(pr, 2) - px_ch(pl, 2)), (float)(px_ch(pd, 2) - px_ch(pu, 2)), m, oo);
                s_mag[py * LW + px + i] = m;
                s_ori[py * LW + px + i] = (uint8_t)oo;
            }
        }
    }
    __syncthreads();
    const int by = tile.y0 + t / TBX, bx = tile.x0 + t % TBX;
    if (by >= d.blk_rows || bx >= d.blk_cols) return;
    const int vish = d.blk_rows * SB, visw = d.blk_cols * SB;
    const HogCoordT<float> *coord = static_cast<const HogCoordT<float> *>(p.coord);
    float *h = bins + t;
#pragma unroll
    for (int o = 0; o < 18; ++o) h[o * NT] = 0.0f;
    int ylo = SB * by - (SB + 1) / 2 - 1, yhi = SB * by + (3 * SB + 1) / 2 + 1;
    int xlo = SB * bx - (SB + 1) / 2 - 1, xhi = SB * bx + (3 * SB + 1) / 2 + 1;
    if (ylo < 1) ylo = 1;
    if (xlo < 1) xlo = 1;
    if (yhi > vish - 1) yhi = vish - 1;
    if (xhi > visw - 1) xhi = visw - 1;
    constexpr int NX = 2 * SB + 2;                             // window width before clamping
    const int x0 = SB * bx - (SB + 1) / 2 - 1;
    float wxs[NX];
    int xoff[NX];                                               // LDS column, -1: this x does not feed the block
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int x = x0 + i;
        xoff[i] = -1; wxs[i] = 0.0f;
        if (x >= xlo && x < xhi) {
            const HogCoordT<float> cx = coord[x];
            if (cx.ip == bx) { wxs[i] = cx.v1; xoff[i] = (x < cols - 2 ? x : cols - 2) - ox; }
            else if (cx.ip + 1 == bx) { wxs[i] = cx.v0; xoff[i] = (x < cols - 2 ? x : cols - 2) - ox; }
        }
    }
    for (int y = ylo; y < yhi; ++y) {
        const HogCoordT<float> cy = coord[y];
        float wy;
        if (cy.ip == by) wy = cy.v1;
        else if (cy.ip + 1 == by) wy = cy.v0;
        else continue;
        const int rowg = ((y < rows - 2 ? y : rows - 2) - oy) * LW;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (xoff[i] < 0) continue;
            const int g = rowg + xoff[i];
            const float contrib = (wy * wxs[i]) * s_mag[g];
            float *bin = h + (int)s_ori[g] * NT;
            *bin = *bin + contrib;
        }
    }
    const long long idx = d.blk_off + (long long)by * d.blk_cols + bx;
    float *hist = static_cast<float *>(p.hist) + (size_t)frame * 18 * p.blk_per_frame + idx;
    float hv[18];
#pragma unroll
    for (int o = 0; o < 18; ++o) { hv[o] = h[o * NT]; hist[(size_t)o * p.blk_per_frame] = hv[o]; }
    float e = 0.0f;
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const float tt = hv[o] + hv[o + 9];
        e += tt * tt;
    }
    static_cast<float *>(p.norm)[(size_t)frame * p.blk_per_frame + idx] = e;
}

void launch_hog_hist(const HogParams &p, int nframes, bool f64, hipStream_t s)
{
    static const bool two_pass = getenv("PBD_HOG_TWO_PASS") != nullptr;     // A/B switch: the unfused kernels
    if (!f64 && p.depth == kDepth8U && p.cn == 3 && (p.sbin == 4 || p.sbin == 8) && !two_pass) {
        if (p.nhtiles == 0) return;
        dim3 grid((unsigned)p.nhtiles, nframes);
        if (p.sbin == 4) PBD_LAUNCH((k_hog_tile<4, kHogTBX, 16>), grid, dim3(kHogTBX * 16), 0, s, p);
        else PBD_LAUNCH((k_hog_tile<8, kHogTBX, 8>), grid, dim3(kHogTBX * 8), 0, s, p);
        return;
    }
    dim3 gridp((unsigned)((p.pix_per_frame + 255) / 256), nframes);
#define PBD_GRAD(PT)                                                                            \
    do {                                                                                        \
        if (f64) PBD_LAUNCH((k_hog_grad_t<double, PT>), gridp, dim3(256), 0, s, p);     \
        else PBD_LAUNCH((k_hog_grad_t<float, PT>), gridp, dim3(256), 0, s, p);          \
    } while (0)
    if (p.depth == kDepth16U) PBD_GRAD(uint16_t);
    else if (p.depth == kDepth32F) PBD_GRAD(float);
    else if (p.depth == kDepth64F) PBD_GRAD(double);
    else if (f64) PBD_LAUNCH(k_hog_grad<double>, gridp, dim3(256), 0, s, p);
    else if (p.cn == 3) {
        dim3 grid4((unsigned)((p.pix_per_frame + 1023) / 1024), nframes);
        PBD_LAUNCH(k_hog_grad4, grid4, dim3(256), 0, s, p);
    } else PBD_LAUNCH(k_hog_grad<float>, gridp, dim3(256), 0, s, p);
#undef PBD_GRAD
    dim3 grid((unsigned)((p.blk_per_frame + 255) / 256), nframes);
    if (f64) PBD_LAUNCH((k_hog_hist<double, 0>), grid, dim3(256), 0, s, p);
    else if (p.sbin == 4) PBD_LAUNCH((k_hog_hist<float, 4>), grid, dim3(256), 0, s, p);
    else if (p.sbin == 8) PBD_LAUNCH((k_hog_hist<float, 8>), grid, dim3(256), 0, s, p);
    else PBD_LAUNCH((k_hog_hist<float, 0>), grid, dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// Normalisation and the 32 output channels per interior cell (src/HOGFeatures.cpp:286-340).
// One thread per cell.  The four normalisers are evaluated in double as the reference does.
// ------------------------------------------------------------------------------------------------
template <typename R>
__device__ __forceinline__ R hog_norm(const R *n, int stride)
{
    const R s4 = ((n[0] + n[1]) + n[stride]) + n[stride + 1];
    return (R)(1.0 / sqrt((double)s4 + 0.0001));
}

template <typename R>
__global__ __launch_bounds__(256) void k_hog_feat(HogParams p)
{
    __shared__ long long s_off[PBD_MAX_LEVELS];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int l = find_level_blk<2>(p.lv, 0, p.nlevels, idx, s_off);
    // float: the 32 values of a cell leave through LDS so that a store instruction writes 1 KB of consecutive addresses (a
    // block's 256 cells are consecutive in the feature buffer); written straight from the registers every 16-byte store of a
    // wave touched 64 different 128-byte lines
    constexpr bool VIA_LDS = sizeof(R) == 4;
    constexpr int kPitch = 36;                       // floats per cell in LDS: 16-byte aligned, 4 banks apart
    __shared__ __attribute__((aligned(16))) float tile[VIA_LDS ? 256 * kPitch : 4];
    const bool valid = idx < p.cell_per_frame;
    if (!VIA_LDS && !valid) return;
    const int frame = p.frame0 + blockIdx.y;
    if (valid) {
    const LevelDesc d = p.lv[l];
    const int local = (int)(idx - d.cell_off);
    const int y = local / d.cols, x = local - y * d.cols;
    const int bw = d.blk_cols;
    const R *norm = static_cast<const R *>(p.norm) + (size_t)frame * p.blk_per_frame + d.blk_off;
    const R n1 = hog_norm<R>(norm + (size_t)(y + 1) * bw + (x + 1), bw);
    const R n2 = hog_norm<R>(norm + (size_t)y * bw + (x + 1), bw);
    const R n3 = hog_norm<R>(norm + (size_t)(y + 1) * bw + x, bw);
    const R n4 = hog_norm<R>(norm + (size_t)y * bw + x, bw);
    const R *hist = static_cast<const R *>(p.hist) + (size_t)frame * 18 * p.blk_per_frame + d.blk_off + (size_t)(y + 1) * bw + (x + 1);
    R hv[18];
#pragma unroll
    for (int o = 0; o < 18; ++o) hv[o] = hist[(size_t)o * p.blk_per_frame];

    const R lim = (R)0.2;
    R out[32];
    R t1 = (R)0, t2 = (R)0, t3 = (R)0, t4 = (R)0;
#pragma unroll
    for (int o = 0; o < 18; ++o) {
        const R val = hv[o];
        R h1 = val * n1; h1 = lim < h1 ? lim : h1;
        R h2 = val * n2; h2 = lim < h2 ? lim : h2;
        R h3 = val * n3; h3 = lim < h3 ? lim : h3;
        R h4 = val * n4; h4 = lim < h4 ? lim : h4;
        out[o] = (R)(0.5 * (double)(((h1 + h2) + h3) + h4));
        t1 += h1; t2 += h2; t3 += h3; t4 += h4;
    }
#pragma unroll
    for (int o = 0; o < 9; ++o) {
        const R sum = hv[o] + hv[o + 9];
        R h1 = sum * n1; h1 = lim < h1 ? lim : h1;
        R h2 = sum * n2; h2 = lim < h2 ? lim : h2;
        R h3 = sum * n3; h3 = lim < h3 ? lim : h3;
        R h4 = sum * n4; h4 = lim < h4 ? lim : h4;
        out[18 + o] = (R)(0.5 * (double)(((h1 + h2) + h3) + h4));
    }
    out[27] = (R)(0.2357 * (double)t1);
    out[28] = (R)(0.2357 * (double)t2);
    out[29] = (R)(0.2357 * (double)t3);
    out[30] = (R)(0.2357 * (double)t4);
    out[31] = (R)0;
    typedef R rv4 __attribute__((ext_vector_type(4)));
    if constexpr (VIA_LDS) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<rv4 *>(tile + threadIdx.x * kPitch + 4 * i) = rv4{out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]};
    } else {
        R *dst = static_cast<R *>(p.feat) + ((size_t)frame * p.cell_per_frame + idx) * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) reinterpret_cast<rv4 *>(dst)[i] = rv4{out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]};
    }
    }
    if constexpr (VIA_LDS) {
        __syncthreads();
        const long long idx0 = (long long)blockIdx.x * blockDim.x;
        const int nq = (int)min((long long)256, p.cell_per_frame - idx0) * 8;          // 16-byte pieces of this block
        typedef float fv4 __attribute__((ext_vector_type(4)));
        fv4 *dst = reinterpret_cast<fv4 *>(static_cast<float *>(p.feat) + ((size_t)frame * p.cell_per_frame + idx0) * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = (int)threadIdx.x + 256 * i;
            if (j < nq) dst[j] = *reinterpret_cast<const fv4 *>(tile + (j >> 3) * kPitch + (j & 7) * 4);
        }
    }
}



void launch_hog_feat(const HogParams &p, int nframes, bool f64, hipStream_t s)
{
    if (p.cell_per_frame == 0) return;
    dim3 grid((unsigned)((p.cell_per_frame + 255) / 256), nframes);
    if (f64) PBD_LAUNCH(k_hog_feat<double>, grid, dim3(256), 0, s, p);
    else PBD_LAUNCH(k_hog_feat<float>, grid, dim3(256), 0, s, p);
}

}  // namespace pbd
