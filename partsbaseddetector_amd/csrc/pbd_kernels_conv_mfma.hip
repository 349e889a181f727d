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
// Mapping: workgroup = 32 x 8 pixel tile of one level/frame x 160 filters; 4 waves, each two pixel rows.
//   A operand (M = 32 filters): weights of one tap, from LDS  [filter][hi 32ch | lo 32ch | pad]  (144 B)
//   B operand (N = 32 pixels along x): features, from the LDS tile [cell][hi 32ch | lo 32ch | pad] (144 B)
//   the 144-byte record stride makes the 16-byte fragment reads of 32 consecutive records bank-conflict free;
//   D[filter][pixel]: lanes = pixels, registers = filters -> each store instruction writes 32 consecutive x.
// The weights of tap t+1 are fetched from HBM/L2 into registers while tap t is computed.
//
// PBD_CONV_MFMA_F16 (BASELINE.json configs[4], SURVEY.md section 8(d) "Config 5"): the same contraction with
// every operand rounded once to fp16 and ONE v_mfma_f32_32x32x16_f16 per product tile, fp32 accumulation.
// Records are 80 bytes (32 fp16 + 16 pad, again conflict-free for the 16-byte fragment reads), a third of
// the matrix-core work and 47 KB of LDS per workgroup (three resident workgroups per CU instead of one).
// The 1e-4 score bar does NOT hold in this mode; tests report the error and the detection agreement.
#include "pbd_internal.h"

namespace pbd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kMfmaFB = 160;           // filters per pass (5 M-tiles of 32)

template <int K, bool F16>
__global__ __launch_bounds__(256) void k_conv_mfma(ConvParams p, const unsigned char *__restrict__ wrec,
                                                   const float *__restrict__ featp, float *__restrict__ respp)
{
    constexpr int kRec = F16 ? kMfmaRecBytesF16 : kMfmaRecBytes;   // bytes per LDS record: 64 hi + 64 lo + 16 pad | 64 fp16 + 16 pad
    constexpr int kRecU4 = kRec / 16;                              // 16-byte chunks per record
    constexpr int TW = kConvTW, TH = kConvTH;
    constexpr int PW = TW + K - 1, PH = TH + K - 1;
    constexpr int NCELL = PW * PH;
    constexpr int MT = kMfmaFB / 32;
    __shared__ __attribute__((aligned(16))) unsigned char sm_f[NCELL * kRec];
    __shared__ __attribute__((aligned(16))) unsigned char sm_w[kMfmaFB * kRec];

    const ConvTile tile = p.tiles[blockIdx.x];
    const int frame = p.frame0 + blockIdx.z;
    const int pass = blockIdx.y;                       // block of 160 filters
    const LevelDesc d = p.lv[tile.level];
    const int H = d.rows, W = d.cols;
    constexpr int a = K / 2;
    const int t = threadIdx.x;
    const float *feat = featp + ((size_t)frame * p.cell_per_frame + d.cell_off) * 32;

    // ---- stage the feature tile: fp32 -> (hi, lo) bf16; thread = (cell, 8-channel group)
    for (int idx = t; idx < NCELL * 4; idx += 256) {
        const int ci = idx >> 2, cg = idx & 3;
        const int cy = ci / PW, cx = ci - cy * PW;
        const int gy = tile.y0 + cy - a, gx = tile.x0 + cx - a;
        float v[8];
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const float4 *src = reinterpret_cast<const float4 *>(feat + ((size_t)gy * W + gx) * 32 + cg * 8);
            const float4 v0 = src[0], v1 = src[1];
            v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.0f;
            if (cg == 3) v[7] = 1.0f;                  // constant border: 1 on channel 31 (SpatialConvolutionEngine.cpp:153-156)
        }
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

    const int lane = t & 63, wave = t >> 6;
    const int r = lane & 31, hh = lane >> 5;
    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

    // weights: [pass][tap][160 filters][kRec B]; one tap = 160*9 = 1440 (fp16: 800) 16-byte chunks, NQ per thread (last partial)
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(wrec) + (size_t)pass * (K * K) * (kMfmaFB * kRecU4);
    constexpr int WCH = kMfmaFB * kRecU4;
    constexpr int NQ = (WCH + 255) / 256;
    u32x4 wreg[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int ch = t + q * 256;
        wreg[q] = ch < WCH ? wsrc[ch] : u32x4{0, 0, 0, 0};
    }
    for (int tp = 0; tp < K * K; ++tp) {
        __syncthreads();                               // previous tap's fragment reads are done (and the tile is staged)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ch = t + q * 256;
            if (ch < WCH) reinterpret_cast<u32x4 *>(sm_w)[ch] = wreg[q];
        }
        __syncthreads();
        if (tp + 1 < K * K) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ch = t + q * 256;
                if (ch < WCH) wreg[q] = wsrc[(size_t)(tp + 1) * WCH + ch];
            }
        }
        const int ti = tp / K, tj = tp - ti * K;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {               // two k-steps of 16 channels
            const int coff = kh * 32 + hh * 16;        // byte offset of this lane's 8 channels inside the hi (or lo) half
            if constexpr (F16) {
                f16x8 bf[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int cell = (wave * 2 + n + ti) * PW + (r + tj);
                    bf[n] = *reinterpret_cast<const f16x8 *>(sm_f + cell * kRec + coff);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f16x8 af = *reinterpret_cast<const f16x8 *>(sm_w + (m * 32 + r) * kRec + coff);
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[n], acc[m][n], 0, 0, 0);
                }
            } else {
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int cell = (wave * 2 + n + ti) * PW + (r + tj);
                    bh[n] = *reinterpret_cast<const bf16x8 *>(sm_f + cell * kRec + coff);
                    bl[n] = *reinterpret_cast<const bf16x8 *>(sm_f + cell * kRec + 64 + coff);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sm_w + (m * 32 + r) * kRec + coff);
                    const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sm_w + (m * 32 + r) * kRec + 64 + coff);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[n], acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
    }

    // D layout (32x32): column = lane & 31 (pixel), row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5) (filter)
    const int x = tile.x0 + r;
    const size_t HW = (size_t)H * W;
    float *resp = respp + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F;
    if (x < W) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int y = tile.y0 + wave * 2 + n;
            if (y < H) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int f = pass * kMfmaFB + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (f < p.F) resp[(size_t)f * HW + (size_t)y * W + x] = acc[m][n][e];
                    }
            }
        }
    }
}

void launch_conv_mfma(const ConvParams &p, const void *wrec, bool f16, int nframes, hipStream_t s)
{
    if (p.ntiles == 0 || p.F == 0) return;
    const int passes = (p.F + kMfmaFB - 1) / kMfmaFB;
    dim3 grid(p.ntiles, passes, nframes);
    if (f16)
        hipLaunchKernelGGL((k_conv_mfma<5, true>), grid, dim3(256), 0, s, p, static_cast<const unsigned char *>(wrec),
                           static_cast<const float *>(p.feat), static_cast<float *>(p.resp));
    else
        hipLaunchKernelGGL((k_conv_mfma<5, false>), grid, dim3(256), 0, s, p, static_cast<const unsigned char *>(wrec),
                           static_cast<const float *>(p.feat), static_cast<float *>(p.resp));
}

}  // namespace pbd
