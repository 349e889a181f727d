/*
 * pbd_oracle.c -- CPU restatement ("oracle") of the PartsBasedDetector detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg as the checker / timed CPU baseline.  Never part of the product path.
 * PARITY UNPINNED: see pbd_oracle.h.
 *
 * Build: oracle/Makefile (gcc -O2 -ftree-vectorize -msse4.1 -fopenmp -ffp-contract=off; the
 * reference's RelWithDebInfo flags are -O2 -g -DNDEBUG -msse4.1 -fopenmp, CMakeLists.txt:56-81).
 */
#define _GNU_SOURCE
#include "pbd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void pbdo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int pbdo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* cvRound: round half to even (SSE cvtsd2si under the default rounding mode) */
static inline int cv_round(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------------------------------
 * Pyramid geometry.  src/HOGFeatures.cpp:95-127; ctor include/HOGFeatures.hpp:74-81.
 * Overload resolution assumed: C++11 <cmath> (pow(float,float)/log(float) -> float versions,
 * pow(float,int) -> double).
 * ---------------------------------------------------------------------------------------- */
int pbdo_pyramid_plan(int rows, int cols, int sbin, int interval, int *lvl_rows, int *lvl_cols, float *scales)
{
    const float sfactor = powf(2.0f, 1.0f / (float)interval); /* HOGFeatures.hpp:78 */
    const float h = (float)rows, w = (float)cols;             /* Size_<float> imsize, :98 */
    const float mn = h < w ? h : w;
    /* :99 */
    const float ns = 1 + floorf(logf(mn / (5.0f * (float)sbin)) / logf(sfactor));
    if (!(ns >= 1)) return 0;
    const int nscales = (int)ns;
    if (nscales > PBDO_MAX_LEVELS) return -1;
    if (nscales < interval) return -2; /* the reference writes pyraimages[i] out of bounds (:114-118) */
    for (int i = 0; i < interval; ++i) {
        /* :116  imsize * (float)(1.0f/pow(sfactor_,(int)i)) then Size_<float> -> Size via cvRound */
        const float f = (float)((double)1.0f / pow((double)sfactor, (double)i));
        lvl_cols[i] = cv_round((double)(w * f));
        lvl_rows[i] = cv_round((double)(h * f));
        scales[i] = (float)(pow((double)sfactor, (double)i) * (double)sbin); /* :118 */
        for (int j = i + interval; j < nscales; j += interval) { /* :120-126 pyrDown */
            lvl_cols[j] = (lvl_cols[j - interval] + 1) / 2;
            lvl_rows[j] = (lvl_rows[j - interval] + 1) / 2;
            scales[j] = 2 * scales[j - interval];
        }
    }
    return nscales;
}

void pbdo_hog_dims(int rows, int cols, int sbin, int *out_rows, int *out_cols)
{ /* src/HOGFeatures.cpp:174-175 */
    int bw = (int)roundf((float)cols / (float)sbin);
    int bh = (int)roundf((float)rows / (float)sbin);
    *out_cols = bw - 2 > 0 ? bw - 2 : 0;
    *out_rows = bh - 2 > 0 ? bh - 2 : 0;
}

/* ------------------------------------------------------------------------------------------
 * cv::resize(src, dst, dsize) with INTER_LINEAR on 8-bit data (call site src/HOGFeatures.cpp:116).
 * The arithmetic is OpenCV's (imgproc/imgwarp.cpp: fixed point, INTER_RESIZE_COEF_BITS = 11),
 * third-party and absent from /root/reference; restated from SURVEY.md Appendix E.
 * ---------------------------------------------------------------------------------------- */
static inline short sat_short_round(float v)
{
    int iv = cv_round((double)v);
    return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv);
}

void pbdo_resize_linear_u8(const uint8_t *src, int srows, int scols, int cn, size_t sstride,
                           uint8_t *dst, int drows, int dcols, size_t dstride)
{
    const double inv_scale_x = (double)dcols / scols, inv_scale_y = (double)drows / srows;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dcols);
    short *alpha = (short *)malloc(sizeof(short) * 2 * (size_t)dcols);
    int *r0 = (int *)malloc(sizeof(int) * (size_t)dcols * cn);
    int *r1 = (int *)malloc(sizeof(int) * (size_t)dcols * cn);
    for (int dx = 0; dx < dcols; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= scols - 1) { fx = 0; sx = scols - 1; }
        xofs[dx] = sx;
        alpha[2 * dx + 0] = sat_short_round((1.f - fx) * 2048);
        alpha[2 * dx + 1] = sat_short_round(fx * 2048);
    }
    for (int dy = 0; dy < drows; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        const short b0 = sat_short_round((1.f - fy) * 2048), b1 = sat_short_round(fy * 2048);
        int y0 = sy, y1 = sy + 1;
        y0 = y0 >= 0 ? (y0 < srows ? y0 : srows - 1) : 0;
        y1 = y1 >= 0 ? (y1 < srows ? y1 : srows - 1) : 0;
        const uint8_t *S0 = src + (size_t)y0 * sstride, *S1 = src + (size_t)y1 * sstride;
        for (int dx = 0; dx < dcols; ++dx) {
            const int sx = xofs[dx], sx1 = sx + 1 < scols ? sx + 1 : sx;
            const int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            for (int c = 0; c < cn; ++c) {
                r0[dx * cn + c] = S0[sx * cn + c] * a0 + S0[sx1 * cn + c] * a1;
                r1[dx * cn + c] = S1[sx * cn + c] * a0 + S1[sx1 * cn + c] * a1;
            }
        }
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int i = 0; i < dcols * cn; ++i)
            D[i] = (uint8_t)((((b0 * (r0[i] >> 4)) >> 16) + ((b1 * (r1[i] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(alpha); free(r0); free(r1);
}

/* ------------------------------------------------------------------------------------------
 * cv::pyrDown on 8-bit data (call site src/HOGFeatures.cpp:122): separable [1 4 6 4 1]/16,
 * BORDER_REFLECT_101, (sum + 128) >> 8.  Third-party arithmetic, restated from SURVEY.md App. E.
 * ---------------------------------------------------------------------------------------- */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

void pbdo_pyrdown_u8(const uint8_t *src, int srows, int scols, int cn, size_t sstride,
                     uint8_t *dst, size_t dstride)
{
    const int drows = (srows + 1) / 2, dcols = (scols + 1) / 2;
    int *hrow = (int *)malloc(sizeof(int) * 5 * (size_t)dcols * cn);
    for (int y = 0; y < drows; ++y) {
        for (int k = 0; k < 5; ++k) {
            const int sy = reflect101(2 * y - 2 + k, srows);
            const uint8_t *S = src + (size_t)sy * sstride;
            int *R = hrow + (size_t)k * dcols * cn;
            for (int x = 0; x < dcols; ++x) {
                const int x0 = reflect101(2 * x - 2, scols), x1 = reflect101(2 * x - 1, scols), x2 = 2 * x,
                          x3 = reflect101(2 * x + 1, scols), x4 = reflect101(2 * x + 2, scols);
                for (int c = 0; c < cn; ++c)
                    R[x * cn + c] = S[x2 * cn + c] * 6 + (S[x1 * cn + c] + S[x3 * cn + c]) * 4 + S[x0 * cn + c] + S[x4 * cn + c];
            }
        }
        uint8_t *D = dst + (size_t)y * dstride;
        const int *R0 = hrow, *R1 = hrow + (size_t)dcols * cn, *R2 = R1 + (size_t)dcols * cn,
                  *R3 = R2 + (size_t)dcols * cn, *R4 = R3 + (size_t)dcols * cn;
        for (int i = 0; i < dcols * cn; ++i)
            D[i] = (uint8_t)((R2[i] * 6 + (R1[i] + R3[i]) * 4 + R0[i] + R4[i] + 128) >> 8);
    }
    free(hrow);
}

int pbdo_pyramid_images_u8(const uint8_t *im, int rows, int cols, int cn, size_t stride,
                           int sbin, int interval, uint8_t *out, int64_t *img_offset,
                           int *lvl_rows, int *lvl_cols, float *scales)
{
    const int n = pbdo_pyramid_plan(rows, cols, sbin, interval, lvl_rows, lvl_cols, scales);
    if (n <= 0) return n;
    int64_t off = 0;
    for (int l = 0; l < n; ++l) {
        img_offset[l] = off;
        off += (int64_t)lvl_rows[l] * lvl_cols[l] * cn;
    }
    img_offset[n] = off;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < interval; ++i) { /* src/HOGFeatures.cpp:111-127 */
        pbdo_resize_linear_u8(im, rows, cols, cn, stride, out + img_offset[i], lvl_rows[i], lvl_cols[i],
                              (size_t)lvl_cols[i] * cn);
        for (int j = i + interval; j < n; j += interval)
            pbdo_pyrdown_u8(out + img_offset[j - interval], lvl_rows[j - interval], lvl_cols[j - interval], cn,
                            (size_t)lvl_cols[j - interval] * cn, out + img_offset[j], (size_t)lvl_cols[j] * cn);
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * The other image depths HOGFeatures::pyramid accepts (src/HOGFeatures.cpp:136-146): CV_16U, CV_32F, CV_64F.
 * cv::resize / cv::pyrDown are third-party and absent from /root/reference; restated from OpenCV's generic
 * (non-SIMD) code paths -- PARITY UNPINNED, as for 8-bit:
 *   resize, INTER_LINEAR: same coordinate mapping as 8-bit; coefficients stay float (alpha = {1-fx, fx},
 *     beta = {1-fy, fy}); horizontal D = S[sx]*a0 + S[sx+1]*a1 in the work type (float for 16U / 32F, double for
 *     64F), exactly S[sx] once sx reaches the last column; vertical dst = cast(R0*b0 + R1*b1), cast =
 *     saturate_cast<ushort> (cvRound, clamp) for 16U.
 *   pyrDown: 16U integer like 8-bit ((sum + 128) >> 8); 32F / 64F the same taps in the work type,
 *     row = s2*6 + (s1 + s3)*4 + s0 + s4, dst = (r2*6 + (r1 + r3)*4 + r0 + r4) * (1/256).
 * ---------------------------------------------------------------------------------------- */
size_t pbdo_depth_size(int depth)
{
    return depth == PBDO_8U ? 1 : depth == PBDO_16U ? 2 : depth == PBDO_32F ? 4 : depth == PBDO_64F ? 8 : 0;
}

#define PBDO_RESIZE_FLT(NAME, IT, WT, CAST)                                                                       \
    static void NAME(const IT *src, int srows, int scols, int cn, size_t sstride, IT *dst, int drows, int dcols,  \
                     size_t dstride)                                                                               \
    {                                                                                                              \
        const double scale_x = 1. / ((double)dcols / scols), scale_y = 1. / ((double)drows / srows);               \
        for (int dy = 0; dy < drows; ++dy) {                                                                       \
            float fy = (float)((dy + 0.5) * scale_y - 0.5);                                                        \
            int sy = (int)floorf(fy);                                                                              \
            fy -= (float)sy;                                                                                       \
            const float b0 = 1.f - fy, b1 = fy;                                                                    \
            int y0 = sy, y1 = sy + 1;                                                                              \
            y0 = y0 >= 0 ? (y0 < srows ? y0 : srows - 1) : 0;                                                      \
            y1 = y1 >= 0 ? (y1 < srows ? y1 : srows - 1) : 0;                                                      \
            const IT *S0 = src + (size_t)y0 * sstride, *S1 = src + (size_t)y1 * sstride;                           \
            IT *D = dst + (size_t)dy * dstride;                                                                    \
            for (int dx = 0; dx < dcols; ++dx) {                                                                   \
                float fx = (float)((dx + 0.5) * scale_x - 0.5);                                                    \
                int sx = (int)floorf(fx);                                                                          \
                fx -= (float)sx;                                                                                   \
                if (sx < 0) { fx = 0; sx = 0; }                                                                    \
                const int last = sx >= scols - 1;                                                                  \
                if (last) { fx = 0; sx = scols - 1; }                                                              \
                const float a0 = 1.f - fx, a1 = fx;                                                                \
                for (int c = 0; c < cn; ++c) {                                                                     \
                    WT r0, r1;                                                                                     \
                    if (last) { r0 = (WT)S0[sx * cn + c] * (WT)1; r1 = (WT)S1[sx * cn + c] * (WT)1; }              \
                    else {                                                                                         \
                        r0 = (WT)S0[sx * cn + c] * (WT)a0 + (WT)S0[(sx + 1) * cn + c] * (WT)a1;                    \
                        r1 = (WT)S1[sx * cn + c] * (WT)a0 + (WT)S1[(sx + 1) * cn + c] * (WT)a1;                    \
                    }                                                                                              \
                    D[dx * cn + c] = CAST(r0 * (WT)b0 + r1 * (WT)b1);                                              \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
static inline uint16_t sat_u16_round(float v)
{
    long iv = lrint((double)v);
    return (uint16_t)(iv < 0 ? 0 : iv > 65535 ? 65535 : iv);
}
#define PBDO_IDENT(v) (v)
PBDO_RESIZE_FLT(resize_linear_u16, uint16_t, float, sat_u16_round)
PBDO_RESIZE_FLT(resize_linear_f32, float, float, PBDO_IDENT)
PBDO_RESIZE_FLT(resize_linear_f64, double, double, PBDO_IDENT)

#define PBDO_PYRDOWN(NAME, IT, WT, FINISH)                                                                         \
    static void NAME(const IT *src, int srows, int scols, int cn, size_t sstride, IT *dst, size_t dstride)         \
    {                                                                                                              \
        const int drows = (srows + 1) / 2, dcols = (scols + 1) / 2;                                                \
        WT *hrow = (WT *)malloc(sizeof(WT) * 5 * (size_t)dcols * cn);                                              \
        for (int y = 0; y < drows; ++y) {                                                                          \
            for (int k = 0; k < 5; ++k) {                                                                          \
                const IT *S = src + (size_t)reflect101(2 * y - 2 + k, srows) * sstride;                            \
                WT *R = hrow + (size_t)k * dcols * cn;                                                             \
                for (int x = 0; x < dcols; ++x) {                                                                  \
                    const int x0 = reflect101(2 * x - 2, scols), x1 = reflect101(2 * x - 1, scols), x2 = 2 * x,    \
                              x3 = reflect101(2 * x + 1, scols), x4 = reflect101(2 * x + 2, scols);                \
                    for (int c = 0; c < cn; ++c)                                                                   \
                        R[x * cn + c] = (WT)S[x2 * cn + c] * 6 + ((WT)S[x1 * cn + c] + (WT)S[x3 * cn + c]) * 4 +   \
                                        (WT)S[x0 * cn + c] + (WT)S[x4 * cn + c];                                   \
                }                                                                                                  \
            }                                                                                                      \
            IT *D = dst + (size_t)y * dstride;                                                                     \
            const WT *R0 = hrow, *R1 = hrow + (size_t)dcols * cn, *R2 = R1 + (size_t)dcols * cn,                   \
                     *R3 = R2 + (size_t)dcols * cn, *R4 = R3 + (size_t)dcols * cn;                                 \
            for (int i = 0; i < dcols * cn; ++i) D[i] = FINISH(R2[i] * 6 + (R1[i] + R3[i]) * 4 + R0[i] + R4[i]);   \
        }                                                                                                          \
        free(hrow);                                                                                                \
    }
#define PBDO_FIX8(v) ((uint16_t)(((v) + 128) >> 8))
#define PBDO_FLT8F(v) ((v) * (1.f / 256.f))
#define PBDO_FLT8D(v) ((v) * (1. / 256.))
PBDO_PYRDOWN(pyrdown_u16, uint16_t, int, PBDO_FIX8)
PBDO_PYRDOWN(pyrdown_f32, float, float, PBDO_FLT8F)
PBDO_PYRDOWN(pyrdown_f64, double, double, PBDO_FLT8D)

void pbdo_resize_linear(const void *src, int depth, int srows, int scols, int cn, size_t sstride, void *dst, int drows,
                        int dcols, size_t dstride)
{   /* strides in elements */
    if (depth == PBDO_8U) pbdo_resize_linear_u8((const uint8_t *)src, srows, scols, cn, sstride, (uint8_t *)dst, drows, dcols, dstride);
    else if (depth == PBDO_16U) resize_linear_u16((const uint16_t *)src, srows, scols, cn, sstride, (uint16_t *)dst, drows, dcols, dstride);
    else if (depth == PBDO_32F) resize_linear_f32((const float *)src, srows, scols, cn, sstride, (float *)dst, drows, dcols, dstride);
    else resize_linear_f64((const double *)src, srows, scols, cn, sstride, (double *)dst, drows, dcols, dstride);
}

void pbdo_pyrdown(const void *src, int depth, int srows, int scols, int cn, size_t sstride, void *dst, size_t dstride)
{
    if (depth == PBDO_8U) pbdo_pyrdown_u8((const uint8_t *)src, srows, scols, cn, sstride, (uint8_t *)dst, dstride);
    else if (depth == PBDO_16U) pyrdown_u16((const uint16_t *)src, srows, scols, cn, sstride, (uint16_t *)dst, dstride);
    else if (depth == PBDO_32F) pyrdown_f32((const float *)src, srows, scols, cn, sstride, (float *)dst, dstride);
    else pyrdown_f64((const double *)src, srows, scols, cn, sstride, (double *)dst, dstride);
}

/* out / img_offset in ELEMENTS of the image depth; stride in elements */
int pbdo_pyramid_images(const void *im, int depth, int rows, int cols, int cn, size_t stride, int sbin, int interval,
                        void *out, int64_t *img_offset, int *lvl_rows, int *lvl_cols, float *scales)
{
    const size_t es = pbdo_depth_size(depth);
    if (!es) return -3;
    const int n = pbdo_pyramid_plan(rows, cols, sbin, interval, lvl_rows, lvl_cols, scales);
    if (n <= 0) return n;
    int64_t off = 0;
    for (int l = 0; l < n; ++l) {
        img_offset[l] = off;
        off += (int64_t)lvl_rows[l] * lvl_cols[l] * cn;
    }
    img_offset[n] = off;
    char *o = (char *)out;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < interval; ++i) { /* src/HOGFeatures.cpp:111-127 */
        pbdo_resize_linear(im, depth, rows, cols, cn, stride, o + img_offset[i] * es, lvl_rows[i], lvl_cols[i],
                           (size_t)lvl_cols[i] * cn);
        for (int j = i + interval; j < n; j += interval)
            pbdo_pyrdown(o + img_offset[j - interval] * es, depth, lvl_rows[j - interval], lvl_cols[j - interval], cn,
                         (size_t)lvl_cols[j - interval] * cn, o + img_offset[j] * es, (size_t)lvl_cols[j] * cn);
    }
    return n;
}

/* back-pointer slot of (part gp, parent mixture m) = ptr_slot[gp] + m; roots own no slots */
int pbdo_ptr_slots(const pbdo_model *m, int *ptr_slot)
{
    int total = 0;
    for (int c = 0; c < m->ncomponents; ++c) {
        for (int gp = m->part_offset[c]; gp < m->part_offset[c + 1]; ++gp) {
            ptr_slot[gp] = total;
            if (m->parentid[gp] >= 0) {
                const int gpar = m->part_offset[c] + m->parentid[gp];
                total += m->mix_offset[gpar + 1] - m->mix_offset[gpar];
            }
        }
    }
    return total;
}

/* ---- T = float ---- */
#define REAL float
#define FN(x) x##_f32
#define SQRT_REAL sqrtf
#define RINT_REAL rintf
static const float *model_filters_f32(const pbdo_model *m) { return m->filters_f32; }
#include "pbd_oracle_impl.inc"
#undef REAL
#undef FN
#undef SQRT_REAL
#undef RINT_REAL

/* ---- T = double ---- */
#define REAL double
#define FN(x) x##_f64
#define SQRT_REAL sqrt
#define RINT_REAL rint
static const double *model_filters_f64(const pbdo_model *m) { return m->filters_f64; }
#include "pbd_oracle_impl.inc"
#undef REAL
#undef FN
#undef SQRT_REAL
#undef RINT_REAL
