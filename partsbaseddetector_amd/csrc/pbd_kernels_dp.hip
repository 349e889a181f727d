// pbd_kernels_dp.hip -- distance transform, dynamic-program message passing and back-tracking (gfx950).
//
// Replaces DynamicProgram<T>::min / argmin (reference src/DynamicProgram.cpp:67-255),
// DistanceTransform<T>::computeRow / compute (include/DistanceTransform.hpp:152-245) and
// Math::reduceMax / reducePickIndex / find (include/Math.hpp:84-185) for T=float.
//
// The envelope algorithm is run literally (the double-precision intersection rounded once to float,
// the `s <= z[k] && k > 0` pop rule, the float `z[k+1] < os` read-out), one thread per row (then per
// column): a brute-force max would pick different arg-max pointers on near ties.  The parts tree is
// processed one depth at a time; a part's input is its raw response plus its children's messages
// added in descending child order, which is the order the reference accumulates them in
// (src/DynamicProgram.cpp:95,154-156).  Compiled with -ffp-contract=off.
#include "pbd_internal.h"

#include <math.h>

namespace pbd {

// The intersection for T = float without the fp64 divide (TAB > 0: entries of the 1/d table in LDS).
// The reference value is RN_float(RN_double(num / den)), den = (2a) * d exactly (a is a float widened to double,
// d < TAB <= 512).  qt = num * (RN(1/(2a)) * RN(1/d)) differs from RN_double(num / den) by at most 5 double ulps
// (three roundings of 2^-53 in the reciprocal, one in the product, half an ulp in RN_double), so (float)qt is the
// reference value unless qt lies within 8 ulps of a midpoint between two floats -- bits 0..28 of the double
// significand within 8 of 2^28 -- or (float)qt is not a normal float (zero, subnormal, infinite, NaN: the float grid
// is different there, or the operands were not finite).  Those lanes (about one in 2^24) take the divide.
// BZERO: the linear coefficient b is exactly -0.0 (deformation weight +0.0f, the usual case): (y1 - y0) - b*d and
// a*x*x + b*x then equal their first terms for every value that can reach the result (x - (-0.0) = x except for
// x = -0.0, and the a-term added next is never zero in the intersection; t + (-0.0) = t for every t in the value).
struct IsectCtx {
    double a, b, inv2a, den2a;
    const double *invd;           // LDS: invd[d] = RN(1.0 / d)
};
template <typename R, bool BZERO, int TAB>
__device__ __forceinline__ R ctx_isect(const IsectCtx &c, int x0, int x1, R y0f, R y1f)
{
    if constexpr (sizeof(R) == 4 && TAB > 0) {
        const int d = x1 - x0;
        const double dd = (double)d;
        double num = (double)y1f - (double)y0f;
        if (!BZERO) num = num - c.b * dd;
        num = num + c.a * (double)(d * (x1 + x0));      // x1*x1 - x0*x0 in int, as the reference
        const double qt = num * (c.inv2a * c.invd[d]);
        const unsigned lo = (unsigned)__double2loint(qt);
        float s = (float)qt;
        const bool near_mid = ((lo & 0x1FFFFFFFu) - (0x10000000u - 8u)) < 17u;
        const bool odd_class = !__builtin_isnormal(s);
        if (__builtin_expect(near_mid || odd_class, 0)) s = (float)(num / (c.den2a * dd));
        return s;
    } else {
        const double y0 = (double)y0f, y1 = (double)y1f;
        const double num = ((y1 - y0) - c.b * (double)(x1 - x0)) + c.a * (double)(x1 * x1 - x0 * x0);
        return (R)(num / ((2 * c.a) * (double)(x1 - x0)));
    }
}
template <typename R, bool BZERO>
__device__ __forceinline__ R ctx_val(const IsectCtx &c, int x, R y)
{   // Quadratic::operator()(x, y), :103-105
    if (BZERO) return (R)(c.a * (double)(x * x) + (double)y);
    return (R)((c.a * (double)(x * x) + c.b * (double)x) + (double)y);
}
template <typename R> struct RealLimits;
template <> struct RealLimits<float> { static __device__ __forceinline__ float inf() { return INFINITY; } };
template <> struct RealLimits<double> { static __device__ __forceinline__ double inf() { return (double)INFINITY; } };

// ------------------------------------------------------------------------------------------------
// One 1-D transform per thread, streamed in chunks of CH elements.
//   * the envelope's top entry (v[k], z[k], src[v[k]]) lives in registers;
//   * the T entries below it live in a per-lane LDS ring (slot = index % T, arrays [T][64] so that a
//     lane always hits bank lane % 32: conflict-free whatever the lanes' indices are); a pop is three
//     ds_read_b32, a push three ds_write_b32;
//   * entries that fall out of the ring are NOT stored anywhere (round 1 spilled them to a global stack: two thirds
//     of the passes' HBM traffic).  The stack is ordered by position, so all that has to be remembered is WHICH
//     positions they are: one bit per position in LDS ([word][64], the top non-empty word cached in a register).
//     When a pop reaches below the ring, the entry comes back as: position = highest set bit; src = re-read from the
//     lane's own source row (a 16-byte chunk of four consecutive positions is kept, survivors are mostly
//     adjacent); z = its intersection with the next survivor below, recomputed with the same expression on the same
//     operands as when it was pushed (bit-identical);
//   * the read-out walks q downwards and POPS: since z[1..ktop] is strictly increasing,
//     "k = 0; while (z[k+1] < os) k++" (DistanceTransform.hpp:172-178, q ascending) selects the same
//     k(q) = max{k : z[k] < os(q)} as "k = ktop; while (!(z[k] < os)) k--" with q descending;
//   * source values are prefetched one chunk ahead and results leave in whole chunks.
// The arithmetic per element is exactly computeRow's (DistanceTransform.hpp:152-182).
// ------------------------------------------------------------------------------------------------
#ifndef PBD_DT_CH
#define PBD_DT_CH 16
#endif
constexpr int kDtCH = PBD_DT_CH;   // elements per streamed chunk (multiple of 8)
constexpr int kDtWaves = 1;        // waves per workgroup of the DT passes (each wave = 64 rows / columns)
#ifndef PBD_DT_RING
#define PBD_DT_RING 8
#endif
constexpr int kDtT = PBD_DT_RING;    // ring entries per lane (power of two)

typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
typedef double v2d_u __attribute__((ext_vector_type(2), aligned(8)));
typedef short v8s_u __attribute__((ext_vector_type(8), aligned(2)));

// LDS layout of a wave's ring: [slot][z: 64 x R | s: 64 x R | v: 64 x int]; one address per lane, the rest
// are immediate offsets.  NW = words of the position mask (rows / columns shorter than 32 * NW).
// The mask has one bit per position that is on the stack (the top included); its word holding the top is cached in
// a register, so a push or a pop inside that word touches no memory.  LDS words below the cached one are exact, the
// ones above it are dead (rewritten before they are read again).
template <typename R, bool BZERO, int TAB, int NW>
struct DtRing {
    static constexpr int kSlotBytes = 64 * (2 * (int)sizeof(R) + 4);
    static constexpr int kLdsBytes = kDtT * kSlotBytes + NW * 256;
    static constexpr int kChunk = 16 / (int)sizeof(R);      // source values per re-read chunk
    char *zs;                     // this lane's z of slot 0 (s is 64 R further)
    char *vp;                     // this lane's v of slot 0
    unsigned *mask;               // this lane's column of the [NW][64] position mask
    const R *src;                 // this lane's source row / column in global memory (contiguous)
    IsectCtx c;                   // the job's quadratic
    int lo;                       // ring holds z, s, v of the stack indices [lo, top)
    unsigned cw; int cwi;         // cached mask word and its index = (position of the top) >> 5
    R ch[kChunk]; int cb;         // src[cb .. cb + kChunk), cb a multiple of kChunk (-1: nothing cached)
    __device__ __forceinline__ R &z(int slot) { return *reinterpret_cast<R *>(zs + slot * kSlotBytes); }
    __device__ __forceinline__ R &s(int slot) { return *reinterpret_cast<R *>(zs + slot * kSlotBytes + 64 * (int)sizeof(R)); }
    __device__ __forceinline__ int &v(int slot) { return *reinterpret_cast<int *>(vp + slot * kSlotBytes); }
    __device__ __forceinline__ void push_below(int idx, R zk, R sk, int vk, int q)
    {   // entry `idx` (the old top, position vk) moves under a new top at position q
        const int slot = idx & (kDtT - 1);
        if (idx - lo >= kDtT) lo += 1;          // the ring's oldest entry is overwritten: from now on it lives in the mask only
        z(slot) = zk; s(slot) = sk; v(slot) = vk;
        const int w = q >> 5;
        if (w != cwi) {
            mask[cwi * 64] = cw;
            for (int i = cwi + 1; i < w; ++i) mask[i * 64] = 0;   // words skipped after a deep pop
            cwi = w; cw = 0;
        }
        cw |= 1u << (q & 31);
    }
    __device__ __forceinline__ R fetch(int pos)
    {   // src[pos] through the one-chunk cache (reads up to kChunk - 1 values past the row: inside the buffers' slack)
        const int base = pos & ~(kChunk - 1);
        if (base != cb) {
            cb = base;
            if constexpr (sizeof(R) == 4) {
                const v4f_u t = *reinterpret_cast<const v4f_u *>(src + base);
                ch[0] = t.x; ch[1] = t.y; ch[2] = t.z; ch[3] = t.w;
            } else {
                const v2d_u t = *reinterpret_cast<const v2d_u *>(src + base);
                ch[0] = t.x; ch[1] = t.y;
            }
        }
        R r = ch[0];
#pragma unroll
        for (int e = 1; e < kChunk; ++e) r = ((pos & (kChunk - 1)) == e) ? ch[e] : r;
        return r;
    }
    __device__ __forceinline__ void pop(int idx, R &zk, R &sk, int &vk)
    {   // the top (position vk) leaves the stack, entry `idx` becomes the top
        cw &= ~(1u << (vk & 31));
        // the ring slot is read unconditionally (always a valid LDS address) so that the common case is
        // plain LDS reads; only a pop below the ring overrides it
        const int slot = idx & (kDtT - 1);
        zk = z(slot); sk = s(slot); vk = v(slot);
        asm volatile("" : "+v"(zk), "+v"(sk), "+v"(vk));   // keep these as LDS reads (not a flat load of a selected pointer)
        if (idx < lo) {
            // below the ring: the entry is the highest position left in the mask; its source value is re-read and its
            // z recomputed as the intersection with the next position below -- as when it was pushed
            while (cw == 0) { mask[cwi * 64] = 0; --cwi; cw = mask[cwi * 64]; }
            vk = (cwi << 5) + 31 - __clz((int)cw);
            sk = fetch(vk);
            if (vk == 0) {
                zk = -RealLimits<R>::inf();                  // z[0], DistanceTransform.hpp:157
            } else {
                unsigned pw = cw & ~(1u << (vk & 31));
                int pi = cwi;
                while (pw == 0) { --pi; pw = mask[pi * 64]; }
                const int vb = (pi << 5) + 31 - __clz((int)pw);
                const R sb = fetch(vb);
                zk = ctx_isect<R, BZERO, TAB>(c, vb, vk, sb, sk);
            }
            lo = idx;
        } else if ((vk >> 5) != cwi) {
            mask[cwi * 64] = cw;
            cwi = vk >> 5;
            cw = mask[cwi * 64];
        }
    }
    static __device__ __forceinline__ DtRing make(char *smem, int lane, const R *src, double a, double b, const double *invd)
    {
        DtRing r;
        r.zs = smem + lane * (int)sizeof(R);
        r.vp = smem + 128 * (int)sizeof(R) + lane * 4;
        r.mask = reinterpret_cast<unsigned *>(smem + kDtT * kSlotBytes) + lane;
        r.src = src;
        r.c = IsectCtx{a, b, 1.0 / (2 * a), 2 * a, invd};
        r.lo = 0; r.cw = 1u; r.cwi = 0; r.cb = -1;       // position 0 is on the stack from the start
#pragma unroll
        for (int e = 0; e < kChunk; ++e) r.ch[e] = (R)0;
        return r;
    }
    // invd[d] = RN(1.0 / d) for d < TAB, filled by the workgroup (followed by a barrier in the kernel)
    static __device__ __forceinline__ void fill_invd(double *invd)
    {
        if constexpr (sizeof(R) == 4 && TAB > 0)
            for (int d = threadIdx.x; d < TAB; d += blockDim.x) invd[d] = 1.0 / (double)max(d, 1);
    }
};

template <typename R, class Ring, bool BZERO, int TAB, class LoadChunk, class StoreChunk>
__device__ __forceinline__ void dt_stream(int N, int os0, Ring ring, LoadChunk load, StoreChunk store)
{
    const IsectCtx c = ring.c;
    constexpr int CH = kDtCH;
    R cur[CH], nxt[CH];
    load(0, cur);
    int k = 0, vk = 0;
    R zk = -RealLimits<R>::inf(), sk = cur[0];
    for (int q0 = 0; q0 < N; q0 += CH) {
        if (q0 + CH < N) load(q0 + CH, nxt);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int q = q0 + i;
            if (q >= 1 && q < N) {
                const R sq = cur[i];
                R s = ctx_isect<R, BZERO, TAB>(c, vk, q, sk, sq);
                while (s <= zk && k > 0) {
                    --k;
                    ring.pop(k, zk, sk, vk);
                    s = ctx_isect<R, BZERO, TAB>(c, vk, q, sk, sq);
                }
                ring.push_below(k, zk, sk, vk, q);
                ++k;
                vk = q; zk = s; sk = sq;
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) cur[i] = nxt[i];
    }
    // read-out, q descending
    const int nch = (N + CH - 1) / CH;
    for (int cidx = nch - 1; cidx >= 0; --cidx) {
        const int q0 = cidx * CH;
        R out[CH];
        int ptr[CH];
#pragma unroll
        for (int i = CH - 1; i >= 0; --i) {
            out[i] = (R)0; ptr[i] = 0;
            const int q = q0 + i;
            if (q < N) {
                const R osf = (R)(os0 + q);
                while (!(zk < osf)) {   // z[0] = -inf ends the walk
                    --k;
                    ring.pop(k, zk, sk, vk);
                }
                out[i] = ctx_val<R, BZERO>(c, os0 + q - vk, sk);
                ptr[i] = vk;
            }
        }
        store(q0, out, ptr);
    }
}

static_assert(kDtCH % 8 == 0, "the rows pass stores its int16 pointers 8 at a time");

template <typename R>
__device__ __forceinline__ void load_chunk(const R *src, int q0, int N, R *buf)
{
    if (sizeof(R) == 4 && q0 + kDtCH <= N) {
        const float *srcf = reinterpret_cast<const float *>(src);
#pragma unroll
        for (int v = 0; v < kDtCH / 4; ++v) {
            const v4f_u a0 = *reinterpret_cast<const v4f_u *>(srcf + q0 + 4 * v);
            buf[4 * v] = a0.x; buf[4 * v + 1] = a0.y; buf[4 * v + 2] = a0.z; buf[4 * v + 3] = a0.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < kDtCH; ++i) buf[i] = (q0 + i < N) ? src[q0 + i] : (R)0;
    }
}

// ---- rows pass: thread = (flat row, job, frame); each lane streams its own row with 16-byte accesses ----
// out: tmp TRANSPOSED ([x][y]: lanes are adjacent rows, every store instruction writes whole lines; the columns pass
// reads its own column back contiguously) and the pointers Ix ROW-MAJOR ([y][x], 16 int16 per lane and store), which
// is what the combine step reads: the reference's final Ix is the rows pass's pointer unchanged
// (include/DistanceTransform.hpp:233-244).
template <typename R, bool BZERO, int TAB, int NW>
__global__ __launch_bounds__(64 * kDtWaves) void k_dt_rows(DpParams p)
{
    typedef DtRing<R, BZERO, TAB, NW> Ring;
    __shared__ double invd[TAB > 0 ? TAB : 1];
    __shared__ __attribute__((aligned(16))) char ring_mem[Ring::kLdsBytes];
    Ring::fill_invd(invd);
    // grid = (job, frame, wave of 64 flat rows): the wave index is the SLOWEST dimension, so the long rows of
    // the large levels are dispatched first and the tail of the launch is made of short ones
    const int wv = blockIdx.z, lane = threadIdx.x;
    const int r = wv * 64 + lane;
    const bool active = r < p.nrows_flat;
    const int rr = active ? r : p.nrows_flat - 1;
    const int j = blockIdx.x, fl = blockIdx.y, frame = p.frame0 + fl;
    const int l = p.row2level[rr];
    const LevelDesc d = p.lv[l];
    const int y = rr - p.rowoff[l];
    const int W = d.cols;
    const size_t HW = (size_t)d.rows * W;
    const DtJob job = p.jobs[j];
    const R *src = (job.from_acc ? static_cast<const R *>(p.acc) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NM
                                 : static_cast<const R *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F) +
                   (size_t)job.plane * HW + (size_t)y * W;
    const int Hl = d.rows;
    const size_t jbase = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG + (size_t)j * HW;
    R *tmpT = static_cast<R *>(p.tmp) + jbase + (size_t)y;
    int16_t *ixr = p.IxRaw + jbase + (size_t)y * W;
    __syncthreads();
    Ring ring = Ring::make(ring_mem, lane, src, job.ax, job.bx, invd);
    const int N = active ? W : 0;
    if (N == 0) return;
    auto load = [&](int q0, R *buf) { load_chunk<R>(src, q0, N, buf); };
    auto store = [&](int q0, const R *out, const int *ptr) {
#ifndef PBD_EXP_NO_TMP
#pragma unroll
        for (int i = 0; i < kDtCH; ++i)
            if (q0 + i < N) tmpT[(size_t)(q0 + i) * Hl] = out[i];
#else
        if (out[0] == (R)12345.678) tmpT[0] = out[1];
#endif
#ifdef PBD_EXP_NO_IX
        if (ptr[0] == -77) ixr[0] = (int16_t)ptr[1];
        return;
#endif
        if (q0 + kDtCH <= N) {
#pragma unroll
            for (int v = 0; v < kDtCH / 8; ++v) {
                v8s_u t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (short)ptr[8 * v + e];
                *reinterpret_cast<v8s_u *>(ixr + q0 + 8 * v) = t;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kDtCH; ++i)
                if (q0 + i < N) ixr[q0 + i] = (int16_t)ptr[i];
        }
    };
    dt_stream<R, Ring, BZERO, TAB>(N, job.osx, ring, load, store);
}

// tab: entries of the 1/d table of the divide-free float intersection (0: divide); nw: words of the position mask
// (both: the longest row / column must be shorter than 256 resp. 512); bzero: every job's linear coefficient in this
// direction is exactly -0.0
#define PBD_DT_LAUNCH(K)                                                                                        \
    do {                                                                                                        \
        if (f64) {                                                                                              \
            if (wide) hipLaunchKernelGGL((K<double, false, 0, 16>), grid, dim3(64), 0, s, p);                   \
            else hipLaunchKernelGGL((K<double, false, 0, 8>), grid, dim3(64), 0, s, p);                         \
        } else if (nodiv && !wide) {                                                                            \
            if (bzero) hipLaunchKernelGGL((K<float, true, 256, 8>), grid, dim3(64), 0, s, p);                   \
            else hipLaunchKernelGGL((K<float, false, 256, 8>), grid, dim3(64), 0, s, p);                        \
        } else if (nodiv) {                                                                                     \
            if (bzero) hipLaunchKernelGGL((K<float, true, 512, 16>), grid, dim3(64), 0, s, p);                  \
            else hipLaunchKernelGGL((K<float, false, 512, 16>), grid, dim3(64), 0, s, p);                       \
        } else {                                                                                                \
            if (wide) hipLaunchKernelGGL((K<float, false, 0, 16>), grid, dim3(64), 0, s, p);                    \
            else hipLaunchKernelGGL((K<float, false, 0, 8>), grid, dim3(64), 0, s, p);                          \
        }                                                                                                       \
    } while (0)

void launch_dt_rows(const DpParams &p, int nframes, bool f64, bool bzero, bool nodiv, bool wide, hipStream_t s)
{
    if (p.JG == 0 || p.nrows_flat == 0) return;
    dim3 grid(p.JG, nframes, (p.nrows_flat + 63) / 64);
    PBD_DT_LAUNCH(k_dt_rows);
}

// ---- columns pass: thread = (flat column, job, frame); lanes are adjacent columns -> coalesced outputs ----
template <typename R, bool BZERO, int TAB, int NW>
__global__ __launch_bounds__(64 * kDtWaves) void k_dt_cols(DpParams p)
{
    typedef DtRing<R, BZERO, TAB, NW> Ring;
    __shared__ double invd[TAB > 0 ? TAB : 1];
    __shared__ __attribute__((aligned(16))) char ring_mem[Ring::kLdsBytes];
    Ring::fill_invd(invd);
    const int wv = blockIdx.z, lane = threadIdx.x;   // longest columns first, as in the rows pass
    const int cidx = wv * 64 + lane;
    const bool active = cidx < p.ncols_flat;
    const int cc = active ? cidx : p.ncols_flat - 1;
    const int j = blockIdx.x, fl = blockIdx.y;
    const int l = p.col2level[cc];
    const LevelDesc d = p.lv[l];
    const int x = cc - p.coloff[l];
    const int H = active ? d.rows : 0, W = d.cols;
    const size_t HW = (size_t)d.rows * W;
    const DtJob job = p.jobs[j];
    const size_t jbase = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG + (size_t)j * HW;
    const R *tmpT = static_cast<const R *>(p.tmp) + jbase + (size_t)x * d.rows;     // this lane's column, contiguous
    R *dt = static_cast<R *>(p.dt) + jbase + x;
    int16_t *iyr = p.IyRaw + jbase + x;
    __syncthreads();
    Ring ring = Ring::make(ring_mem, lane, tmpT, job.ay, job.by, invd);
    if (H == 0) return;
    auto load = [&](int q0, R *buf) { load_chunk<R>(tmpT, q0, H, buf); };
    auto store = [&](int q0, const R *out, const int *ptr) {
#pragma unroll
        for (int i = 0; i < kDtCH; ++i)
            if (q0 + i < H) {
                dt[(size_t)(q0 + i) * W] = out[i];
                iyr[(size_t)(q0 + i) * W] = (int16_t)ptr[i];
            }
    };
    dt_stream<R, Ring, BZERO, TAB>(H, job.osy, ring, load, store);
}

void launch_dt_cols(const DpParams &p, int nframes, bool f64, bool bzero, bool nodiv, bool wide, hipStream_t s)
{
    if (p.JG == 0 || p.ncols_flat == 0) return;
    dim3 grid(p.JG, nframes, (p.ncols_flat + 63) / 64);
    PBD_DT_LAUNCH(k_dt_cols);
}
#undef PBD_DT_LAUNCH

// ---- combine: thread = 4 consecutive cells of one level, one PARENT part (block.y) ---------------------
// For every parent mixture m: acc = response(parent, m); then for each child in descending index order
//   weighted[mm] = score_dt[child][mm] + bias(mm)[m]; reduceMax (strict >, first wins, start -inf; K==1 copies);
//   Ix/Iy picked from the winning mixture with the reference's composition Iy[y][x] = IyRaw[y][Ix[y][x]]
//   (include/DistanceTransform.hpp:233-244); acc += max   (src/DynamicProgram.cpp:134-156).
// The accumulated plane is the input of the parent's own distance transform in the next group.
// Four cells per thread: every plane is read and written with 16 / 8 / 4-byte accesses per lane.
template <typename T, int N> struct CellVec;
#define PBD_CELLVEC(T, E, N) template <> struct CellVec<T, N> { typedef E type __attribute__((ext_vector_type(N), aligned(sizeof(E)))); }
PBD_CELLVEC(float, float, 4); PBD_CELLVEC(float, float, 2); PBD_CELLVEC(double, double, 4); PBD_CELLVEC(double, double, 2);
PBD_CELLVEC(int16_t, short, 4); PBD_CELLVEC(int16_t, short, 2); PBD_CELLVEC(uint8_t, unsigned char, 4); PBD_CELLVEC(uint8_t, unsigned char, 2);
#undef PBD_CELLVEC

template <typename T, int kCpt>
__device__ __forceinline__ void load_cells(const T *src, int n, T *dst)
{   // n valid cells (1..kCpt); unaligned wide access is fine in global memory
    if (n == kCpt) {
        const typename CellVec<T, kCpt>::type v = *reinterpret_cast<const typename CellVec<T, kCpt>::type *>(src);
#pragma unroll
        for (int e = 0; e < kCpt; ++e) dst[e] = (T)v[e];
    } else {
#pragma unroll
        for (int e = 0; e < kCpt; ++e) dst[e] = (e < n) ? src[e] : (T)0;
    }
}
template <typename T, int kCpt>
__device__ __forceinline__ void store_cells(T *dst, int n, const T *src)
{
    if (n == kCpt) {
        typename CellVec<T, kCpt>::type v;
#pragma unroll
        for (int e = 0; e < kCpt; ++e) v[e] = src[e];
        *reinterpret_cast<typename CellVec<T, kCpt>::type *>(dst) = v;
    } else {
#pragma unroll
        for (int e = 0; e < kCpt; ++e) if (e < n) dst[e] = src[e];
    }
}

// MAXM: compile-time bound on the mixtures per part of the model (register arrays are sized by it)
template <typename R, int kCpt, int MAXM>
__global__ __launch_bounds__(256) void k_dp_combine(DpParams p)
{
    constexpr int SUB = 4 / kCpt;   // threads per group of 4 cells
    const long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long qidx = gidx / SUB;
    if (qidx >= p.quad_per_frame) return;
    const int fl = blockIdx.z, frame = p.frame0 + fl;
    const CombineJob cj = p.cjobs[blockIdx.y];
    int lo = 0, hi = p.nlevels;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (p.lv[mid].quad_off <= qidx) lo = mid; else hi = mid; }
    const LevelDesc d = p.lv[lo];
    const int W = d.cols;
    const int HWi = d.rows * W;
    const size_t HW = (size_t)HWi;
    const int local = (int)(qidx - d.quad_off) * 4 + (int)(gidx % SUB) * kCpt;
    if (local >= HWi) return;
    const int n = min(kCpt, HWi - local);
    int rowbase[kCpt];       // y * W of each cell
#pragma unroll
    for (int e = 0; e < kCpt; ++e) rowbase[e] = ((local + e) / W) * W;
    const R *resp = static_cast<const R *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + local;
    const R *dtp = static_cast<const R *>(p.dt);
    const size_t gbase0 = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG;
    const size_t pbase = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NS + local;
    R accv[MAXM][kCpt];
#pragma unroll
    for (int pm = 0; pm < MAXM; ++pm) {
#pragma unroll
        for (int e = 0; e < kCpt; ++e) accv[pm][e] = (R)0;
        if (pm < cj.npar) load_cells<R, kCpt>(resp + (size_t)cj.filter[pm] * HW, n, accv[pm]);
    }
    // the job tables and biases are read-only and wave-uniform: reading them through the constant address
    // space keeps them on the scalar unit (a plain global load after the first store would be a vector load
    // the compiler has to wait for at every use)
    const int __attribute__((address_space(4))) *childs = (const int __attribute__((address_space(4))) *)p.childs;
    static_assert(sizeof(ChildDesc) == 11 * sizeof(int), "ChildDesc is read as 11 ints");
    const float __attribute__((address_space(4))) *biasw = (const float __attribute__((address_space(4))) *)p.biasw;
    for (int ch = cj.child_begin; ch < cj.child_end; ++ch) {
        ChildDesc cd;
        {
            const int __attribute__((address_space(4))) *ci = childs + (size_t)ch * 11;
            cd.job_begin = ci[0]; cd.nmix = ci[1]; cd.slot = ci[2];
#pragma unroll
            for (int k = 0; k < 8; ++k) cd.bias_off[k] = ci[3 + k];
        }
        const size_t gbase = gbase0 + (size_t)cd.job_begin * HW;
        float bw[MAXM][MAXM]; // bias(mm)[pm]
#pragma unroll
        for (int mm = 0; mm < MAXM; ++mm)
#pragma unroll
            for (int pm = 0; pm < MAXM; ++pm)
                bw[mm][pm] = (mm < cd.nmix && pm < cj.npar) ? biasw[cd.bias_off[mm] + pm] : 0.0f;
        R dtv[MAXM][kCpt];
        int16_t ixv[MAXM][kCpt];
#pragma unroll
        for (int mm = 0; mm < MAXM; ++mm) {
#pragma unroll
            for (int e = 0; e < kCpt; ++e) { dtv[mm][e] = (R)0; ixv[mm][e] = 0; }
            if (mm < cd.nmix) {
                load_cells<R, kCpt>(dtp + gbase + (size_t)mm * HW + local, n, dtv[mm]);
                load_cells<int16_t, kCpt>(p.IxRaw + gbase + (size_t)mm * HW + local, n, ixv[mm]);
            }
        }
#pragma unroll
        for (int pm = 0; pm < MAXM; ++pm) {
            if (pm < cj.npar) {
                int16_t oix[kCpt], oiy[kCpt];
                uint8_t oik[kCpt];
#pragma unroll
                for (int e = 0; e < kCpt; ++e) {
                    R best;
                    int bi = 0, ix = ixv[0][e];
                    if (cd.nmix == 1) {
                        best = dtv[0][e] + (R)bw[0][pm];
                    } else {
                        best = -RealLimits<R>::inf();
#pragma unroll
                        for (int mm = 0; mm < MAXM; ++mm) {
                            if (mm < cd.nmix) {
                                const R wv = dtv[mm][e] + (R)bw[mm][pm];
                                if (wv > best) { bi = mm; best = wv; ix = ixv[mm][e]; }
                            }
                        }
                    }
                    int iy = 0;
                    if (e < n) iy = p.IyRaw[gbase + (size_t)bi * HW + rowbase[e] + ix];
                    oix[e] = (int16_t)ix; oiy[e] = (int16_t)iy; oik[e] = (uint8_t)bi;
                    accv[pm][e] = accv[pm][e] + best;
                }
                const size_t o = pbase + (size_t)(cd.slot + pm) * HW;
                store_cells<int16_t, kCpt>(p.Ix + o, n, oix);
                store_cells<int16_t, kCpt>(p.Iy + o, n, oiy);
                store_cells<uint8_t, kCpt>(p.Ik + o, n, oik);
            }
        }
    }
    R *acc = static_cast<R *>(p.acc) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NM + local;
#pragma unroll
    for (int pm = 0; pm < MAXM; ++pm)
        if (pm < cj.npar) store_cells<R, kCpt>(acc + (size_t)(cj.acc_plane + pm) * HW, n, accv[pm]);
}

void launch_dp_combine(const DpParams &p, int ncjobs, int nframes, bool f64, hipStream_t s)
{
    if (ncjobs == 0 || p.quad_per_frame == 0) return;
    const dim3 g2((unsigned)((p.quad_per_frame * 2 + 255) / 256), ncjobs, nframes), g4((unsigned)((p.quad_per_frame + 255) / 256), ncjobs, nframes);
#define PBD_COMBINE(M)                                                                      \
    do {                                                                                    \
        if (f64) hipLaunchKernelGGL((k_dp_combine<double, 2, M>), g2, dim3(256), 0, s, p);   \
        else hipLaunchKernelGGL((k_dp_combine<float, 4, M>), g4, dim3(256), 0, s, p);        \
    } while (0)
    if (p.max_mix <= 2) PBD_COMBINE(2);
    else if (p.max_mix <= 4) PBD_COMBINE(4);
    else if (p.max_mix <= 6) PBD_COMBINE(6);
    else PBD_COMBINE(8);
#undef PBD_COMBINE
}

// ---- root: rootv = max over root mixtures of (accumulated score + bias) ----------------------------
template <typename R>
__global__ __launch_bounds__(256) void k_dp_root(DpParams p)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.cell_per_frame) return;
    const int c = blockIdx.y, fl = blockIdx.z, frame = p.frame0 + fl;
    const RootJob rj = p.rjobs[c];
    int lo = 0, hi = p.nlevels;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (p.lv[mid].cell_off <= idx) lo = mid; else hi = mid; }
    const LevelDesc d = p.lv[lo];
    const int local = (int)(idx - d.cell_off);
    const size_t HW = (size_t)d.rows * d.cols;
    const R *src = rj.from_acc ? static_cast<const R *>(p.acc) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NM
                               : static_cast<const R *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F;
    R best;
    int bi = 0;
    if (rj.nmix == 1) {
        best = src[(size_t)rj.plane[0] * HW + local] + (R)rj.bias;
    } else {
        best = -RealLimits<R>::inf();
        for (int mm = 0; mm < rj.nmix; ++mm) {
            const R wv = src[(size_t)rj.plane[mm] * HW + local] + (R)rj.bias;
            if (wv > best) { bi = mm; best = wv; }
        }
    }
    const size_t o = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NC + (size_t)c * HW + local;
    static_cast<R *>(p.rootv)[o] = best;
    p.rooti[o] = bi;
}

void launch_dp_root(const DpParams &p, int nframes, bool f64, hipStream_t s)
{
    if (p.cell_per_frame == 0) return;
    dim3 grid((unsigned)((p.cell_per_frame + 255) / 256), p.NC, nframes);
    if (f64) hipLaunchKernelGGL(k_dp_root<double>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_dp_root<float>, grid, dim3(256), 0, s, p);
}

// ---- argmin ------------------------------------------------------------------------------------
// find: rootv > thresh (strict, src/DynamicProgram.cpp:208) -> append (frame, component, level, x, y, score, mix)
template <typename R>
__global__ __launch_bounds__(256) void k_argmin_find(ArgminParams p)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.cell_per_frame) return;
    const int c = blockIdx.y, frame = blockIdx.z;
    int lo = 0, hi = p.nlevels;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (p.lv[mid].cell_off <= idx) lo = mid; else hi = mid; }
    const LevelDesc d = p.lv[lo];
    const int local = (int)(idx - d.cell_off);
    const size_t HW = (size_t)d.rows * d.cols;
    const size_t o = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NC + (size_t)c * HW + local;
    const R v = static_cast<const R *>(p.rootv)[o];
    if (!(v > (R)p.thresh)) return;
    const int slot = atomicAdd(p.count, 1);
    if (slot >= p.capacity) return;
    int32_t *rec = p.cand + (size_t)slot * p.stride;
    rec[0] = frame; rec[1] = c; rec[2] = lo;
    rec[3] = local % d.cols; rec[4] = local / d.cols;
    rec[5] = __float_as_int((float)v);   // Candidate::confidence_ is float for every T (include/Candidate.hpp:72)
    rec[6] = 0;
    rec[7] = p.rooti[o];   // root mixture, consumed by the walk kernel
}

void launch_argmin_find(const ArgminParams &p, bool f64, hipStream_t s)
{
    if (p.cell_per_frame == 0) return;
    dim3 grid((unsigned)((p.cell_per_frame + 255) / 256), p.NC, p.nframes);
    if (f64) hipLaunchKernelGGL(k_argmin_find<double>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_argmin_find<float>, grid, dim3(256), 0, s, p);
}

template <typename R> __device__ __forceinline__ int round_mul(int a, R s);
// cv::Point_<int> * T -> saturate_cast<int>(a*s) = cvRound: round half to even
template <> __device__ __forceinline__ int round_mul<float>(int a, float s) { return __float2int_rn((float)a * s); }
template <> __device__ __forceinline__ int round_mul<double>(int a, double s) { return __double2int_rn((double)a * s); }

// walk: one thread per candidate follows Ix/Iy/Ik from the root (src/DynamicProgram.cpp:218-244)
template <typename R>
__global__ __launch_bounds__(64) void k_argmin_walk(ArgminParams p, int ncand)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncand) return;
    int32_t *rec = p.cand + (size_t)i * p.stride;
    const int frame = rec[0], c = rec[1], l = rec[2];
    const LevelDesc d = p.lv[l];
    const int W = d.cols;
    const size_t HW = (size_t)d.rows * W;
    const R scale = (R)p.scales[l];   // T scale = scales[n] (vectorf), src/DynamicProgram.cpp:199
    const PartWalk *walk = p.walk + p.walk_off[c];
    const int nparts = p.walk_off[c + 1] - p.walk_off[c];
    const size_t pbase = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NS;
    int32_t *rects = rec + 8;
    // xv/yv/mv of already visited parts are kept in the record itself: x,y in the rect slots
    // (overwritten by the final rect once all children are done is not possible in one pass), so use
    // a small per-thread walk: parents precede children, keep coordinates in local arrays.
    int xv[80], yv[80], mv[80];
    for (int pidx = 0; pidx < nparts; ++pidx) {
        int x, y, m;
        if (pidx == 0) {
            x = rec[3]; y = rec[4]; m = rec[7];
        } else {
            const PartWalk w = walk[pidx];
            const int px = xv[w.parent], py = yv[w.parent], pm = mv[w.parent];
            const size_t o = pbase + (size_t)(w.slot + pm) * HW + (size_t)py * W + px;
            x = p.Ix[o]; y = p.Iy[o]; m = p.Ik[o];
        }
        xv[pidx] = x; yv[pidx] = y; mv[pidx] = m;
        const int ks = walk[pidx].ksize[m];
        const int x1 = round_mul<R>(x - 1, scale), y1 = round_mul<R>(y - 1, scale);
        const int x2 = x1 + round_mul<R>(ks, scale) - 1, y2 = y1 + round_mul<R>(ks, scale) - 1;
        const int rx = min(x1, x2), ry = min(y1, y2);
        rects[pidx * 4 + 0] = rx;
        rects[pidx * 4 + 1] = ry;
        rects[pidx * 4 + 2] = max(x1, x2) - rx;
        rects[pidx * 4 + 3] = max(y1, y2) - ry;
    }
    rec[6] = nparts;
    rec[7] = 0;
}

void launch_argmin_walk(const ArgminParams &p, int ncand, bool f64, hipStream_t s)
{
    if (ncand == 0) return;
    if (f64) hipLaunchKernelGGL(k_argmin_walk<double>, dim3((ncand + 63) / 64), dim3(64), 0, s, p, ncand);
    else hipLaunchKernelGGL(k_argmin_walk<float>, dim3((ncand + 63) / 64), dim3(64), 0, s, p, ncand);
}

}  // namespace pbd
