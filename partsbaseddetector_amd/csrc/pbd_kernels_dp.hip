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

#include <algorithm>

namespace pbd {

// Instrumented build (-DPBD_DT_STATS, tools/dt_stats.sh): counts how often the wave executes each part of the transform
// and for how many lanes -- the difference is what the lock step of 64 independent rows costs.  Not in the product build.
#ifdef PBD_DT_STATS
__device__ unsigned long long g_dt_stats[16];
#define DT_STAT(i)                                                                                      \
    do {                                                                                                \
        const unsigned long long m_ = __ballot(1);                                                      \
        if ((int)__lane_id() == __ffsll((long long)m_) - 1) {                                           \
            atomicAdd(&g_dt_stats[2 * (i)], 1ull);                                                      \
            atomicAdd(&g_dt_stats[2 * (i) + 1], (unsigned long long)__popcll(m_));                      \
        }                                                                                               \
    } while (0)
#else
#define DT_STAT(i)
#endif

// BZ: the linear coefficient b is exactly -0.0 (deformation weight +0.0f, the usual model) and a != 0.  Then
// b * d is -0.0 for the d > 0 of an intersection and (y1 - y0) - (-0.0) == y1 - y0 for every value, -0.0 included;
// in the value expression b * x is +-0 and a * x*x + (+-0) == a * x*x unless that is itself a zero of the other
// sign, which happens only for x == 0, where b * x = -0.0 * 0 = -0.0 and t + (-0.0) == t for every t.
template <typename R, bool BZ>
__device__ __forceinline__ R quad_isect(double a, double b, int x0, int x1, R y0f, R y1f)
{   // Quadratic::operator()(x0, x1, y0, y1), include/DistanceTransform.hpp:98-100, rounded to T
    const double y0 = (double)y0f, y1 = (double)y1f;
    // x1*x1 - x0*x0 as (x1 - x0) * (x1 + x0): the same integer (0 <= x0 < x1 < 2^16), one 24-bit multiply instead of two
    // full-width ones (quarter rate on this VALU)
    const int dx = x1 - x0;
    const double dd = (double)dx;
    const double sq = (double)(int)__umul24((unsigned)dx, (unsigned)(x1 + x0));
    const double num = BZ ? (y1 - y0) + a * sq : ((y1 - y0) - b * dd) + a * sq;
    return (R)(num / ((2 * a) * dd));
}
template <typename R, bool BZ>
__device__ __forceinline__ R quad_val(double a, double b, int x, R y)
{   // Quadratic::operator()(x, y), :103-105
    if (BZ) return (R)(a * (double)__mul24(x, x) + (double)y);
    return (R)((a * (double)__mul24(x, x) + b * (double)x) + (double)y);     // |x| < 2^16
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
//   * when a lane's ring is full its two oldest entries (2p, 2p+1) are spilled TOGETHER, as one 16-byte record
//     {s[2p], s[2p+1], z[2p], v[2p] | v[2p+1] << 16}, to the wave-private, lane-interleaved global stack
//     ([p][lane]); z[2p+1] is the intersection of the two, recomputed on reload with the same expression on the
//     same operands (bit-identical).  A lane spills about once per ten elements, but with 64 independent lanes
//     the WAVE runs the spill and reload paths about once per element (tools/dt_stats.sh: 0.78 and 1.17
//     executions per element, 8 and 5 lanes active), each a memory round trip the whole wave waits for: what
//     counts is the NUMBER of requests, which pairing halves (a 6-byte s/v split over two stores was slower; so
//     was carrying z[2p+1] in a second 4-byte array to save the 24-instruction recomputation: rows pass 8.6 ->
//     9.1 ms on the same box);
//   * the read-out walks q downwards and POPS: since z[1..ktop] is strictly increasing,
//     "k = 0; while (z[k+1] < os) k++" (DistanceTransform.hpp:172-178, q ascending) selects the same
//     k(q) = max{k : z[k] < os(q)} as "k = ktop; while (!(z[k] < os)) k--" with q descending;
//   * source values are prefetched one chunk ahead and results leave in whole chunks.
// The arithmetic per element is exactly computeRow's (DistanceTransform.hpp:152-182).
// ------------------------------------------------------------------------------------------------
template <typename R> struct StkPairT { R sa, sb, za; unsigned vv; };
static_assert(sizeof(StkPairT<float>) == kStkPairF32 && sizeof(StkPairT<double>) == kStkPairF64, "host sizes the spill stack with these");

#ifndef PBD_DT_CH
#define PBD_DT_CH 16
#endif
constexpr int kDtCH = PBD_DT_CH;   // elements per streamed chunk of the rows pass (multiple of 4)
#ifndef PBD_DT_CHC
#define PBD_DT_CHC 16
#endif
constexpr int kDtCHC = PBD_DT_CHC;  // ... of the columns pass
#ifndef PBD_DT_WAVES
#define PBD_DT_WAVES 1
#endif
constexpr int kDtWaves = PBD_DT_WAVES;   // waves per workgroup of the DT passes (each wave = 64 rows / columns)
#ifndef PBD_DT_RING
#define PBD_DT_RING 8
#endif
constexpr int kDtT = PBD_DT_RING;    // ring entries per lane (power of two)

// LDS layout of a wave's ring: [slot][z: 64 x R | s: 64 x R | v: 64 x int]; one address per lane, the rest
// are immediate offsets.
// NARROW (launches that do not fill the chip, see k_dt_rows: the wave uses its first L = 64 >> lane_shift lanes only): the same
// 6 KB hold kDtT << lane_shift entries per lane, laid out [slot][z: L x R | s: L x R | v: L x int] -- with 16 lanes the ring is
// 32 deep, with 4 lanes 128: rows of a VGA pyramid never spill, and a wave alone on its SIMD no longer waits for the memory
// round trips of the spill / reload paths.  The geometry is then a run-time value (three more integer registers).
template <typename R, bool NARROW>
struct DtRing {
    static constexpr int kSlotBytes = 64 * (2 * (int)sizeof(R) + 4);
    char *zs;                     // this lane's z of slot 0 (s is 64 (L) R further)
    char *vp;                     // this lane's v of slot 0
    StkPairT<R> *g;               // this lane's column of the global [pair][lane] stack
    double a, b;                  // the job's quadratic (z of the upper entry of a reloaded pair)
    int lo;                       // ring holds indices [lo, top); lo is even; entries below lo are spilled
    int tmask, sbytes, soff;      // NARROW: entries - 1, bytes per slot, byte offset of s behind z
    __device__ __forceinline__ int T() const { return NARROW ? tmask + 1 : kDtT; }
    __device__ __forceinline__ int slot_of(int idx) const { return NARROW ? (idx & tmask) : (idx & (kDtT - 1)); }
    __device__ __forceinline__ int off(int slot) const { return NARROW ? slot * sbytes : slot * kSlotBytes; }
    __device__ __forceinline__ R &z(int slot) { return *reinterpret_cast<R *>(zs + off(slot)); }
    __device__ __forceinline__ R &s(int slot) { return *reinterpret_cast<R *>(zs + off(slot) + (NARROW ? soff : 64 * (int)sizeof(R))); }
    __device__ __forceinline__ int &v(int slot) { return *reinterpret_cast<int *>(vp + off(slot)); }
    __device__ __forceinline__ void push_below(int idx, R zk, R sk, int vk)
    {   // entry `idx` (the old top) moves under a new top
        const int slot = slot_of(idx);
        if (idx - lo >= T()) {   // ring full: spill its two oldest entries lo, lo + 1 as one record
            DT_STAT(3);
            const int sl = slot_of(lo);
            g[(size_t)(lo >> 1) * 64] = StkPairT<R>{s(sl), s(sl + 1), z(sl), (unsigned)v(sl) | ((unsigned)v(sl + 1) << 16)};
            lo += 2;
        }
        z(slot) = zk; s(slot) = sk; v(slot) = vk;
    }
    template <bool BZ>
    __device__ __forceinline__ void pop(int idx, R &zk, R &sk, int &vk)
    {   // entry `idx` becomes the top
        // the ring slot is read unconditionally (always a valid LDS address) so that the common case is
        // plain LDS reads; only a pop below the ring overrides it from the spill stack
        const int slot = slot_of(idx);
        zk = z(slot); sk = s(slot); vk = v(slot);
        asm volatile("" : "+v"(zk), "+v"(sk), "+v"(vk));   // keep these as LDS reads (not a flat load of a selected pointer)
        if (idx < lo) {
            DT_STAT(4);
            // the ring is empty (idx == lo - 1, odd): reload the pair (lo-2, lo-1); the lower entry goes back
            // into the ring, the upper one is the new top and gets its z recomputed
            const StkPairT<R> e = g[(size_t)((lo - 2) >> 1) * 64];
            const int va = (int)(e.vv & 0xffffu), vb = (int)(e.vv >> 16);
            const int sl = slot_of(lo - 2);
            z(sl) = e.za; s(sl) = e.sa; v(sl) = va;
            sk = e.sb; vk = vb;
            zk = BZ ? quad_isect<R, true>(a, b, va, vb, e.sa, e.sb) : quad_isect<R, false>(a, b, va, vb, e.sa, e.sb);   // as computed when entry lo-1 was pushed onto entry lo-2
            lo -= 2;
        }
    }
    // the ring of the workgroup's wave inside `smem`; `lanep` = the lane's number inside the wave, sh = lane_shift
    static __device__ __forceinline__ DtRing make(char *smem, int lanep, int sh, StkPairT<R> *g, double a, double b)
    {
        const int L = NARROW ? (64 >> sh) : 64;
        return DtRing{smem + lanep * (int)sizeof(R), smem + 2 * L * (int)sizeof(R) + lanep * 4, g, a, b, 0,
                      (kDtT << (NARROW ? sh : 0)) - 1, L * (2 * (int)sizeof(R) + 4), L * (int)sizeof(R)};
    }
};

// Position chunks (the read-out's pointers, the columns pass's carried pointers) are kept EPW to a 32-bit register: 4 when the
// positions are bytes (uint8 planes), 1 otherwise -- sixteen-element chunks then cost 4 registers instead of 16 each, which is
// what keeps the columns pass at 6 waves per SIMD without shortening its chunks (8-element chunks were 4 % faster than
// unpacked 16-element ones but moved 1.5x the bytes: every 128-byte line of a column was fetched four times instead of twice).
template <int EPW> __device__ __forceinline__ int dt_get(const int *w, int i)
{
    constexpr int BITS = 32 / EPW;          // 8-bit fields (uint8 planes), 16-bit fields (int16 planes; positions are >= 0)
    return EPW == 1 ? w[i] : (int)(((unsigned)w[i / EPW] >> (BITS * (i % EPW))) & ((1u << BITS) - 1u));
}
template <int EPW> __device__ __forceinline__ void dt_put(int *w, int i, int v)     // the word was zeroed before its first element
{
    constexpr int BITS = 32 / EPW;
    if (EPW == 1) w[i] = v;
    else w[i / EPW] |= v << (BITS * (i % EPW));
}

// AUX: the read-out additionally streams an int chunk per output chunk (prefetched one chunk ahead, q
// descending) and hands it to `store` -- the columns pass uses it to carry the rows pass's pointers along.
template <typename R, bool AUX, bool BZ, int CH, int EPW, bool NARROW, class LoadChunk, class StoreChunk, class AuxChunk>
__device__ __forceinline__ void dt_stream(int N, double a, double b, int os0, DtRing<R, NARROW> ring, LoadChunk load, StoreChunk store,
                                          AuxChunk aux)
{
    R cur[CH], nxt[CH];
    load(0, cur);
    int k = 0, vk = 0;
    R zk = -RealLimits<R>::inf(), sk = cur[0];
    ring.lo = 0;
    for (int q0 = 0; q0 < N; q0 += CH) {
        if (q0 + CH < N) load(q0 + CH, nxt);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int q = q0 + i;
            if (q >= 1 && q < N) {
                const R sq = cur[i];
                DT_STAT(0);
                R s = quad_isect<R, BZ>(a, b, vk, q, sk, sq);
                while (s <= zk && k > 0) {
                    DT_STAT(1);
                    --k;
                    ring.template pop<BZ>(k, zk, sk, vk);
                    s = quad_isect<R, BZ>(a, b, vk, q, sk, sq);
                }
                ring.push_below(k, zk, sk, vk);
                ++k;
                vk = q; zk = s; sk = sq;
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) cur[i] = nxt[i];
    }
    // read-out, q descending
    const int nch = (N + CH - 1) / CH;
    constexpr int NW = CH / EPW;
    int aux_cur[NW], aux_nxt[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) { aux_cur[i] = 0; aux_nxt[i] = 0; }
    if (AUX) aux((nch - 1) * CH, aux_cur);
    for (int cidx = nch - 1; cidx >= 0; --cidx) {
        const int q0 = cidx * CH;
        if (AUX && cidx > 0) aux(q0 - CH, aux_nxt);
        R out[CH];
        int ptr[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) ptr[i] = 0;
#pragma unroll
        for (int i = CH - 1; i >= 0; --i) {
            out[i] = (R)0;
            const int q = q0 + i;
            if (q < N) {
                const R osf = (R)(os0 + q);
                DT_STAT(5);
                while (!(zk < osf)) {   // z[0] = -inf ends the walk
                    DT_STAT(2);
                    --k;
                    ring.template pop<BZ>(k, zk, sk, vk);
                }
                out[i] = quad_val<R, BZ>(a, b, os0 + q - vk, sk);
                dt_put<EPW>(ptr, i, vk);
            }
        }
        store(q0, out, ptr, aux_cur);
        if (AUX) {
#pragma unroll
            for (int i = 0; i < NW; ++i) aux_cur[i] = aux_nxt[i];
        }
    }
}

typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
typedef short v8s_u __attribute__((ext_vector_type(8), aligned(2)));
typedef _Float16 v8h_u __attribute__((ext_vector_type(8), aligned(2)));
typedef unsigned vchw_u __attribute__((ext_vector_type(PBD_DT_CHC / 4), aligned(1)));
static_assert(kDtCHC % 8 == 0 && kDtCH % 8 == 0, "int16 pointers and fp16 responses are read 8 at a time");

// ---- rows pass: thread = (flat row, job, frame); each lane streams its own row with 16-byte accesses ----
// RH: the responses are fp16 (PBD_CONV_MFMA_F16); a template parameter so that the default kernels carry none of it
// PT: element type of the position planes (uint8_t when no map side exceeds 256, else int16_t)
template <typename R, bool RH, typename PT, bool BZ, bool NARROW>
__global__ __launch_bounds__(64 * kDtWaves) void k_dt_rows(DpParams p)
{
    constexpr int EPW = 4 / (int)sizeof(PT);
    // grid = (job, frame, wave of 64 flat rows): the wave index is the SLOWEST dimension, so the long rows of
    // the large levels are dispatched first and the tail of the launch is made of short ones
    // lane_shift (0 .. 6): a group of 64 flat rows is spread over 1 .. 64 waves that use their first 64 .. 1 lanes only.
    // A launch that does not fill the chip anyway (one frame, a few 1080p frames) then runs as more, narrower waves: the wave's
    // pop loops iterate for the slowest of a few lanes instead of 64, and the launch takes what its longest rows take (launch_dt_rows).
    static_assert(kDtWaves == 1, "one wave per workgroup");
    const int sh = p.lane_shift, lanep = threadIdx.x;
    if (lanep >= (64 >> sh)) return;
    const int wv = (int)blockIdx.z >> sh, lane = (((int)blockIdx.z & ((1 << sh) - 1)) << (6 - sh)) + lanep;
    if (wv * 64 >= p.nrows_flat) return;
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
    // fp16 responses (PBD_CONV_MFMA_F16): a leaf part's input is read as halves, same element index
    const bool hsrc = RH && !job.from_acc;
    const _Float16 *srch = static_cast<const _Float16 *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F +
                           (size_t)job.plane * HW + (size_t)y * W;
    // both outputs go out TRANSPOSED ([x][y]): lanes are adjacent rows y, so every store instruction writes whole lines.  The
    // columns pass reads its column of values back with wide per-lane loads; the pointers go straight to their persistent plane
    // (IxRaw, kept transposed: only the candidates' walk and pbd_dp_min's read-back ever index it) -- the columns pass used to
    // carry them along (a load, 16 extracts and 16 byte stores per chunk, 8 registers) only to transpose them for the combine
    // step, which no longer reads them
    const int Hl = d.rows;
    const size_t jb = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG + (size_t)j * HW;
    R *tmpT = static_cast<R *>(p.tmp) + jb + (size_t)y;
    PT *ixT = static_cast<PT *>(p.IxRaw) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NJ + (size_t)job.gm * HW + (size_t)y;
    __shared__ __attribute__((aligned(16))) char ring_mem[kDtWaves * kDtT * DtRing<R, NARROW>::kSlotBytes];
    DtRing<R, NARROW> ring = DtRing<R, NARROW>::make(ring_mem, lanep, sh,
                                     reinterpret_cast<StkPairT<R> *>(p.stk) +
                                         ((size_t)(fl * p.JG + j) * p.stk_per_jf + p.stk_row_off[wv]) + lane, job.ax, job.bx);
    const int N = active ? W : 0;
    if (N == 0) return;
    auto load = [&](int q0, R *buf) {
        if (hsrc) {
            if (q0 + kDtCH <= N) {
#pragma unroll
                for (int v = 0; v < kDtCH / 8; ++v) {
                    const v8h_u a0 = *reinterpret_cast<const v8h_u *>(srch + q0 + 8 * v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) buf[8 * v + e] = (R)(float)a0[e];
                }
            } else {
#pragma unroll
                for (int i = 0; i < kDtCH; ++i) buf[i] = (q0 + i < N) ? (R)(float)srch[q0 + i] : (R)0;
            }
        } else if (sizeof(R) == 4 && q0 + kDtCH <= N) {
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
    };
    auto store = [&](int q0, const R *out, const int *ptr, const int *) {
        // the element pointers advance by additions (a 64-bit multiply per store is a quarter-rate instruction)
        R *tp = tmpT + (size_t)q0 * Hl;
        PT *ip = ixT + (size_t)q0 * Hl;
#pragma unroll
        for (int i = 0; i < kDtCH; ++i) {
            if (q0 + i < N) { *tp = out[i]; *ip = (PT)dt_get<EPW>(ptr, i); }
            tp += Hl; ip += Hl;
        }
    };
    auto noaux = [](int, int *) {};
    dt_stream<R, false, BZ, kDtCH, EPW, NARROW>(N, job.ax, job.bx, job.osx, ring, load, store, noaux);
}

// Rows (columns) per wave of a pass: 64 when the launch fills the chip (1024 SIMDs), else 32 .. 1 -- see k_dt_rows.
static int dt_lane_shift(long long waves64)
{
    static const int forced = getenv("PBD_DT_LANESHIFT") ? atoi(getenv("PBD_DT_LANESHIFT")) : -1;
    if (forced >= 0 && forced <= 6) return forced;
    int sh = 0;
    while (sh < 6 && (waves64 << (sh + 1)) <= 6144) ++sh;       // at most one round of waves (six per SIMD)
    return sh;
}

template <bool COLS> static void launch_dt_coop(const DpParams &p, int nframes, int nflat, bool bz, hipStream_t s);
static bool dt_coop_enabled(int longest);
static int dt_coop_min_shift();

void launch_dt_rows(const DpParams &p0, int nframes, bool f64, hipStream_t s)
{
    if (p0.JG == 0 || p0.nrows_flat == 0) return;
    const int nwv = (p0.nrows_flat + 63) / 64;
    DpParams p = p0;
    p.lane_shift = dt_lane_shift((long long)p.JG * nframes * nwv);
    if (p.lane_shift >= dt_coop_min_shift() && !f64 && !p.resp_half && p.ptr8 && dt_coop_enabled(p.longest)) { launch_dt_coop<false>(p, nframes, p.nrows_flat, p.bz_x != 0, s); return; }
    dim3 grid(p.JG, nframes, nwv << p.lane_shift);
#define PBD_ROWS(PT, BZ)                                                                                              \
    do {                                                                                                              \
        if (f64 && p.lane_shift > 0) PBD_LAUNCH((k_dt_rows<double, false, PT, BZ, true>), grid, dim3(64 * kDtWaves), 0, s, p); \
        else if (f64) PBD_LAUNCH((k_dt_rows<double, false, PT, BZ, false>), grid, dim3(64 * kDtWaves), 0, s, p);          \
        else if (p.resp_half) PBD_LAUNCH((k_dt_rows<float, true, PT, BZ, false>), grid, dim3(64 * kDtWaves), 0, s, p); \
        else if (p.lane_shift > 0) PBD_LAUNCH((k_dt_rows<float, false, PT, BZ, true>), grid, dim3(64 * kDtWaves), 0, s, p); \
        else PBD_LAUNCH((k_dt_rows<float, false, PT, BZ, false>), grid, dim3(64 * kDtWaves), 0, s, p);               \
    } while (0)
    if (p.ptr8) { if (p.bz_x) PBD_ROWS(uint8_t, true); else PBD_ROWS(uint8_t, false); }
    else { if (p.bz_x) PBD_ROWS(int16_t, true); else PBD_ROWS(int16_t, false); }
#undef PBD_ROWS
}

// ---- columns pass: thread = (flat column, job, frame); lanes are adjacent columns -> coalesced ----
template <typename R, typename PT, bool BZ, bool NARROW>
__global__ __launch_bounds__(64 * kDtWaves) __attribute__((amdgpu_waves_per_eu(sizeof(R) == 4 ? (sizeof(PT) == 1 && !NARROW ? 6 : 5) : 1)))
void k_dt_cols(DpParams p)
{
    constexpr int EPW = 4 / (int)sizeof(PT);
    const int sh = p.lane_shift, lanep = threadIdx.x;                                      // as in the rows pass
    if (lanep >= (64 >> sh)) return;
    const int wv = (int)blockIdx.z >> sh, lane = (((int)blockIdx.z & ((1 << sh) - 1)) << (6 - sh)) + lanep;   // longest columns first
    const int cidx = wv * 64 + lane;
    if (cidx >= p.ncols_flat) return;
    const int j = blockIdx.x, fl = blockIdx.y;
    const int l = p.col2level[cidx];
    const LevelDesc d = p.lv[l];
    const int x = cidx - p.coloff[l];
    const int H = d.rows, W = d.cols;
    const size_t HW = (size_t)H * W;
    const DtJob job = p.jobs[j];
    const size_t jbase = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG + (size_t)j * HW;
    const R *tmpT = static_cast<const R *>(p.tmp) + jbase + (size_t)x * H;     // this lane's column, contiguous
    R *dt = static_cast<R *>(p.dt) + jbase + x;
    PT *iyr = static_cast<PT *>(p.IyRaw) + ((size_t)(p.frame0 + fl) * p.cell_per_frame + d.cell_off) * p.NJ + (size_t)job.gm * HW + x;
    __shared__ __attribute__((aligned(16))) char ring_mem[kDtWaves * kDtT * DtRing<R, NARROW>::kSlotBytes];
    DtRing<R, NARROW> ring = DtRing<R, NARROW>::make(ring_mem, lanep, sh,
                                     reinterpret_cast<StkPairT<R> *>(p.stk) +
                                         ((size_t)(fl * p.JG + j) * p.stk_per_jf + p.stk_col_off[wv]) + lane, job.ay, job.by);
    auto load = [&](int q0, R *buf) {
        if (sizeof(R) == 4 && q0 + kDtCHC <= H) {
            const float *srcf = reinterpret_cast<const float *>(tmpT);
#pragma unroll
            for (int v = 0; v < kDtCHC / 4; ++v) {
                const v4f_u a0 = *reinterpret_cast<const v4f_u *>(srcf + q0 + 4 * v);
                buf[4 * v] = a0.x; buf[4 * v + 1] = a0.y; buf[4 * v + 2] = a0.z; buf[4 * v + 3] = a0.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kDtCHC; ++i) buf[i] = (q0 + i < H) ? tmpT[q0 + i] : (R)0;
        }
    };
    auto noaux = [](int, int *) {};
    auto store = [&](int q0, const R *out, const int *ptr, const int *) {
        R *dp = dt + (size_t)q0 * W;
        PT *yp = iyr + (size_t)q0 * W;
#pragma unroll
        for (int i = 0; i < kDtCHC; ++i) {
            if (q0 + i < H) {
                *dp = out[i];
                *yp = (PT)dt_get<EPW>(ptr, i);
            }
            dp += W; yp += W;
        }
    };
    dt_stream<R, false, BZ, kDtCHC, EPW, NARROW>(H, job.ay, job.by, job.osy, ring, load, store, noaux);
}

void launch_dt_cols(const DpParams &p0, int nframes, bool f64, hipStream_t s)
{
    if (p0.JG == 0 || p0.ncols_flat == 0) return;
    const int nwv = (p0.ncols_flat + 63) / 64;
    DpParams p = p0;
    p.lane_shift = dt_lane_shift((long long)p.JG * nframes * nwv);
    if (p.lane_shift >= dt_coop_min_shift() && !f64 && p.ptr8 && dt_coop_enabled(p.longest)) { launch_dt_coop<true>(p, nframes, p.ncols_flat, p.bz_y != 0, s); return; }
    dim3 grid(p.JG, nframes, nwv << p.lane_shift);
#define PBD_COLS(PT, BZ)                                                                                   \
    do {                                                                                                   \
        if (f64 && p.lane_shift > 0) PBD_LAUNCH((k_dt_cols<double, PT, BZ, true>), grid, dim3(64 * kDtWaves), 0, s, p); \
        else if (f64) PBD_LAUNCH((k_dt_cols<double, PT, BZ, false>), grid, dim3(64 * kDtWaves), 0, s, p);      \
        else if (p.lane_shift > 0) PBD_LAUNCH((k_dt_cols<float, PT, BZ, true>), grid, dim3(64 * kDtWaves), 0, s, p); \
        else PBD_LAUNCH((k_dt_cols<float, PT, BZ, false>), grid, dim3(64 * kDtWaves), 0, s, p);           \
    } while (0)
    if (p.ptr8) { if (p.bz_y) PBD_COLS(uint8_t, true); else PBD_COLS(uint8_t, false); }
    else { if (p.bz_y) PBD_COLS(int16_t, true); else PBD_COLS(int16_t, false); }
#undef PBD_COLS
}

// ---- wavefront-cooperative form of a pass: FOUR rows (columns) per wave, sixteen lanes each --------------------------------
// For launches of so few rows that narrow waves of eight lanes or fewer would be used (one frame: lane_shift >= 3).  A narrow wave is
// bound by the number of instructions it issues, and three quarters of the lanes of a four-row wave do nothing; here a row's
// sixteen lanes hold the TOP SIXTEEN entries of its envelope (entry e in lane e & 15 while e is in the aligned block of the
// top; every entry is also written through to LDS, from where a lower block is fetched back when the whole window is popped).
// Inserting element q evaluates the reference's pop predicate -- s(v[e], q) <= z[e] && e > 0, the same expression on the same
// operands -- for all window entries AT ONCE; the sequential loop pops from the top until the first entry whose predicate is
// false, i.e. the new top is the highest entry that stays (ballot + find-first-bit) and the pushed z is that entry's
// intersection.  No pop loop, no divergence: about 95 instructions per element for four rows, whatever the data.  The read-out is
// a binary search of the (increasing) z of the finished envelope in LDS for sixteen positions of a row at a time:
// max{k : z[k] < os + q}, what the reference's "while (z[k+1] < os) k++" selects.  Outputs as in the plain passes.
// G rows per wave, W = 64 / G lanes each (G = 4: window of 16 entries; G = 8: window of 8 -- half the waves for launches that
// would otherwise need more than one round of them, at the price of more fetches of a lower block)
template <bool BZ, typename PT, bool COLS, int G>
__global__ __launch_bounds__(64) void k_dt_coop(DpParams p)
{
    constexpr int kCoopRows = G, kCoopW = 64 / G, kCoopLog = (G == 4 ? 4 : 3);
    static_assert(G == 4 || G == 8, "four or eight rows per wave");
    extern __shared__ __attribute__((aligned(16))) char coop_mem[];
    const int lanep = threadIdx.x, grp = lanep >> kCoopLog, sub = lanep & (kCoopW - 1), gl0 = grp << kCoopLog;
    const int nflat = COLS ? p.ncols_flat : p.nrows_flat;
    const int r = (int)blockIdx.z * kCoopRows + grp;
    const bool active = r < nflat;
    const int rr = active ? r : nflat - 1;
    const int j = blockIdx.x, fl = blockIdx.y, frame = p.frame0 + fl;
    const int l = (COLS ? p.col2level : p.row2level)[rr];
    const LevelDesc d = p.lv[l];
    const int idx = rr - (COLS ? p.coloff : p.rowoff)[l];          // y of the row / x of the column
    const int H = d.rows, W = d.cols;
    const size_t HW = (size_t)H * W;
    const DtJob job = p.jobs[j];
    const int N = active ? (COLS ? H : W) : 0;
    const double a = COLS ? job.ay : job.ax, b = COLS ? job.by : job.bx;
    const int os0 = COLS ? job.osy : job.osx;
    const size_t jb = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG + (size_t)j * HW;
    const float *src;
    if (COLS) src = static_cast<const float *>(p.tmp) + jb + (size_t)idx * H;
    else src = (job.from_acc ? static_cast<const float *>(p.acc) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NM
                             : static_cast<const float *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F) +
               (size_t)job.plane * HW + (size_t)idx * W;
    // outputs: element q at out + q * ostr (rows pass: transposed planes [x][y]; columns pass: [y][x])
    const int ostr = COLS ? W : H;
    float *outv = (COLS ? static_cast<float *>(p.dt) : static_cast<float *>(p.tmp)) + jb + idx;
    PT *outp = static_cast<PT *>(COLS ? p.IyRaw : p.IxRaw) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NJ + (size_t)job.gm * HW + idx;
    const int maxn = p.longest;
    float *zs = reinterpret_cast<float *>(coop_mem) + (size_t)grp * 3 * maxn, *ys = zs + maxn;
    int *vs = reinterpret_cast<int *>(ys + maxn);
    int nm = N;
#pragma unroll
    for (int o = kCoopW; o < 64; o <<= 1) nm = max(nm, __shfl_xor(nm, o));
    const int Nmax = __builtin_amdgcn_readfirstlane(nm);
    if (Nmax == 0) return;

    // ---- scan: the envelope of the row, top block in registers (this lane: entry wbase + sub), all of it in LDS
    // (requesting the step's two cross-lane values -- the element, the pushed entry's z -- one step early was measured and
    // changed nothing: the kernel is bound by the instructions it issues, not by the latency of these reads)
    float cur = (sub < N) ? src[sub] : 0.0f;
    float ez = -INFINITY, ey = __shfl(cur, gl0);
    int ev = 0, k = 0, wbase = 0;
    if (sub == 0 && N > 0) { zs[0] = ez; ys[0] = ey; vs[0] = 0; }
    for (int q0 = 0; q0 < Nmax; q0 += kCoopW) {
        const float nxt = (q0 + kCoopW + sub < N) ? src[q0 + kCoopW + sub] : 0.0f;
        for (int i = (q0 == 0 ? 1 : 0); i < kCoopW && q0 + i < Nmax; ++i) {
            const int q = q0 + i;
            const bool rowact = q < N;
            const float yq = __shfl(cur, gl0 + i);
            float s;
            unsigned m;
            for (;;) {
                const int e = wbase + sub;
                s = quad_isect<float, BZ>(a, b, ev, q, ey, yq);
                const bool stay = rowact && e <= k && !((s <= ez) && e > 0);      // entry 0 always stays (the reference's k > 0)
                m = (unsigned)(__ballot(stay) >> gl0) & ((1u << kCoopW) - 1u);
                const bool lower = rowact && m == 0;       // the whole window is popped: the top moves into the block below
                if (!__any(lower)) break;
                if (lower) { wbase -= kCoopW; k = wbase + kCoopW - 1; ez = zs[wbase + sub]; ey = ys[wbase + sub]; ev = vs[wbase + sub]; }
            }
            const int top = wbase + (rowact ? 31 - __clz((int)m) : 0);     // the highest entry that stays
            const float snew = __shfl(s, gl0 + (top & (kCoopW - 1)));       // its intersection with q: z of the pushed entry
            if (rowact) {
                k = top + 1;
                if ((k & (kCoopW - 1)) == 0) wbase = k;                      // the pushed entry opens a new block
                if (sub == (k & (kCoopW - 1))) { ev = q; ey = yq; ez = snew; zs[k] = snew; ys[k] = yq; vs[k] = q; }
            }
        }
        cur = nxt;
    }
    // ---- read-out: sixteen positions of the row at a time
    for (int p0 = 0; p0 < Nmax; p0 += kCoopW) {
        const int pq = p0 + sub;
        const float osf = (float)(os0 + pq);
        int lo = 0, hi = k;
        while (__any(lo < hi)) {
            const int mid = (lo + hi + 1) >> 1;
            const bool lt = zs[mid] < osf;
            if (lo < hi) { if (lt) lo = mid; else hi = mid - 1; }
        }
        if (pq < N) {
            outv[(size_t)pq * ostr] = quad_val<float, BZ>(a, b, os0 + pq - vs[lo], ys[lo]);
            outp[(size_t)pq * ostr] = (PT)vs[lo];
        }
    }
}

// launches that would run with 8 or fewer rows per wave go to the cooperative kernel (measured, one 640x480 frame: from 4 rows
// per wave on 2.49 ms, from 8 on 2.37, from 16 on 2.43; a 1080p frame gets slower from 16 on: more waves than the chip holds)
static int dt_coop_min_shift() { return 3; }
static bool dt_coop_enabled(int longest)
{
    static const int v = getenv("PBD_DT_COOP") ? atoi(getenv("PBD_DT_COOP")) : 1;
    // the envelopes of a wave's rows live in LDS (12 bytes per element): beyond 12 KB per wave too few waves fit a CU (a
    // 1080p frame with four rows per wave: 23 KB, and its single-frame rate fell from 83 to 69 detections/s)
    return v != 0 && (size_t)4 * 3 * longest * sizeof(float) <= (size_t)12 * 1024;
}

template <bool COLS, int G>
static void launch_dt_coop_g(const DpParams &p, int nframes, int nflat, bool bz, hipStream_t s)
{
    dim3 grid(p.JG, nframes, (nflat + G - 1) / G);
    const unsigned lds = (unsigned)((size_t)G * 3 * p.longest * sizeof(float));
    // (rows of at most 256 elements: the position planes are uint8 -- dt_coop_enabled)
    if (bz) PBD_LAUNCH((k_dt_coop<true, uint8_t, COLS, G>), grid, dim3(64), lds, s, p);
    else PBD_LAUNCH((k_dt_coop<false, uint8_t, COLS, G>), grid, dim3(64), lds, s, p);
}
template <bool COLS>
static void launch_dt_coop(const DpParams &p, int nframes, int nflat, bool bz, hipStream_t s)
{
    // four rows per wave while that stays within one round of waves, else eight
    static const int forced = getenv("PBD_DT_COOP_G") ? atoi(getenv("PBD_DT_COOP_G")) : 0;
    const long long waves4 = (long long)p.JG * nframes * ((nflat + 3) / 4);
    const bool g8 = forced ? forced == 8 : (waves4 > 6144 && (size_t)8 * 3 * p.longest * sizeof(float) <= (size_t)12 * 1024);
    if (g8) launch_dt_coop_g<COLS, 8>(p, nframes, nflat, bz, s);
    else launch_dt_coop_g<COLS, 4>(p, nframes, nflat, bz, s);
}

// ---- combine: thread = 4 consecutive cells of one level, one PARENT part (block.y) ---------------------
// For every parent mixture m: acc = response(parent, m); then for each child in descending index order
//   weighted[mm] = score_dt[child][mm] + bias(mm)[m]; reduceMax (strict >, first wins, start -inf; K==1 copies);
//   the winning mixture is recorded as Ik (reducePickIndex); acc += max   (src/DynamicProgram.cpp:134-156).
//   The reference also picks Ix / Iy of the winner here for every cell (with the composition Iy[y][x] = IyRaw[y][Ix[y][x]],
//   include/DistanceTransform.hpp:233-244); this path leaves them in the transform's own planes and composes them for the
//   candidates only (k_argmin_walk) and on read-back (pbd_dp_min).
// The accumulated plane is the input of the parent's own distance transform in the next group.
// Four cells per thread: every plane is read and written with 16 / 8 / 4-byte accesses per lane.
template <typename T, int N> struct CellVec;
#define PBD_CELLVEC(T, E, N) template <> struct CellVec<T, N> { typedef E type __attribute__((ext_vector_type(N), aligned(sizeof(E)))); }
PBD_CELLVEC(float, float, 4); PBD_CELLVEC(float, float, 2); PBD_CELLVEC(double, double, 4); PBD_CELLVEC(double, double, 2);
PBD_CELLVEC(int16_t, short, 4); PBD_CELLVEC(int16_t, short, 2); PBD_CELLVEC(uint8_t, unsigned char, 4); PBD_CELLVEC(uint8_t, unsigned char, 2);
PBD_CELLVEC(_Float16, _Float16, 4); PBD_CELLVEC(_Float16, _Float16, 2);
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
// the same without the test: the caller tolerates cells past the end of the level (they come from the next plane, the buffers
// end in slack) because it never stores them -- one wide load instead of a divergent region per access
template <typename T, int kCpt>
__device__ __forceinline__ void load_cells_wide(const T *src, T *dst)
{
    const typename CellVec<T, kCpt>::type v = *reinterpret_cast<const typename CellVec<T, kCpt>::type *>(src);
#pragma unroll
    for (int e = 0; e < kCpt; ++e) dst[e] = (T)v[e];
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
template <typename R, int kCpt, int MAXM, bool RH, typename PT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MAXM <= 6 && sizeof(R) == 4 ? 4 : 1)))
void k_dp_combine(DpParams p)
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
    const R *resp = static_cast<const R *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + local;
    const R *dtp = static_cast<const R *>(p.dt);
    const size_t gbase0 = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG;
    const size_t pbase = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NS + local;
    R accv[MAXM][kCpt];
#pragma unroll
    for (int pm = 0; pm < MAXM; ++pm) {
#pragma unroll
        for (int e = 0; e < kCpt; ++e) accv[pm][e] = (R)0;
        if (pm < cj.npar) {
            if constexpr (RH) {
                _Float16 hv[kCpt];
                load_cells<_Float16, kCpt>(reinterpret_cast<const _Float16 *>(p.resp) +
                                               ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + local + (size_t)cj.filter[pm] * HW, n, hv);
#pragma unroll
                for (int e = 0; e < kCpt; ++e) accv[pm][e] = (R)(float)hv[e];
            } else {
                load_cells<R, kCpt>(resp + (size_t)cj.filter[pm] * HW, n, accv[pm]);
            }
        }
    }
    // the job tables and biases are read-only and wave-uniform: reading them through the constant address
    // space keeps them on the scalar unit (a plain global load after the first store would be a vector load
    // the compiler has to wait for at every use)
    const int __attribute__((address_space(4))) *childs = (const int __attribute__((address_space(4))) *)p.childs;
    constexpr int kChildInts = 3 + kMaxMix;
    static_assert(sizeof(ChildDesc) == kChildInts * sizeof(int), "ChildDesc is read as 3 + kMaxMix ints");
    const float __attribute__((address_space(4))) *biasw = (const float __attribute__((address_space(4))) *)p.biasw;
    for (int ch = cj.child_begin; ch < cj.child_end; ++ch) {
        ChildDesc cd;
        {
            const int __attribute__((address_space(4))) *ci = childs + (size_t)ch * kChildInts;
            cd.job_begin = ci[0]; cd.nmix = ci[1]; cd.slot = ci[2];
#pragma unroll
            for (int k = 0; k < MAXM; ++k) cd.bias_off[k] = ci[3 + k];
        }
        const size_t gbase = gbase0 + (size_t)cd.job_begin * HW;
        float bw[MAXM][MAXM]; // bias(mm)[pm]
#pragma unroll
        for (int mm = 0; mm < MAXM; ++mm)
#pragma unroll
            for (int pm = 0; pm < MAXM; ++pm)
                bw[mm][pm] = (mm < cd.nmix && pm < cj.npar) ? biasw[cd.bias_off[mm] + pm] : 0.0f;
        R dtv[MAXM][kCpt];
#pragma unroll
        for (int mm = 0; mm < MAXM; ++mm) {
#pragma unroll
            // a mixture the child does not have scores -inf: it can never win the strict `>` below, so the selection loop
            // needs no `mm < nmix` test (a uniform branch per candidate, i.e. 144 basic blocks per child with their copies)
            for (int e = 0; e < kCpt; ++e) dtv[mm][e] = -RealLimits<R>::inf();
            if (mm < cd.nmix) load_cells_wide<R, kCpt>(dtp + gbase + (size_t)mm * HW + local, dtv[mm]);
        }
#pragma unroll
        for (int pm = 0; pm < MAXM; ++pm) {
            if (pm < cj.npar) {
                // only the winning child mixture is recorded (Ik): the positions it points to stay in the transform's own planes
                // and are composed where they are needed (k_argmin_walk, pbd_dp_min's read-back)
                uint8_t oik[kCpt];
                R best[kCpt];
                int bi[kCpt];
#pragma unroll
                for (int e = 0; e < kCpt; ++e) bi[e] = 0;
                if (cd.nmix == 1) {     // K == 1 copies (Math::reduceMax)
#pragma unroll
                    for (int e = 0; e < kCpt; ++e) best[e] = dtv[0][e] + (R)bw[0][pm];
                } else {
#pragma unroll
                    for (int e = 0; e < kCpt; ++e) {
                        best[e] = -RealLimits<R>::inf();
#pragma unroll
                        for (int mm = 0; mm < MAXM; ++mm) {
                            const R wv = dtv[mm][e] + (R)bw[mm][pm];
                            const bool t = wv > best[e];
                            best[e] = t ? wv : best[e];
                            bi[e] = t ? mm : bi[e];
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < kCpt; ++e) {
                    oik[e] = (uint8_t)bi[e];
                    accv[pm][e] = accv[pm][e] + best[e];
                }
                const size_t o = pbase + (size_t)(cd.slot + pm) * HW;
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
        if (p.ptr8) {                                                                                                      \
            if (f64) PBD_LAUNCH((k_dp_combine<double, 2, M, false, uint8_t>), g2, dim3(256), 0, s, p);            \
            else if (p.resp_half) PBD_LAUNCH((k_dp_combine<float, 4, M, true, uint8_t>), g4, dim3(256), 0, s, p); \
            else PBD_LAUNCH((k_dp_combine<float, 4, M, false, uint8_t>), g4, dim3(256), 0, s, p);                 \
        } else {                                                                                                           \
            if (f64) PBD_LAUNCH((k_dp_combine<double, 2, M, false, int16_t>), g2, dim3(256), 0, s, p);            \
            else if (p.resp_half) PBD_LAUNCH((k_dp_combine<float, 4, M, true, int16_t>), g4, dim3(256), 0, s, p); \
            else PBD_LAUNCH((k_dp_combine<float, 4, M, false, int16_t>), g4, dim3(256), 0, s, p);                 \
        }                                                                                                                  \
    } while (0)
    if (p.max_mix <= 2) PBD_COMBINE(2);
    else if (p.max_mix <= 4) PBD_COMBINE(4);
    else if (p.max_mix <= 6) PBD_COMBINE(6);
    else if (p.max_mix <= 8) PBD_COMBINE(8);
    else PBD_COMBINE(16);
#undef PBD_COMBINE
}

// ---- combine, sequential schedule: thread = one cell, block.y = one (component, part) of the step -------------
// For every parent mixture pm, in order: weighted[mm] = score_dt[mm] + bias(mm)[pm]; reduceMax; winning mixture -> Ik;
// then `parent.score[pm] += max` IN PLACE on the accumulator keyed by the parent mixture's filter id,
// which starts as a copy of the raw response the first time it is touched (src/DynamicProgram.cpp:134-156).
template <typename R, typename PT>
__global__ __launch_bounds__(256) void k_dp_combine_seq(DpParams p)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.cell_per_frame) return;
    const int fl = blockIdx.z, frame = p.frame0 + fl;
    const SeqCombineJob sj = p.sjobs[blockIdx.y];
    int lo = 0, hi = p.nlevels;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (p.lv[mid].cell_off <= idx) lo = mid; else hi = mid; }
    const LevelDesc d = p.lv[lo];
    const int W = d.cols;
    const int local = (int)(idx - d.cell_off);
    const size_t HW = (size_t)d.rows * W;
    const size_t gbase = ((size_t)fl * p.cell_per_frame + d.cell_off) * p.JG + (size_t)sj.job_begin * HW;
    const size_t pbase = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NS + local;
    const R *respp = static_cast<const R *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + local;
    R *accp = static_cast<R *>(p.acc) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NM + local;
    const R *dtp = static_cast<const R *>(p.dt) + gbase + local;
    R dtv[kMaxMix];
    for (int mm = 0; mm < sj.nmix; ++mm) dtv[mm] = dtp[(size_t)mm * HW];
    for (int pm = 0; pm < sj.npar; ++pm) {
        R best;
        int bi = 0;
        if (sj.nmix == 1) {
            best = dtv[0] + (R)p.biasw[sj.bias_off[0] + pm];
        } else {
            best = -RealLimits<R>::inf();
            for (int mm = 0; mm < sj.nmix; ++mm) {
                const R wv = dtv[mm] + (R)p.biasw[sj.bias_off[mm] + pm];
                if (wv > best) { bi = mm; best = wv; }
            }
        }
        const size_t o = pbase + (size_t)(sj.slot + pm) * HW;
        p.Ik[o] = (uint8_t)bi;
        R *t = accp + (size_t)sj.target[pm] * HW;
        const R base = !sj.init[pm] ? *t
                       : (sizeof(R) == 4 && p.resp_half)
                             ? (R)(float)(reinterpret_cast<const _Float16 *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F + local)[(size_t)sj.filter[pm] * HW]
                             : respp[(size_t)sj.filter[pm] * HW];
        *t = base + best;
    }
}

void launch_dp_combine_seq(const DpParams &p, int nsjobs, int nframes, bool f64, hipStream_t s)
{
    if (nsjobs == 0 || p.cell_per_frame == 0) return;
    dim3 grid((unsigned)((p.cell_per_frame + 255) / 256), nsjobs, nframes);
    if (p.ptr8) {
        if (f64) PBD_LAUNCH((k_dp_combine_seq<double, uint8_t>), grid, dim3(256), 0, s, p);
        else PBD_LAUNCH((k_dp_combine_seq<float, uint8_t>), grid, dim3(256), 0, s, p);
    } else {
        if (f64) PBD_LAUNCH((k_dp_combine_seq<double, int16_t>), grid, dim3(256), 0, s, p);
        else PBD_LAUNCH((k_dp_combine_seq<float, int16_t>), grid, dim3(256), 0, s, p);
    }
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
    const R *accp = static_cast<const R *>(p.acc) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NM;
    const R *respp = static_cast<const R *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F;
    const _Float16 *resph = static_cast<const _Float16 *>(p.resp) + ((size_t)frame * p.cell_per_frame + d.cell_off) * p.F;
    const bool rh = sizeof(R) == 4 && p.resp_half;
    auto score = [&](int mm) -> R {
        const size_t o = (size_t)rj.plane[mm] * HW + local;
        return ((rj.from_acc >> mm) & 1) ? accp[o] : rh ? (R)(float)resph[o] : respp[o];
    };
    R best;
    int bi = 0;
    if (rj.nmix == 1) {
        best = score(0) + (R)rj.bias;
    } else {
        best = -RealLimits<R>::inf();
        for (int mm = 0; mm < rj.nmix; ++mm) {
            const R wv = score(mm) + (R)rj.bias;
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
    if (f64) PBD_LAUNCH(k_dp_root<double>, grid, dim3(256), 0, s, p);
    else PBD_LAUNCH(k_dp_root<float>, grid, dim3(256), 0, s, p);
}

// ---- argmin ------------------------------------------------------------------------------------
// find: rootv > thresh (strict, src/DynamicProgram.cpp:208), in RASTER ORDER per (frame, level, component) as the
// reference's Math::find (include/Math.hpp:84-93) lists them -- rootv is laid out [frame][level][component][y][x], so the
// order (frame, level, component, y, x) the ABI promises IS the order of the element index.  Three small launches instead
// of an atomic append + a host sort: hits per block of 1024 elements, an exclusive scan of the block counts (which also
// leaves the TRUE number found in word 0 of the payload), then every hit writes its record header at
// (block offset + rank inside the block).  The candidate list leaves the device already ordered, with its count in front:
// one D2H, no host pass, and the same buffer is what a multi-GPU job hands to the collective (dist.CandidateGatherer).
constexpr int kFindEPT = 4, kFindBlock = 256, kFindSpan = kFindEPT * kFindBlock;

template <typename R>
__device__ __forceinline__ int find_hits(const ArgminParams &p, long long base, bool hit[kFindEPT])
{
    const R *rv = static_cast<const R *>(p.rootv);
    int c = 0;
#pragma unroll
    for (int e = 0; e < kFindEPT; ++e) {
        const long long o = base + e;
        hit[e] = o < p.ntotal && rv[o] > (R)p.thresh;       // NaN compares false, as in the reference's `rootv > thresh` mask
        c += hit[e] ? 1 : 0;
    }
    return c;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

template <typename R>
__global__ __launch_bounds__(kFindBlock) void k_argmin_count(ArgminParams p)
{
    bool hit[kFindEPT];
    const int c = find_hits<R>(p, (long long)blockIdx.x * kFindSpan + threadIdx.x * kFindEPT, hit);
    __shared__ int wsum[kFindBlock / 64];
    const int lane = threadIdx.x & 63, incl = wave_incl_scan(c, lane);
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < kFindBlock / 64; ++w) t += wsum[w];
        p.blk[blockIdx.x] = t;
    }
}

// one workgroup: blk[b] <- sum of blk[0..b) ; payload[0] <- total
__global__ __launch_bounds__(1024) void k_argmin_scan(ArgminParams p)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b0 = 0; b0 < p.nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const int v = b < p.nblk ? p.blk[b] : 0;
        const int incl = wave_incl_scan(v, lane);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int off = carry_s;
        for (int k = 0; k < w; ++k) off += wsum[k];
        if (b < p.nblk) p.blk[b] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) p.payload[0] = carry_s;
}

template <typename R>
__global__ __launch_bounds__(kFindBlock) void k_argmin_emit(ArgminParams p)
{
    bool hit[kFindEPT];
    const long long base = (long long)blockIdx.x * kFindSpan + threadIdx.x * kFindEPT;
    const int c = find_hits<R>(p, base, hit);
    __shared__ int wsum[kFindBlock / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, incl = wave_incl_scan(c, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    if (c == 0) return;
    int slot = p.blk[blockIdx.x] + incl - c;
    for (int k = 0; k < w; ++k) slot += wsum[k];
    const long long per_frame = p.cell_per_frame * p.NC;
#pragma unroll
    for (int e = 0; e < kFindEPT; ++e) {
        if (!hit[e]) continue;
        const int mine = slot++;
        if (mine >= p.capacity) continue;                      // word 0 still carries the true count: the caller sees the overflow
        const long long o = base + e;
        const int frame = (int)(o / per_frame);
        const long long rem = o - (long long)frame * per_frame;
        int lo = 0, hi = p.nlevels;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (p.lv[mid].cell_off * p.NC <= rem) lo = mid; else hi = mid; }
        const LevelDesc d = p.lv[lo];
        const int HWi = d.rows * d.cols;
        const int rem2 = (int)(rem - d.cell_off * p.NC);
        const int comp = rem2 / HWi, local = rem2 - comp * HWi;
        int32_t *rec = p.payload + 1 + (size_t)mine * p.stride;
        rec[0] = frame; rec[1] = comp; rec[2] = lo;
        rec[3] = local % d.cols; rec[4] = local / d.cols;
        rec[5] = __float_as_int((float)static_cast<const R *>(p.rootv)[o]);   // Candidate::confidence_ is float for every T (include/Candidate.hpp:72)
        rec[6] = 0;
        rec[7] = p.rooti[o];   // root mixture, consumed by the walk kernel
    }
}

void launch_argmin_find(const ArgminParams &p, bool f64, hipStream_t s)
{   // p.nblk = ceil(p.ntotal / kFindSpan) >= 1; p.blk holds nblk ints
    if (f64) PBD_LAUNCH(k_argmin_count<double>, dim3(p.nblk), dim3(kFindBlock), 0, s, p);
    else PBD_LAUNCH(k_argmin_count<float>, dim3(p.nblk), dim3(kFindBlock), 0, s, p);
    PBD_LAUNCH(k_argmin_scan, dim3(1), dim3(1024), 0, s, p);
    if (f64) PBD_LAUNCH(k_argmin_emit<double>, dim3(p.nblk), dim3(kFindBlock), 0, s, p);
    else PBD_LAUNCH(k_argmin_emit<float>, dim3(p.nblk), dim3(kFindBlock), 0, s, p);
}
int argmin_find_span() { return kFindSpan; }

template <typename R> __device__ __forceinline__ int round_mul(int a, R s);
// cv::Point_<int> * T -> saturate_cast<int>(a*s) = cvRound: round half to even
template <> __device__ __forceinline__ int round_mul<float>(int a, float s) { return __float2int_rn((float)a * s); }
template <> __device__ __forceinline__ int round_mul<double>(int a, double s) { return __double2int_rn((double)a * s); }

// walk: one thread per candidate follows Ix/Iy/Ik from the root (src/DynamicProgram.cpp:218-244).  The number of
// candidates is read from word 0 of the payload (the host never needs it to launch this); the grid strides over
// min(found, capacity) records.  The positions and mixtures of the parts already visited sit in LDS (x | y << 16 and the
// mixture, one column per thread: 51 KB for up to kWalkMaxParts = 160 parts) -- not in per-thread scratch.
template <typename R, typename PT>
__global__ __launch_bounds__(64) void k_argmin_walk(ArgminParams p)
{
    __shared__ int visited[kWalkMaxParts][64];
    __shared__ uint8_t visited_m[kWalkMaxParts][64];
    const int ncand = min(p.payload[0], p.capacity);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ncand; i += gridDim.x * blockDim.x) {
    int32_t *rec = p.payload + 1 + (size_t)i * p.stride;
    const int frame = rec[0], c = rec[1], l = rec[2];
    const LevelDesc d = p.lv[l];
    const int W = d.cols;
    const size_t HW = (size_t)d.rows * W;
    const R scale = (R)p.scales[l];   // T scale = scales[n] (vectorf), src/DynamicProgram.cpp:199
    const PartWalk *walk = p.walk + p.walk_off[c];
    const int nparts = p.walk_off[c + 1] - p.walk_off[c];
    const size_t pbase = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NS;
    const size_t jbase = ((size_t)frame * p.cell_per_frame + d.cell_off) * p.NJ;
    int32_t *rects = rec + 8;
    for (int pidx = 0; pidx < nparts; ++pidx) {
        int x, y, m;
        if (pidx == 0) {
            x = rec[3]; y = rec[4]; m = rec[7];
        } else {
            const PartWalk w = walk[pidx];
            const int par = visited[w.parent][threadIdx.x];
            const int px = par & 0xffff, py = par >> 16, pm = visited_m[w.parent][threadIdx.x];
            // Ix = IxRaw[k][py][px], Iy = IyRaw[k][py][Ix] with k = the winning mixture (the reference's composition)
            m = p.Ik[pbase + (size_t)(w.slot + pm) * HW + (size_t)py * W + px];
            const size_t jo = jbase + (size_t)(w.mix0 + m) * HW;
            x = static_cast<const PT *>(p.IxRaw)[jo + (size_t)px * d.rows + py];      // IxRaw is kept transposed ([x][y])
            y = static_cast<const PT *>(p.IyRaw)[jo + (size_t)py * W + x];
        }
        visited[pidx][threadIdx.x] = x | (y << 16);
        visited_m[pidx][threadIdx.x] = (uint8_t)m;
        const int ks = walk[pidx].ksize[m];
        const int x1 = round_mul<R>(x - 1, scale), y1 = round_mul<R>(y - 1, scale);
        const int x2 = x1 + round_mul<R>(ks, scale) - 1, y2 = y1 + round_mul<R>(ks, scale) - 1;
        const int rx = min(x1, x2), ry = min(y1, y2);
        rects[pidx * 4 + 0] = rx;
        rects[pidx * 4 + 1] = ry;
        rects[pidx * 4 + 2] = max(x1, x2) - rx;
        rects[pidx * 4 + 3] = max(y1, y2) - ry;
    }
    rec[0] = frame + p.frame_offset;   // index within the batch -> global frame id of a sharded job (0 on one GPU)
    rec[6] = nparts;
    rec[7] = 0;
    }
}

void launch_argmin_walk(const ArgminParams &p, bool f64, hipStream_t s)
{
    const int blocks = std::max(std::min((p.capacity + 63) / 64, 2048), 1);
    if (p.ptr8) {
        if (f64) PBD_LAUNCH((k_argmin_walk<double, uint8_t>), dim3(blocks), dim3(64), 0, s, p);
        else PBD_LAUNCH((k_argmin_walk<float, uint8_t>), dim3(blocks), dim3(64), 0, s, p);
    } else {
        if (f64) PBD_LAUNCH((k_argmin_walk<double, int16_t>), dim3(blocks), dim3(64), 0, s, p);
        else PBD_LAUNCH((k_argmin_walk<float, int16_t>), dim3(blocks), dim3(64), 0, s, p);
    }
}

}  // namespace pbd

#ifdef PBD_DT_STATS
extern "C" int pbd_debug_dt_stats(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(pbd::g_dt_stats), sizeof(pbd::g_dt_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pbd::g_dt_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
