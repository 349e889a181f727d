// pbd_kernels_dt.hip -- the 1-D generalized distance transform passes (gfx950), on-chip formulation.
//
// Replaces DistanceTransform<T>::computeRow (reference include/DistanceTransform.hpp:152-182) as called by
// DistanceTransform<T>::compute (:203-245): a rows pass (fx, os.x) and a columns pass (fy, os.y) over every
// (frame, level, part-mixture) score plane of one tree-depth group.
//
// One lane = one 1-D problem (a row, or a column), one workgroup = one wave = up to 64 problems of the SAME
// pyramid level (consecutive rows of consecutive planes), so every lane walks the same number of elements in
// lockstep.  Nothing but the input and the outputs touches HBM:
//   * the wave's source rows are staged once into LDS with coalesced loads ([problem][q], odd pitch: a lockstep
//     read and the staging writes are conflict-free);
//   * the envelope stack is NOT stored.  Which parabolas are on the envelope is a bit mask per problem (the stack is
//     ordered by position, so "the entry below" = the next lower set bit); src[v] of an entry is re-read from the
//     staged row; z of an entry is the intersection with the entry below it, recomputed -- with the same expression
//     on the same operands as when it was pushed, hence bit-identical -- when a pop makes it the top.  Registers
//     hold the top entry (v, src, z) and the entry below it (v, src); the mask of the entries further down lives in
//     LDS ([word][lane]) with the current word cached in a register;
//   * the read-out walks q downwards and pops (z[1..ktop] is strictly increasing, so
//     "k = 0; while (z[k+1] < os) k++" of :172-178 selects the same k(q) as "k = ktop; while (!(z[k] < os)) k--").
// The arithmetic per element is exactly computeRow's: the double-precision intersection rounded once to T, the
// `s <= z[k] && k > 0` pop rule (k > 0 <=> the top is not position 0, which is never popped), the T-typed
// `z[k+1] < os` read-out.  Compiled with -ffp-contract=off.
//
// Levels are launched in classes of similar length so that the dynamic LDS of a launch (64 x longest row of the
// class) does not cap the occupancy of the short levels; very long rows use fewer problems per wave.
#include "pbd_internal.h"

#include <math.h>

namespace pbd {

namespace {

template <typename R>
__device__ __forceinline__ R dt_isect(double a, double b, int x0, int x1, R y0f, R y1f)
{   // Quadratic::operator()(x0, x1, y0, y1), include/DistanceTransform.hpp:98-100, rounded to T
    const double y0 = (double)y0f, y1 = (double)y1f;
    const double num = ((y1 - y0) - b * (double)(x1 - x0)) + a * (double)(x1 * x1 - x0 * x0);
    return (R)(num / ((2 * a) * (double)(x1 - x0)));
}
template <typename R, bool BZERO>
__device__ __forceinline__ R dt_val(double a, double b, int x, R y)
{   // Quadratic::operator()(x, y), :103-105
    if (BZERO) return (R)(a * (double)(x * x) + (double)y);     // b == -0.0: t + (-0.0) == t for every t
    return (R)((a * (double)(x * x) + b * (double)x) + (double)y);
}
// entries of the 1/d table, padded so that what follows stays 16-byte aligned
__host__ __device__ inline size_t dt_invd_entries(int pitch) { return (size_t)((pitch + 1) & ~1); }

template <typename R> __device__ __forceinline__ R neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

constexpr int kBuildCH = 8;     // source elements fetched from LDS ahead of their use
constexpr int kOutCH = 16;      // outputs leaving together (one 32-byte pointer store per lane in the rows pass)
constexpr int kRing = 8;        // z of the entries just below the top (power of two)

typedef short v8s_u __attribute__((ext_vector_type(8), aligned(2)));

// The intersection for T = float without the fp64 divide.  The reference value is RN_float(RN_double(num / den)),
// den = (2a) * d exactly (a is a float widened to double, d < 2^13).  qt = num * (RN(1/(2a)) * RN(1/d)) differs
// from RN_double(num / den) by at most 6 double ulps (three roundings of 2^-53 in the reciprocal, one in the
// product, half an ulp in RN_double), so (float)qt is the reference value unless qt lies within 6 ulps of a
// midpoint between two floats -- bits 0..28 of the double significand within 6 of 2^28 -- or outside the normal
// float range (where the float grid is different) or not finite.  Those lanes (about one in 2^22) take the divide.
struct FastIsect {
    double a, b, inv2a, den2a;
    const double *invd;           // LDS: invd[d] = RN(1.0 / d)
};
template <bool BZERO>
__device__ __forceinline__ float isect_f32(const FastIsect &f, int x0, int x1, float y0f, float y1f)
{
    const int d = x1 - x0;
    const double dd = (double)d;
    double num = (double)y1f - (double)y0f;
    if (!BZERO) num = num - f.b * dd;          // b == -0.0: (y1 - y0) - (-0.0) leaves every value but -0.0 unchanged, and the sum below is never 0
    num = num + f.a * (double)(d * (x1 + x0)); // x1*x1 - x0*x0 in int, as the reference
    const double qt = num * (f.inv2a * f.invd[d]);
    const unsigned lo = (unsigned)__double2loint(qt), hi = (unsigned)__double2hiint(qt);
    const unsigned dist = (lo & 0x1FFFFFFFu) - (0x10000000u - 8u);          // < 17  <=>  within 8 of the midpoint
    const unsigned ex = ((hi >> 20) & 0x7FFu) - (1023u - 120u);              // < 241 <=>  2^-120 <= |qt| < 2^121
    float s = (float)qt;
    if (__builtin_expect(dist < 17u || ex >= 241u, 0)) s = (float)(num / (f.den2a * dd));
    return s;
}

// per-lane envelope state
template <typename R, bool BZERO>
struct Envelope {
    const R *row;                 // this problem's staged source row (LDS)
    unsigned *mask;               // this lane's column of the [word][64] survivor mask (LDS)
    R *ring;                      // this lane's column of the [kRing][64] z ring (LDS)
    double a, b;
    FastIsect fi;
    int vk; R sk, zk;             // top entry
    int k, lo;                    // index of the top entry in the stack; the ring holds z of the entries [lo, k)
    int pb; R sb;                 // entry directly below the top when known from the last pop (pb < 0: not known)
    unsigned cw; int cwi;         // survivors below the top: cached mask word and its index; LDS words above cwi are 0

    __device__ __forceinline__ R isect(int x0, int x1, R y0, R y1) const
    {
        if constexpr (sizeof(R) == 4) return isect_f32<BZERO>(fi, x0, x1, y0, y1);
        else return dt_isect<R>(a, b, x0, x1, y0, y1);
    }
    __device__ __forceinline__ void add_below(int p)
    {   // p is higher than every member
        const int w = p >> 5;
        const unsigned bit = 1u << (p & 31);
        if (w == cwi) {
            cw |= bit;
        } else {
            mask[cwi * 64] = cw;
            cwi = w; cw = bit;
        }
    }
    __device__ __forceinline__ void settle()
    {   // move the cursor down to the highest non-empty word (the set is not empty)
        while (cw == 0) {
            mask[cwi * 64] = 0;
            --cwi;
            cw = mask[cwi * 64];
        }
    }
    __device__ __forceinline__ void pop()
    {   // requires vk != 0: the set of survivors below the top is not empty
        settle();
        const int hb = 31 - __clz((int)cw);
        const int pnew = (cwi << 5) + hb;
        cw &= ~(1u << hb);
        sk = (pb == pnew) ? sb : row[pnew];
        vk = pnew;
        pb = -1;
        --k;
        if (pnew == 0) {
            zk = neg_inf<R>();
        } else if (k >= lo) {
            zk = ring[(k & (kRing - 1)) * 64];
        } else {
            // z of the new top = its intersection with the entry below it, as computed when it was pushed
            settle();
            const int p2 = (cwi << 5) + 31 - __clz((int)cw);
            const R s2 = row[p2];
            zk = isect(p2, vk, s2, sk);
            pb = p2; sb = s2;
            lo = k;
        }
    }
    __device__ __forceinline__ void step(int q, R sq)
    {   // include/DistanceTransform.hpp:158-169
        R s = isect(vk, q, sk, sq);
        while (s <= zk && vk != 0) {
            pop();
            s = isect(vk, q, sk, sq);
        }
        add_below(vk);
        ring[(k & (kRing - 1)) * 64] = zk;
        ++k;
        if (k - lo > kRing) lo = k - kRing;
        pb = vk; sb = sk;
        vk = q; sk = sq; zk = s;
    }
};

}  // namespace

// ROWS: problem = row y of a score plane, N = W; value out transposed tmp[x][y], pointer out row-major Ix[y][x].
// !ROWS: problem = column x (contiguous in the transposed tmp), N = H; value out dt[y][x], pointer out IyRaw[y][x].
template <typename R, bool ROWS, bool BZERO>
__global__ __launch_bounds__(64) void k_dt_pass(DpParams p, DtPassArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    // ---- which level, which problems (wave-uniform): binary search in the by-value class table ----
    int lo = 0, hi = A.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.wbegin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    const int lev = A.level[lo];
    const int H = p.lv[lev].rows, W = p.lv[lev].cols;
    const long long cell_off = p.lv[lev].cell_off;
    const int L = ROWS ? H : W, N = ROWS ? W : H;
    const size_t HW = (size_t)H * W;
    const long long total = (long long)A.nplanes * L;
    const long long g0 = (long long)((int)blockIdx.x - A.wbegin[lo]) * A.rw;
    const int nact = (int)min((long long)A.rw, total - g0);           // >= 1 by construction of the table
    const bool act = lane < nact;
    const long long g = g0 + (act ? lane : lane % nact);              // idle lanes shadow an active problem (same trip counts)
    const int pi = (int)(g / L), r = (int)(g - (long long)pi * L);
    const int fl = pi / p.JG, j = pi - fl * p.JG;
    const DtJob job = p.jobs[j];
    const size_t jbase = ((size_t)fl * p.cell_per_frame + cell_off) * p.JG + (size_t)j * HW;
    const R *src;
    if (ROWS) {
        const size_t fbase = (size_t)(p.frame0 + fl) * p.cell_per_frame + cell_off;
        src = (job.from_acc ? static_cast<const R *>(p.acc) + fbase * p.NM : static_cast<const R *>(p.resp) + fbase * p.F) +
              (size_t)job.plane * HW + (size_t)r * N;
    } else {
        src = static_cast<const R *>(p.tmp) + jbase + (size_t)r * N;
    }
    R *vout = static_cast<R *>(ROWS ? p.tmp : p.dt) + jbase + r;                         // element q at vout[q * L]
    int16_t *pout = ROWS ? p.IxRaw + jbase + (size_t)r * N : p.IyRaw + jbase + r;       // ROWS: pout[q], else pout[q * L]

    // ---- LDS: 1/d table (doubles) | 64 row pointers | [kRing][64] z ring | [nw][64] survivor-mask words | [rw][pitch] source rows
    double *invd = reinterpret_cast<double *>(smem);
    const R **tab = reinterpret_cast<const R **>(invd + dt_invd_entries(A.pitch));
    R *ringL = reinterpret_cast<R *>(tab + 64);
    unsigned *maskL = reinterpret_cast<unsigned *>(ringL + kRing * 64);
    R *srcL = reinterpret_cast<R *>(maskL + (size_t)A.nw * 64);
    tab[lane] = src;
    for (int w = 0; w < A.nw; ++w) maskL[w * 64 + lane] = 0;
    if (sizeof(R) == 4)
        for (int d = lane; d < N; d += 64) invd[d] = 1.0 / (double)max(d, 1);
    __syncthreads();
    {   // staging: element e of the wave's nact x N block -> (problem e / N, q = e % N); consecutive lanes read consecutive q
        const unsigned magic = N > 1 ? 0xFFFFFFFFu / (unsigned)N + 1u : 0u;   // e / N = umulhi(e, magic) for e * N < 2^32
        const int nel = nact * N;
        constexpr int U = 4;
        for (int e0 = 0; e0 < nel; e0 += 64 * U) {
            R val[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u * 64 + lane;
                const int ec = min(e, nel - 1);
                const int rr = N > 1 ? (int)__umulhi((unsigned)ec, magic) : ec;
                const int q = ec - rr * N;
                val[u] = tab[rr][q];
                dst[u] = e < nel ? rr * A.pitch + q : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) srcL[dst[u]] = val[u];
        }
    }
    __syncthreads();

    Envelope<R, BZERO> env;
    env.row = srcL + (size_t)(act ? lane : lane % nact) * A.pitch;
    env.mask = maskL + lane;
    env.ring = ringL + lane;
    env.a = ROWS ? job.ax : job.ay;
    env.b = ROWS ? job.bx : job.by;
    env.fi.a = env.a; env.fi.b = env.b; env.fi.den2a = 2 * env.a; env.fi.inv2a = 1.0 / (2 * env.a); env.fi.invd = invd;
    const int os0 = ROWS ? job.osx : job.osy;
    env.vk = 0; env.sk = env.row[0]; env.zk = neg_inf<R>();
    env.k = 0; env.lo = 0; env.pb = -1; env.sb = (R)0;
    env.cw = 0; env.cwi = 0;
    // ---- build the envelope, q ascending in lockstep ----
    for (int q0 = 1; q0 < N; q0 += kBuildCH) {
        R cur[kBuildCH];
#pragma unroll
        for (int i = 0; i < kBuildCH; ++i) cur[i] = env.row[min(q0 + i, N - 1)];
#pragma unroll
        for (int i = 0; i < kBuildCH; ++i)
            if (q0 + i < N) env.step(q0 + i, cur[i]);
    }
    // ---- read-out, q descending ----
    const int nch = (N + kOutCH - 1) / kOutCH;
    for (int c = nch - 1; c >= 0; --c) {
        const int q0 = c * kOutCH;
        R out[kOutCH];
        int ptr[kOutCH];
#pragma unroll
        for (int i = kOutCH - 1; i >= 0; --i) {
            out[i] = (R)0; ptr[i] = 0;
            const int q = q0 + i;
            if (q < N) {
                const R osf = (R)(os0 + q);
                while (!(env.zk < osf)) env.pop();      // z of position 0 is -inf: the walk ends there
                out[i] = dt_val<R, BZERO>(env.a, env.b, os0 + q - env.vk, env.sk);
                ptr[i] = env.vk;
            }
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < kOutCH; ++i)
                if (q0 + i < N) vout[(size_t)(q0 + i) * L] = out[i];
            if (ROWS) {
                if (q0 + kOutCH <= N) {
#pragma unroll
                    for (int v = 0; v < kOutCH / 8; ++v) {
                        v8s_u t;
#pragma unroll
                        for (int e = 0; e < 8; ++e) t[e] = (short)ptr[8 * v + e];
                        *reinterpret_cast<v8s_u *>(pout + q0 + 8 * v) = t;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < kOutCH; ++i)
                        if (q0 + i < N) pout[q0 + i] = (int16_t)ptr[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < kOutCH; ++i)
                    if (q0 + i < N) pout[(size_t)(q0 + i) * L] = (int16_t)ptr[i];
            }
        }
    }
}

template <typename R, bool ROWS, bool BZERO>
static void launch_one(const DpParams &p, const DtPassArgs &a, size_t lds_bytes, hipStream_t s)
{
    static bool attr_set = false;       // dynamic LDS above 64 KB has to be allowed once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dt_pass<R, ROWS, BZERO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_dt_pass<R, ROWS, BZERO>), dim3(a.wbegin[a.n]), dim3(64), lds_bytes, s, p, a);
}

size_t dt_pass_lds_bytes(int n, int rw, size_t rs)
{
    return dt_invd_entries(n | 1) * sizeof(double) + 64 * sizeof(void *) + (size_t)kRing * 64 * rs + (size_t)((n + 31) / 32) * 256 +
           (size_t)rw * (size_t)(n | 1) * rs;
}

// bzero: the linear deformation term of every job of the group is exactly -0.0 in this pass's direction
void launch_dt_pass(const DpParams &p, const DtPassArgs &a, size_t lds_bytes, bool rows, bool bzero, bool f64, hipStream_t s)
{
    if (a.n == 0 || a.wbegin[a.n] == 0) return;
#define PBD_DT_GO(R, RW)                                                            \
    do {                                                                            \
        if (bzero) launch_one<R, RW, true>(p, a, lds_bytes, s);                     \
        else launch_one<R, RW, false>(p, a, lds_bytes, s);                          \
    } while (0)
    if (f64) { if (rows) PBD_DT_GO(double, true); else PBD_DT_GO(double, false); }
    else { if (rows) PBD_DT_GO(float, true); else PBD_DT_GO(float, false); }
#undef PBD_DT_GO
}

}  // namespace pbd
