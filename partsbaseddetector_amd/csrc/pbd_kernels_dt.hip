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
template <typename R>
__device__ __forceinline__ R dt_val(double a, double b, int x, R y)
{   // Quadratic::operator()(x, y), :103-105
    return (R)((a * (double)(x * x) + b * (double)x) + (double)y);
}
template <typename R> __device__ __forceinline__ R neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

constexpr int kBuildCH = 8;     // source elements fetched from LDS ahead of their use
constexpr int kOutCH = 16;      // outputs leaving together (one 32-byte pointer store per lane in the rows pass)

typedef short v8s_u __attribute__((ext_vector_type(8), aligned(2)));

// per-lane envelope state
template <typename R>
struct Envelope {
    const R *row;                 // this problem's staged source row (LDS)
    unsigned *mask;               // this lane's column of the [word][64] survivor mask (LDS)
    double a, b;
    int vk; R sk, zk;             // top entry
    int vb; R sb;                 // entry below the top (vb == -1: none)
    unsigned cw; int cwi;         // survivors strictly below `vb`: cached word and its index; LDS words above cwi are 0

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
    __device__ __forceinline__ int take_highest()
    {   // the set is not empty
        while (cw == 0) {
            mask[cwi * 64] = 0;
            --cwi;
            cw = mask[cwi * 64];
        }
        const int hb = 31 - __clz((int)cw);
        cw &= ~(1u << hb);
        return (cwi << 5) + hb;
    }
    __device__ __forceinline__ void pop()
    {   // requires vk != 0, i.e. an entry below exists
        vk = vb; sk = sb;
        if (vb == 0) {
            zk = neg_inf<R>(); vb = -1;
        } else {
            const int pp = take_highest();
            const R sp = row[pp];
            zk = dt_isect<R>(a, b, pp, vk, sp, sk);   // as computed when entry vk was pushed onto entry pp
            vb = pp; sb = sp;
        }
    }
    __device__ __forceinline__ void step(int q, R sq)
    {   // include/DistanceTransform.hpp:158-169
        R s = dt_isect<R>(a, b, vk, q, sk, sq);
        while (s <= zk && vk != 0) {
            pop();
            s = dt_isect<R>(a, b, vk, q, sk, sq);
        }
        if (vb >= 0) add_below(vb);
        vb = vk; sb = sk;
        vk = q; sk = sq; zk = s;
    }
};

}  // namespace

// ROWS: problem = row y of a score plane, N = W; value out transposed tmp[x][y], pointer out row-major Ix[y][x].
// !ROWS: problem = column x (contiguous in the transposed tmp), N = H; value out dt[y][x], pointer out IyRaw[y][x].
template <typename R, bool ROWS>
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

    // ---- LDS: [rw][pitch] source rows | 64 row pointers | [nw][64] survivor-mask words ----
    R *srcL = reinterpret_cast<R *>(smem);
    const R **tab = reinterpret_cast<const R **>(smem + (size_t)A.rw * A.pitch * sizeof(R));
    unsigned *maskL = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(tab) + 64 * sizeof(const R *));
    tab[lane] = src;
    for (int w = 0; w < A.nw; ++w) maskL[w * 64 + lane] = 0;
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

    Envelope<R> env;
    env.row = srcL + (size_t)(act ? lane : lane % nact) * A.pitch;
    env.mask = maskL + lane;
    env.a = ROWS ? job.ax : job.ay;
    env.b = ROWS ? job.bx : job.by;
    const int os0 = ROWS ? job.osx : job.osy;
    env.vk = 0; env.sk = env.row[0]; env.zk = neg_inf<R>();
    env.vb = -1; env.sb = (R)0;
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
                out[i] = dt_val<R>(env.a, env.b, os0 + q - env.vk, env.sk);
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

template <typename R, bool ROWS>
static void launch_one(const DpParams &p, const DtPassArgs &a, size_t lds_bytes, hipStream_t s)
{
    static bool attr_set = false;       // dynamic LDS above 64 KB has to be allowed once per kernel
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dt_pass<R, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_dt_pass<R, ROWS>), dim3(a.wbegin[a.n]), dim3(64), lds_bytes, s, p, a);
}

void launch_dt_pass(const DpParams &p, const DtPassArgs &a, size_t lds_bytes, bool rows, bool f64, hipStream_t s)
{
    if (a.n == 0 || a.wbegin[a.n] == 0) return;
    if (f64) { if (rows) launch_one<double, true>(p, a, lds_bytes, s); else launch_one<double, false>(p, a, lds_bytes, s); }
    else { if (rows) launch_one<float, true>(p, a, lds_bytes, s); else launch_one<float, false>(p, a, lds_bytes, s); }
}

}  // namespace pbd
