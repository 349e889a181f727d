// pbd_capi.hip -- handle, plans, device workspace and the C entry points of include/pbd.h.
//
// Host logic restated from the reference for this path:
//   pyramid geometry            src/HOGFeatures.cpp:95-127, include/HOGFeatures.hpp:74-81
//   engine wiring               src/PartsBasedDetector.cpp:69-127
//   Parts index tables          include/Parts.hpp:172-187
// There is no CPU compute path: every stage runs as a HIP kernel (pbd_kernels_*.hip).
#include "pbd_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <numeric>
#include <new>
#include <set>
#include <stdexcept>

using namespace pbd;

namespace pbd {
thread_local ProfHook *g_prof_hook = nullptr;
}

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

template <typename T>
struct DevTable {   // small immutable table uploaded once
    T *d = nullptr;
    size_t n = 0;
    hipError_t upload(const std::vector<T> &h)
    {
        release();
        n = h.size();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), n * sizeof(T));
        if (e != hipSuccess) return e;
        return hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() { if (d) (void)hipFree(d); d = nullptr; n = 0; }
};

// ---- pyramid geometry (host) -------------------------------------------------------------------
// Overload resolution assumed for the reference's expressions: C++11 <cmath>, i.e. pow(float,float)
// and log(float) are the float versions, pow(float,int) promotes to double.
int plan_pyramid(int rows, int cols, int sbin, int interval, std::vector<int> &lr, std::vector<int> &lc,
                 std::vector<float> &scales)
{
    const float sfactor = powf(2.0f, 1.0f / (float)interval);             // HOGFeatures.hpp:78
    const float h = (float)rows, w = (float)cols;
    const float mn = std::min(h, w);
    const float ns = 1 + floorf(logf(mn / (5.0f * (float)sbin)) / logf(sfactor));   // HOGFeatures.cpp:99
    if (!(ns >= 1)) return 0;
    const int n = (int)ns;
    if (n > PBD_MAX_LEVELS) return -1;
    if (n < interval) return -2;   // the reference writes out of bounds here (HOGFeatures.cpp:114-118)
    lr.assign(n, 0); lc.assign(n, 0); scales.assign(n, 0.f);
    for (int i = 0; i < interval; ++i) {
        const float f = (float)((double)1.0f / pow((double)sfactor, (double)i));     // :116
        lc[i] = (int)lrint((double)(w * f));   // Size_<float> -> Size: cvRound, half to even
        lr[i] = (int)lrint((double)(h * f));
        scales[i] = (float)(pow((double)sfactor, (double)i) * (double)sbin);          // :118
        for (int j = i + interval; j < n; j += interval) {                            // :120-126
            lc[j] = (lc[j - interval] + 1) / 2;
            lr[j] = (lr[j - interval] + 1) / 2;
            scales[j] = 2 * scales[j - interval];
        }
    }
    return n;
}

inline short sat_short_round(float v)
{
    long iv = lrint((double)v);
    return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv);
}

struct Plan {
    // key
    int kind = 0;   // 0: from image size, 1: from explicit feature-map sizes
    int rows = 0, cols = 0;         // the geometry does not depend on the channel count (offsets are in pixels)
    std::vector<int> key_dims;
    // geometry
    int nlevels = 0;
    std::vector<LevelDesc> lv;
    std::vector<float> scales;
    long long pix_per_frame = 0, blk_per_frame = 0, cell_per_frame = 0, npix_resized = 0, quad_per_frame = 0;
    int interval = 0;
    int nrows_flat = 0, ncols_flat = 0;
    bool ptr8 = false;              // no feature map side exceeds 256: positions fit uint8 (back-pointer planes at half the bytes)
    int longest = 0;                // longest side of any feature map of the plan (rows / columns of the distance transform)
    int ntiles = 0;
    // device tables
    DevTable<LevelDesc> d_lv;
    DevTable<ResizeTabX> d_tabx;
    DevTable<ResizeTabY> d_taby;
    DevTable<ResizeTabXf> d_tabxf;   // the same mapping with float coefficients (16U / 32F / 64F images)
    DevTable<ResizeTabYf> d_tabyf;
    DevTable<ConvTile> d_tiles, d_shaped, d_htiles;
    int nshaped[3] = {0, 0, 0}, nhtiles = 0;
    // strip-sequence tiles of the exact 5 x 5 convolution, per number of frames in a launch (built on first use)
    std::map<int, DevTable<ConvSegTile>> segtiles;
    DevTable<int> d_row2level, d_rowoff, d_col2level, d_coloff;
    DevTable<long long> d_stk_row_off, d_stk_col_off;
    long long stk_per_jf = 0;
    DevTable<float> d_scales;
    void release()
    {
        d_lv.release(); d_tabx.release(); d_taby.release(); d_tabxf.release(); d_tabyf.release(); d_tiles.release(); d_shaped.release(); d_htiles.release();
        for (auto &kv : segtiles) kv.second.release();
        segtiles.clear();
        d_row2level.release(); d_rowoff.release(); d_col2level.release(); d_coloff.release(); d_scales.release();
        d_stk_row_off.release(); d_stk_col_off.release();
    }
    ~Plan() { release(); }
};

struct Group {   // DT jobs of the parts of one tree depth + combine jobs of their parents
    std::vector<DtJob> jobs;
    std::vector<ChildDesc> childs;
    std::vector<CombineJob> cjobs;
    std::vector<SeqCombineJob> sjobs;     // sequential schedule (shared filter ids): replaces childs / cjobs
    DevTable<DtJob> d_jobs;
    int bz_x = 0, bz_y = 0;               // all jobs: linear coefficient exactly -0.0 and a != 0 (DpParams::bz_x / bz_y)
    DevTable<ChildDesc> d_childs;
    DevTable<CombineJob> d_cjobs;
    DevTable<SeqCombineJob> d_sjobs;
};

struct Prof {
    int on = 0;                      // 0: off, 1: every kernel, 2: the convolution only (pbd_profile_enable)
    struct Rec { int k; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double total[PBD_K_COUNT] = {0};
    int launches[PBD_K_COUNT] = {0};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void flush()
    {
        for (auto &r : recs) {
            (void)hipEventSynchronize(r.b);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total[r.k] += ms; launches[r.k] += 1; }
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
    void release()
    {
        flush();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

}  // namespace

struct pbd_handle {
    pbd_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;      // pyramid, HOG, convolution (and everything in the staged calls)
    hipStream_t stream2 = nullptr;     // dynamic program of the previous chunk, overlapped with the convolution
    std::vector<hipEvent_t> chunk_events;
    bool own_stream = false;

    // model (host copies)
    int NC = 0, F = 0, flen = 32, sbin = 4, interval = 10, norient = 18, NS = 0, NM = 0, max_parts = 0;
    float thresh = 0.f;
    int ksize = 0, Fpad = 0;
    std::vector<int> filter_ksize, part_offset, parentid, mix_offset, filterid, biasid, defid, ptr_slot, anchors;
    std::vector<float> biasw, defw;
    std::vector<Group> groups;       // deepest first
    std::vector<RootJob> rjobs;
    std::vector<PartWalk> walk;
    std::vector<int> walk_off;
    int JGmax = 0;
    int max_mix = 1;                 // largest number of mixtures of any part
    bool filters_set = false;
    bool seq_mode = false;           // a filter id occurs twice inside a component: sequential schedule, accumulators keyed by filter id
    bool bank_matches_model = true;  // false after a setFilters() whose bank no longer covers the model's filter ids

    // device model tables
    // convolution bank: the filters grouped by size (one class in every known model; the reference builds one engine
    // per filter and so takes any mix: src/SpatialConvolutionEngine.cpp:141-158)
    struct ConvClass {
        int K = 0, nf = 0, Fpad = 0;
        DevBuf wts;                  // real-typed weights of the class
        DevTable<int> fmap;          // class-local index -> filter id (empty when the class is the whole bank in order)
        // k_conv3 (float, 5 x 5): the class cut into units of 2..8 filters, weights [unit][32][25][8]
        DevBuf wts3;
        DevTable<int> unit_f0, unit_ql, unit_woff;
        int nunits = 0;
        DevTable<float> c31tab;      // [81][c31stride]: see pbd_kernels_conv.hip (channel 31)
        int c31stride = 0;
    };
    std::vector<ConvClass> conv_classes;
    DevBuf d_wrec;                   // bf16 hi/lo weight records of the matrix-core path
    DevTable<float> d_biasw;
    DevTable<int> d_walk_off;
    DevTable<RootJob> d_rjobs;
    DevTable<PartWalk> d_walk;
    DevBuf d_coord;                  // HogCoordT<R>[]
    int coord_n = 0;
    bool f64 = false;                // reference template parameter T = double
    size_t rs = sizeof(float);       // sizeof(T)
    bool resp_half = false;          // PBD_CONV_MFMA_F16: the responses live on the device as fp16 (BASELINE configs[4])
    size_t resp_es = sizeof(float);  // bytes per response element on the device

    // plans
    std::vector<std::unique_ptr<Plan>> plans;
    Plan *cur = nullptr;
    int cur_frames = 0, cur_cn = 3;
    int cur_depth = kDepth8U;        // image depth of the frames being processed (set by the entry point)
    int shard_rank = 0, shard_world = 1;   // level sharding of single frames over several GPUs (pbd_set_level_shard)
    bool have_features = false, have_resp = false, have_dp = false;
    bool feat_c31_zero = false;      // h->feat was written by the HOG kernels (channel 31 = 0), not uploaded by the caller

    // workspace
    DevBuf frames, pyr, gmag, gori, hist, norm, feat, resp, acc, Ik, rootv, rooti;
    int totmix = 0;                  // (part, mixture) pairs of the model = planes of IxRaw / IyRaw per cell block
    DevBuf tmp, dt, IxRaw, IyRaw, stk, scales_tmp, find_blk;

    // A candidate list on its way out.  The device side is the "payload" the find / walk kernels write: word 0 = roots
    // found, then the records, already in (frame, level, component, y, x) order.  The host side is a pinned mirror: the
    // count and the first `guess` records (what the previous batch needed + 25 %) are copied by ONE asynchronous D2H
    // enqueued right behind the walk kernel, so a steady stream of batches never waits for a count before it can ask for
    // the records; a batch that outgrows the guess costs one more copy.
    struct CandBuf {
        DevBuf payload;
        int32_t *host = nullptr; size_t host_words = 0;
        int copied = 0;                                   // records covered by the enqueued copy
    } cb;
    int cand_guess = 1024;                                // records the next speculative copy covers (shared by every CandBuf)

    // pipelined host entry points (pbd_detect_batch_submit / _wait): two batches may be in flight
    struct Slot {
        void *pinned = nullptr; size_t pinned_cap = 0;   // host staging of the frames (hipHostMalloc)
        DevBuf frames;
        CandBuf cb;
        hipEvent_t copied = nullptr, done = nullptr;
        Plan *plan = nullptr;
        int nframes = 0;
    } slot[2];
    hipStream_t stream_copy = nullptr, stream_d2h = nullptr;
    long long nsubmitted = 0, nwaited = 0;

    Prof prof;
};

namespace {

int fail(pbd_handle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try {
        if (h) h->err = buf; else g_create_error = buf;
    } catch (...) {   // the message itself could not be stored: the status code still goes out
    }
    return code;
}

#define HIPCHK(h, expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (void)hipGetLastError();   /* the error is reported through the status code, not left sticky */ \
            return fail(h, e_ == hipErrorOutOfMemory ? PBD_ERR_NOMEM : PBD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                 \
        }                                                                                           \
    } while (0)

// "No exception crosses this ABI" (include/pbd.h): every extern "C" body runs inside guarded().  The
// reference's errors on this path are CV_Error / bool returns, never process death
// (src/HOGFeatures.cpp:141-145, src/FileStorageModel.cpp:100-101).
template <class F>
int guarded(pbd_handle *h, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(h, PBD_ERR_NOMEM, "out of host memory");
    } catch (const std::length_error &e) {
        return fail(h, PBD_ERR_NOMEM, "host allocation too large: %s", e.what());
    } catch (const std::exception &e) {
        return fail(h, PBD_ERR_INVALID, "unexpected exception: %s", e.what());
    } catch (...) {
        return fail(h, PBD_ERR_INVALID, "unexpected exception");
    }
}

// While a ProfScope is alive, every kernel launched by this thread is timed under kernel id `k` (see PBD_LAUNCH).
struct ProfScope {
    pbd_handle *h; int k; ProfHook hook; ProfHook *prev;
    static void take(void *ctx, hipEvent_t *a, hipEvent_t *b)
    {
        ProfScope *self = static_cast<ProfScope *>(ctx);
        *a = self->h->prof.get(); *b = self->h->prof.get();
        self->h->prof.recs.push_back({self->k, *a, *b});
    }
    ProfScope(pbd_handle *h_, int k_, hipStream_t) : h(h_), k(k_), hook{this, &ProfScope::take}, prev(g_prof_hook)
    {
        if (h->prof.on == 1 || (h->prof.on == 2 && k == PBD_K_CONV)) g_prof_hook = &hook;
    }
    ~ProfScope() { g_prof_hook = prev; }
    ProfScope(const ProfScope &) = delete;
    ProfScope &operator=(const ProfScope &) = delete;
};

// ---- plan construction -------------------------------------------------------------------------
// Cover a rows x cols level with 256-cell tiles of shape 32x8, 16x16 or 8x32 (shape k: 32>>k wide, 8<<k
// high) so that the fewest lanes compute cells outside the level: a dynamic program over the columns picks
// the vertical strips, each strip is then cut into tiles of its shape.
static void cover_level(int l, int rows, int cols, std::vector<ConvTile> *shaped)
{
    const long long INF = 1LL << 60;
    std::vector<long long> cost(cols + 1, INF);
    std::vector<int> pick(cols + 1, -1);
    cost[0] = 0;
    for (int w = 1; w <= cols; ++w)
        for (int k = 0; k < 3; ++k) {
            const int tw = kConvTW >> k, th = kConvTH << k;
            const int prev = std::max(w - tw, 0);
            const long long c = cost[prev] + (long long)tw * ((rows + th - 1) / th) * th;
            if (c < cost[w]) { cost[w] = c; pick[w] = k; }
        }
    std::vector<std::pair<int, int>> strips;   // (x0, shape), right to left
    for (int w = cols; w > 0;) {
        const int k = pick[w], tw = kConvTW >> k;
        const int x0 = std::max(w - tw, 0);
        strips.push_back({x0, k});
        w = x0;
    }
    for (auto it = strips.rbegin(); it != strips.rend(); ++it) {
        const int k = it->second, th = kConvTH << k;
        for (int y0 = 0; y0 < rows; y0 += th) shaped[k].push_back({l, y0, it->first});
    }
}

// Tiles of the exact 5 x 5 convolution for a launch of `nb` frames (pbd_kernels_conv.hip): the strips of four rows of every
// level of every frame, left to right, form one sequence of positions; a tile takes 64 consecutive positions, in at most
// kConvMaxSeg runs (a run stays inside one strip) -- when a fourth run would be needed the tile ends early.
void build_seg_tiles(const std::vector<LevelDesc> &lv, int nb, std::vector<ConvSegTile> &out)
{
    out.clear();
    ConvSegTile cur{};
    int lanes = 0;
    auto flush = [&]() { if (cur.nseg) out.push_back(cur); cur = ConvSegTile{}; lanes = 0; };
    for (int f = 0; f < nb; ++f)
        for (int l = 0; l < (int)lv.size(); ++l) {
            const int H = lv[l].rows, W = lv[l].cols;
            if (H <= 0 || W <= 0) continue;
            for (int st = 0; st < (H + 3) / 4; ++st)
                for (int x = 0; x < W;) {
                    if (cur.nseg == kConvMaxSeg || lanes == 64) flush();
                    const int take = std::min(W - x, 64 - lanes);
                    cur.len[cur.nseg] = take;
                    cur.seg[cur.nseg] = ConvSeg{f, l, st, x};
                    cur.nseg += 1;
                    lanes += take;
                    x += take;
                }
        }
    flush();
}

hipError_t finish_plan_tables(Plan &P, int sbin)
{
    // flat row / column lookup and conv tiles over the feature maps
    std::vector<int> row2level, rowoff(P.nlevels + 1, 0), col2level, coloff(P.nlevels + 1, 0);
    std::vector<ConvTile> tiles, shaped[3], htiles;
    P.quad_per_frame = 0;
    for (int l = 0; l < P.nlevels; ++l) {
        P.lv[l].quad_off = P.quad_per_frame;
        P.quad_per_frame += ((long long)P.lv[l].rows * P.lv[l].cols + 3) / 4;
        const LevelDesc &d = P.lv[l];
        for (int by0 = 0; by0 < d.blk_rows && d.blk_cols > 0; by0 += hog_tile_rows(sbin))
            for (int bx0 = 0; bx0 < d.blk_cols; bx0 += kHogTBX) htiles.push_back({l, by0, bx0});
        rowoff[l] = (int)row2level.size();
        coloff[l] = (int)col2level.size();
        if (d.rows > 0 && d.cols > 0) {
            for (int y = 0; y < d.rows; ++y) row2level.push_back(l);
            for (int x = 0; x < d.cols; ++x) col2level.push_back(l);
            for (int y0 = 0; y0 < d.rows; y0 += kConvTH)
                for (int x0 = 0; x0 < d.cols; x0 += kConvTW) tiles.push_back({l, y0, x0});
            cover_level(l, d.rows, d.cols, shaped);
        }
    }
    {
        int longest = 0;
        for (const LevelDesc &d : P.lv) longest = std::max(longest, std::max(d.rows, d.cols));
        P.ptr8 = longest <= 256;
        P.longest = longest;
    }
    rowoff[P.nlevels] = (int)row2level.size();
    coloff[P.nlevels] = (int)col2level.size();
    P.nrows_flat = (int)row2level.size();
    P.ncols_flat = (int)col2level.size();
    P.ntiles = (int)tiles.size();
    P.nhtiles = (int)htiles.size();
    // wave-private stack regions: a wave of 64 flat rows (columns) needs 64 x ceil(longest row in the wave / 2)
    // two-entry records; levels are ordered large to small, but take the maximum to be safe
    std::vector<long long> srow, scol;
    long long tot_r = 0, tot_c = 0;
    for (int w0 = 0; w0 < P.nrows_flat; w0 += 64) {
        int mx = 0;
        for (int r = w0; r < std::min(w0 + 64, P.nrows_flat); ++r) mx = std::max(mx, P.lv[row2level[r]].cols);
        srow.push_back(tot_r);
        tot_r += 64LL * ((mx + 1) / 2);
    }
    for (int w0 = 0; w0 < P.ncols_flat; w0 += 64) {
        int mx = 0;
        for (int c = w0; c < std::min(w0 + 64, P.ncols_flat); ++c) mx = std::max(mx, P.lv[col2level[c]].rows);
        scol.push_back(tot_c);
        tot_c += 64LL * ((mx + 1) / 2);
    }
    P.stk_per_jf = std::max(tot_r, tot_c);
    std::vector<ConvTile> all;
    for (int k = 0; k < 3; ++k) { P.nshaped[k] = (int)shaped[k].size(); all.insert(all.end(), shaped[k].begin(), shaped[k].end()); }
    hipError_t e;
    if ((e = P.d_stk_row_off.upload(srow)) != hipSuccess) return e;
    if ((e = P.d_stk_col_off.upload(scol)) != hipSuccess) return e;
    if ((e = P.d_lv.upload(P.lv)) != hipSuccess) return e;
    if ((e = P.d_tiles.upload(tiles)) != hipSuccess) return e;
    if ((e = P.d_shaped.upload(all)) != hipSuccess) return e;
    if ((e = P.d_htiles.upload(htiles)) != hipSuccess) return e;
    if ((e = P.d_row2level.upload(row2level)) != hipSuccess) return e;
    if ((e = P.d_rowoff.upload(rowoff)) != hipSuccess) return e;
    if ((e = P.d_col2level.upload(col2level)) != hipSuccess) return e;
    if ((e = P.d_coloff.upload(coloff)) != hipSuccess) return e;
    return P.d_scales.upload(P.scales);
}

// A finished plan joins the cache; the cache holds at most 16 plans and never evicts the plan the handle's
// staged results refer to (h->cur).
void cache_plan(pbd_handle *h, std::unique_ptr<Plan> P)
{
    h->plans.push_back(std::move(P));
    while (h->plans.size() > 16) {
        auto it = h->plans.begin();
        if (it->get() == h->cur) ++it;
        h->plans.erase(it);
    }
}

int get_image_plan(pbd_handle *h, int rows, int cols, Plan **out)
{
    for (auto &p : h->plans)
        if (p->kind == 0 && p->rows == rows && p->cols == cols) { *out = p.get(); return PBD_OK; }
    std::vector<int> lr, lc;
    std::vector<float> scales;
    const int n = plan_pyramid(rows, cols, h->sbin, h->interval, lr, lc, scales);
    if (n <= 0)
        return fail(h, PBD_ERR_INVALID, "frame %dx%d too small for sbin %d / interval %d (nscales %d)", rows, cols,
                    h->sbin, h->interval, n);
    auto P = std::make_unique<Plan>();
    P->kind = 0; P->rows = rows; P->cols = cols;
    P->nlevels = n; P->scales = scales; P->interval = h->interval;
    P->lv.resize(n);
    std::vector<ResizeTabX> tabx;
    std::vector<ResizeTabY> taby;
    std::vector<ResizeTabXf> tabxf;
    std::vector<ResizeTabYf> tabyf;
    long long pix = 0, blk = 0, cell = 0;
    for (int l = 0; l < n; ++l) {
        LevelDesc &d = P->lv[l];
        d.img_rows = lr[l]; d.img_cols = lc[l];
        if (d.img_rows < 4 || d.img_cols < 4) return fail(h, PBD_ERR_INVALID, "pyramid level %d is %dx%d", l, lr[l], lc[l]);
        d.blk_cols = (int)roundf((float)lc[l] / (float)h->sbin);   // HOGFeatures.cpp:174
        d.blk_rows = (int)roundf((float)lr[l] / (float)h->sbin);
        d.cols = std::max(d.blk_cols - 2, 0);
        d.rows = std::max(d.blk_rows - 2, 0);
        d.src_level = l >= h->interval ? l - h->interval : -1;
        d.img_off = pix; d.blk_off = blk; d.cell_off = cell;
        d.tab_x = d.tab_y = 0;
        pix += (long long)lr[l] * lc[l];
        blk += (long long)d.blk_rows * d.blk_cols;
        cell += (long long)d.rows * d.cols;
        if (l == h->interval - 1) P->npix_resized = pix;
        if (l < h->interval) {
            // cv::resize INTER_LINEAR 8U coefficient tables (OpenCV imgwarp.cpp)
            const double scale_x = 1. / ((double)lc[l] / cols), scale_y = 1. / ((double)lr[l] / rows);
            d.tab_x = (int)tabx.size();
            d.tab_y = (int)taby.size();
            for (int dx = 0; dx < lc[l]; ++dx) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = (int)floorf(fx);
                fx -= (float)sx;
                if (sx < 0) { fx = 0; sx = 0; }
                const int last = sx >= cols - 1;
                if (last) { fx = 0; sx = cols - 1; }
                tabx.push_back({sx, sat_short_round((1.f - fx) * 2048), sat_short_round(fx * 2048)});
                tabxf.push_back({sx, last, 1.f - fx, fx});
            }
            for (int dy = 0; dy < lr[l]; ++dy) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = (int)floorf(fy);
                fy -= (float)sy;
                const int y0 = std::min(std::max(sy, 0), rows - 1), y1 = std::min(std::max(sy + 1, 0), rows - 1);
                taby.push_back({y0, y1, sat_short_round((1.f - fy) * 2048), sat_short_round(fy * 2048)});
                tabyf.push_back({y0, y1, 1.f - fy, fy});
            }
        }
    }
    if (h->shard_world > 1) {
        // Level sharding (secondary partitioning): levels are independent through HOG, convolution and DP, so
        // one frame can be split over GPUs by giving each a subset of the levels.  Longest-processing-time assignment
        // over the cell counts (level 0 alone is 13 % of a VGA frame): levels by decreasing size, each to the rank
        // with the least work so far.  Levels of other ranks keep their pyramid image here (a pyrDown chain may run
        // through them) but get empty block / feature maps, so every later stage skips them.
        std::vector<int> order(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            return (long long)P->lv[a].rows * P->lv[a].cols > (long long)P->lv[b].rows * P->lv[b].cols;
        });
        std::vector<long long> load(h->shard_world, 0);
        std::vector<int> owner(n, 0);
        for (int l : order) {
            const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            owner[l] = r;
            load[r] += (long long)P->lv[l].rows * P->lv[l].cols;
        }
        blk = 0; cell = 0;
        for (int l = 0; l < n; ++l) {
            LevelDesc &d = P->lv[l];
            if (owner[l] != h->shard_rank) d.blk_rows = d.blk_cols = d.rows = d.cols = 0;
            d.blk_off = blk; d.cell_off = cell;
            blk += (long long)d.blk_rows * d.blk_cols;
            cell += (long long)d.rows * d.cols;
        }
    }
    P->pix_per_frame = pix; P->blk_per_frame = blk; P->cell_per_frame = cell;
    if (P->npix_resized == 0) P->npix_resized = pix;
    HIPCHK(h, P->d_tabx.upload(tabx));
    HIPCHK(h, P->d_taby.upload(taby));
    HIPCHK(h, P->d_tabxf.upload(tabxf));
    HIPCHK(h, P->d_tabyf.upload(tabyf));
    HIPCHK(h, finish_plan_tables(*P, h->sbin));
    // HOG coordinate table grows with the largest frame seen
    const int need = std::max(rows, cols) + 4 * h->sbin + 8;
    if (need > h->coord_n) {
        // HOGFeatures.cpp:252-259: yp = ((T)y+0.5)/(T)sbin - 0.5; iyp = floor(yp); vy0 = yp-iyp; vy1 = 1.0-vy0
        if (h->f64) {
            std::vector<HogCoordD> coord(need);
            for (int t = 0; t < need; ++t) {
                const double tp = ((double)t + 0.5) / (double)h->sbin - 0.5;
                const int ip = (int)floor(tp);
                const double v0 = tp - (double)ip;
                coord[t] = {ip, v0, 1.0 - v0};
            }
            HIPCHK(h, h->d_coord.ensure(coord.size() * sizeof(HogCoordD)));
            HIPCHK(h, hipMemcpy(h->d_coord.p, coord.data(), coord.size() * sizeof(HogCoordD), hipMemcpyHostToDevice));
        } else {
            std::vector<HogCoord> coord(need);
            for (int t = 0; t < need; ++t) {
                const float tp = (float)(((double)(float)t + 0.5) / (double)(float)h->sbin - 0.5);
                const int ip = (int)floorf(tp);
                const float v0 = tp - (float)ip;
                const float v1 = (float)(1.0 - (double)v0);
                coord[t] = {ip, v0, v1};
            }
            HIPCHK(h, h->d_coord.ensure(coord.size() * sizeof(HogCoord)));
            HIPCHK(h, hipMemcpy(h->d_coord.p, coord.data(), coord.size() * sizeof(HogCoord), hipMemcpyHostToDevice));
        }
        h->coord_n = need;
    }
    *out = P.get();
    cache_plan(h, std::move(P));
    return PBD_OK;
}

int get_dims_plan(pbd_handle *h, int nlevels, const int *rows, const int *cols, Plan **out)
{
    if (nlevels <= 0 || nlevels > PBD_MAX_LEVELS) return fail(h, PBD_ERR_INVALID, "nlevels %d out of range", nlevels);
    std::vector<int> key;
    for (int l = 0; l < nlevels; ++l) {
        if (rows[l] < 0 || cols[l] < 0 || rows[l] > 32000 || cols[l] > 32000)
            return fail(h, PBD_ERR_INVALID, "level %d size %dx%d out of range", l, rows[l], cols[l]);
        key.push_back(rows[l]); key.push_back(cols[l]);
    }
    for (auto &p : h->plans)
        if (p->kind == 1 && p->key_dims == key) { *out = p.get(); return PBD_OK; }
    auto P = std::make_unique<Plan>();
    P->kind = 1; P->key_dims = key; P->nlevels = nlevels; P->interval = h->interval;
    P->lv.resize(nlevels);
    P->scales.assign(nlevels, 1.f);
    long long cell = 0;
    for (int l = 0; l < nlevels; ++l) {
        LevelDesc &d = P->lv[l];
        memset(&d, 0, sizeof d);
        d.rows = rows[l]; d.cols = cols[l]; d.src_level = -1; d.cell_off = cell;
        cell += (long long)rows[l] * cols[l];
    }
    P->cell_per_frame = cell;
    HIPCHK(h, finish_plan_tables(*P, h->sbin));
    *out = P.get();
    cache_plan(h, std::move(P));
    return PBD_OK;
}

// IEEE binary16 <-> float on the host (round to nearest even, what v_cvt_f16_f32 does)
uint16_t host_f2h(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t ax = u & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 // NaN
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                // rounds to >= 65520: inf
    if (ax < 0x33000001u) return sign;                                       // below half the smallest subnormal: 0
    const int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                              // bits dropped from the 24-bit significand
    const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1);
    uint32_t q = m >> shift;
    if (rest > half || (rest == half && (q & 1u))) ++q;
    const uint32_t bits = e >= -14 ? (((uint32_t)(e + 15) << 10) + (q - 0x400u)) : q;   // carry propagates into the exponent
    return (uint16_t)(sign | bits);
}
float host_h2f(uint16_t hv)
{
    const uint32_t sign = (uint32_t)(hv & 0x8000u) << 16;
    const uint32_t ex = (hv >> 10) & 0x1fu, man = hv & 0x3ffu;
    uint32_t u;
    if (ex == 0x1f) u = sign | 0x7f800000u | (man << 13);
    else if (ex != 0) u = sign | ((ex + 112u) << 23) | (man << 13);
    else if (man == 0) u = sign;
    else {                                                                   // subnormal half: normalise
        int sh = 0;
        uint32_t mm = man;
        while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
        u = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13);
    }
    float f; memcpy(&f, &u, 4);
    return f;
}

// ---- model tables --------------------------------------------------------------------------------
// Work items of k_conv3: the nf filters of a size class cut into units of 8 / 6 (and at most one of 4 or 2) filters so that
// `nw` waves taking units largest-first finish together.  156 filters, 6 waves: 6 x 8 + 18 x 6 -- every wave 8 + 6 + 6 + 6.
void conv_units(int nf, int nw, std::vector<int> &f0, std::vector<int> &ql)
{
    const int P = (nf + 1) / 2;                            // filter pairs (an odd bank ends in a padding lane)
    double best = 1e30;
    int ba = 0, bb = 0, bc = 0;
    for (int a = P / 4; a >= 0; --a)
        for (int c = 0; c <= 2; ++c) {                     // c pairs in one last small unit (0: none)
            const int rest = P - 4 * a - c;
            if (rest < 0 || rest % 3) continue;
            const int b = rest / 3;
            // largest-first hand-out to nw equally fast waves; a unit costs its pairs + a fixed overhead (window reads, stores)
            std::vector<double> load(nw, 0.0);
            auto give = [&](int n, double cost) { for (int i = 0; i < n; ++i) *std::min_element(load.begin(), load.end()) += cost; };
            give(a, 4 + 0.2); give(b, 3 + 0.2); if (c) give(1, c + 0.2);
            const double span = *std::max_element(load.begin(), load.end()) + (c ? 0.01 : 0.0);    // ties: no small unit
            if (span < best - 1e-9) { best = span; ba = a; bb = b; bc = c; }
        }
    f0.clear(); ql.clear();
    int f = 0;
    for (int i = 0; i < ba; ++i) { f0.push_back(f); ql.push_back(8); f += 8; }
    for (int i = 0; i < bb; ++i) { f0.push_back(f); ql.push_back(6); f += 6; }
    if (bc) { f0.push_back(f); ql.push_back(2 * bc); f += 2 * bc; }
}

template <typename R>
int upload_filters_t(pbd_handle *h, int nfilters, const void *const *filters, const int *ksize)
{
    if (nfilters <= 0) return fail(h, PBD_ERR_INVALID, "no filters");
    for (int f = 0; f < nfilters; ++f)
        if (ksize[f] < 1 || ksize[f] > kConvMaxK) return fail(h, PBD_ERR_UNSUPPORTED, "filter %d: size %d not supported (1..%d)", f, ksize[f], kConvMaxK);
    // size classes in order of first appearance
    std::vector<int> sizes;
    for (int f = 0; f < nfilters; ++f)
        if (std::find(sizes.begin(), sizes.end(), ksize[f]) == sizes.end()) sizes.push_back(ksize[f]);
    const int K = ksize[0];
    const bool fast = (sizeof(R) == 4 && K == 5 && sizes.size() == 1);
    const bool mfma = h->cfg.conv_mode == PBD_CONV_MFMA || h->cfg.conv_mode == PBD_CONV_MFMA_F16;
    if (mfma && !fast) return fail(h, PBD_ERR_UNSUPPORTED, "PBD_CONV_MFMA / PBD_CONV_MFMA_F16 need 5x5 filters and PBD_REAL_F32");
    // The new bank is built beside the old one and swapped in only when every upload has succeeded: a failed setFilters()
    // (out of memory, an unsupported size) leaves the handle with its previous, complete bank.
    struct NewBank {
        std::vector<pbd_handle::ConvClass> classes;
        DevBuf wrec;
        bool keep = false;
        ~NewBank() { if (!keep) { for (auto &c : classes) { c.wts.release(); c.fmap.release(); c.wts3.release(); c.unit_f0.release(); c.unit_ql.release(); c.unit_woff.release(); c.c31tab.release(); } wrec.release(); } }
    } nb;
    nb.classes.assign(sizes.size(), pbd_handle::ConvClass{});
    for (size_t ci = 0; ci < sizes.size(); ++ci) {
        pbd_handle::ConvClass &C = nb.classes[ci];
        C.K = sizes[ci];
        std::vector<int> ids;
        for (int f = 0; f < nfilters; ++f) if (ksize[f] == C.K) ids.push_back(f);
        C.nf = (int)ids.size();
        C.Fpad = (C.nf + kConvQ - 1) / kConvQ * kConvQ;
        // device layout: float 5x5 kernel [group][channel][tap][8] (800 contiguous bytes per (group, channel));
        // generic kernel (other sizes, T=double) [channel][tap][Fpad]
        const bool fast5 = (sizeof(R) == 4 && C.K == 5);
        const int KK = C.K * C.K;
        std::vector<R> w((size_t)32 * KK * C.Fpad, (R)0);
        for (int fl = 0; fl < C.nf; ++fl) {
            const R *src = static_cast<const R *>(filters[ids[fl]]);
            for (int t = 0; t < KK; ++t)
                for (int c = 0; c < 32; ++c) {
                    const R v = src[(size_t)t * 32 + c];
                    if (fast5) w[(((size_t)(fl / kConvQ) * 32 + c) * KK + t) * kConvQ + (fl % kConvQ)] = v;
                    else w[((size_t)c * KK + t) * C.Fpad + fl] = v;
                }
        }
        HIPCHK(h, C.wts.ensure(w.size() * sizeof(R)));
        HIPCHK(h, hipMemcpy(C.wts.p, w.data(), w.size() * sizeof(R), hipMemcpyHostToDevice));
        if (sizes.size() > 1) HIPCHK(h, C.fmap.upload(ids));
        if (fast5) {
            std::vector<int> uf0, uql;
            conv_units(C.nf, kConv3NW, uf0, uql);
            C.nunits = (int)uf0.size();
            std::vector<int> uoff(C.nunits, 0);
            size_t tot = 0;
            for (int u = 0; u < C.nunits; ++u) { uoff[u] = (int)tot; tot += (size_t)32 * KK * uql[u]; }
            std::vector<float> w3(tot + 16, 0.0f);       // (slack: the last tap row is fetched once more past the last channel)
            for (int u = 0; u < C.nunits; ++u)
                for (int q = 0; q < uql[u] && uf0[u] + q < C.nf; ++q) {
                    const R *src = static_cast<const R *>(filters[ids[uf0[u] + q]]);
                    for (int t = 0; t < KK; ++t)
                        for (int c = 0; c < 32; ++c) w3[uoff[u] + ((size_t)c * KK + t) * uql[u] + q] = (float)src[(size_t)t * 32 + c];
                }
            HIPCHK(h, C.unit_woff.upload(uoff));
            // channel 31 of a window that leaves the image: the reference's sum over the out-of-image taps (border value 1,
            // src/SpatialConvolutionEngine.cpp:147-156) in its tap order (raster, zero weights skipped: src/filter.cpp:3818-3856,
            // 3916-3922), for every combination of rows / columns outside at the top, bottom, left and right (0..2 each)
            C.c31stride = (C.nf + 15) & ~7;                     // a unit's 8 consecutive entries stay inside the row
            std::vector<float> tab((size_t)81 * C.c31stride, 0.0f);
            for (int cs = 0; cs < 81; ++cs) {
                const int right = cs % 3, left = cs / 3 % 3, bot = cs / 9 % 3, top = cs / 27;
                for (int fl = 0; fl < C.nf; ++fl) {
                    const R *src = static_cast<const R *>(filters[ids[fl]]);
                    float sum = 0.0f;
                    for (int i = 0; i < 5; ++i)
                        for (int j = 0; j < 5; ++j) {
                            if (!(i < top || i > 4 - bot || j < left || j > 4 - right)) continue;
                            const float w = (float)src[(size_t)(i * 5 + j) * 32 + 31];
                            if (w != 0.0f) sum = sum + w;
                        }
                    tab[(size_t)cs * C.c31stride + fl] = sum;
                }
            }
            HIPCHK(h, C.c31tab.upload(tab));
            HIPCHK(h, C.wts3.ensure(w3.size() * sizeof(float)));
            HIPCHK(h, hipMemcpy(C.wts3.p, w3.data(), w3.size() * sizeof(float), hipMemcpyHostToDevice));
            HIPCHK(h, C.unit_f0.upload(uf0));
            HIPCHK(h, C.unit_ql.upload(uql));
        }
    }
    const int Fpad = (nfilters + kConvQ - 1) / kConvQ * kConvQ;
    if (mfma) {
        const bool f16 = h->cfg.conv_mode == PBD_CONV_MFMA_F16;
        // bf16 mode: x = hi + lo with round-to-nearest-even; fp16 mode: one rounding
        auto f2bf = [](float f) -> uint16_t {
            uint32_t u; memcpy(&u, &f, 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        auto bf2f = [](uint16_t b) -> float { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
        // A-operand fragments in the order the kernel consumes them: [pass][tap][k-step][M-tile][hi|lo][lane] x 8 values.
        // Lane (r = lane & 31, hh = lane >> 5) of v_mfma_f32_32x32x16 holds row r (filter), k = hh*8 .. hh*8+7 (channels)
        const int NV = f16 ? 1 : 2, MT = kMfmaFilterBlock / 32;
        const int passes = (nfilters + kMfmaFilterBlock - 1) / kMfmaFilterBlock;
        std::vector<uint16_t> rec((size_t)passes * K * K * 2 * MT * NV * 64 * 8, 0);
        for (int ps = 0; ps < passes; ++ps)
            for (int t = 0; t < K * K; ++t)
                for (int kh = 0; kh < 2; ++kh)
                    for (int mt = 0; mt < MT; ++mt)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int f = ps * kMfmaFilterBlock + mt * 32 + (lane & 31);
                            if (f >= nfilters) continue;
                            const float *src = reinterpret_cast<const float *>(filters[f]) + (size_t)t * 32 + kh * 16 + (lane >> 5) * 8;
                            const size_t base = (((((size_t)ps * K * K + t) * 2 + kh) * MT + mt) * NV) * 64 * 8;
                            for (int j = 0; j < 8; ++j) {
                                const float v = src[j];
                                if (f16) { rec[base + (size_t)lane * 8 + j] = host_f2h(v); continue; }
                                const uint16_t hi = f2bf(v);
                                rec[base + (size_t)lane * 8 + j] = hi;
                                rec[base + (size_t)(64 + lane) * 8 + j] = f2bf(v - bf2f(hi));
                            }
                        }
        HIPCHK(h, nb.wrec.ensure(rec.size() * 2));
        HIPCHK(h, hipMemcpy(nb.wrec.p, rec.data(), rec.size() * 2, hipMemcpyHostToDevice));
    }
    // commit
    for (auto &c : h->conv_classes) { c.wts.release(); c.fmap.release(); c.wts3.release(); c.unit_f0.release(); c.unit_ql.release(); c.unit_woff.release(); c.c31tab.release(); }
    h->conv_classes.swap(nb.classes);
    nb.classes.clear();
    if (mfma) { h->d_wrec.release(); h->d_wrec = nb.wrec; nb.wrec = DevBuf{}; }
    nb.keep = true;
    h->F = nfilters; h->Fpad = Fpad; h->ksize = K;
    h->filter_ksize.assign(ksize, ksize + nfilters);
    h->filters_set = true;
    return PBD_OK;
}

int upload_filters(pbd_handle *h, int nfilters, const void *const *filters, const int *ksize)
{
    return h->f64 ? upload_filters_t<double>(h, nfilters, filters, ksize) : upload_filters_t<float>(h, nfilters, filters, ksize);
}

// After pbd_conv_set_filters replaced the bank: the model tables built by build_model index response planes by
// filter id, and the part boxes of argmin use the filter size (include/Parts.hpp:185-187), so both are re-checked /
// rebuilt against the new bank.  A bank that does not cover the model's ids leaves the convolution engine usable on
// its own (IConvolutionEngine::pdf) and makes the model-dependent calls fail with PBD_ERR_STATE.
int revalidate_bank(pbd_handle *h)
{
    h->bank_matches_model = true;
    for (int f : h->filterid)
        if (f < 0 || f >= h->F) { h->bank_matches_model = false; return PBD_OK; }
    for (int c = 0; c < h->NC; ++c) {
        const int p0 = h->part_offset[c], np = h->part_offset[c + 1] - p0;
        for (int p = 0; p < np; ++p) {
            PartWalk &w = h->walk[h->walk_off[c] + p];
            const int K = h->mix_offset[p0 + p + 1] - h->mix_offset[p0 + p];
            for (int mm = 0; mm < K; ++mm) w.ksize[mm] = h->filter_ksize[h->filterid[h->mix_offset[p0 + p] + mm]];
        }
    }
    HIPCHK(h, h->d_walk.upload(h->walk));
    return PBD_OK;
}

int build_model(pbd_handle *h, const pbd_model *m)
{
    if (m->flen != 32 || m->norient != 18)
        return fail(h, PBD_ERR_UNSUPPORTED, "flen %d / norient %d: only 32 / 18 are supported", m->flen, m->norient);
    if (m->ncomponents < 1 || m->nfilters < 1 || m->sbin < 2 || m->interval < 1)
        return fail(h, PBD_ERR_INVALID, "bad model header");
    h->NC = m->ncomponents; h->sbin = m->sbin; h->interval = m->interval; h->norient = m->norient;
    h->thresh = m->thresh;
    const int totparts = m->part_offset[m->ncomponents];
    const int totmix = m->mix_offset[totparts];
    h->part_offset.assign(m->part_offset, m->part_offset + m->ncomponents + 1);
    h->parentid.assign(m->parentid, m->parentid + totparts);
    h->mix_offset.assign(m->mix_offset, m->mix_offset + totparts + 1);
    h->filterid.assign(m->filterid, m->filterid + totmix);
    h->biasid.assign(m->biasid, m->biasid + totmix);
    h->defid.assign(m->defid, m->defid + totmix);
    h->biasw.assign(m->biasw, m->biasw + m->nbias);
    h->defw.assign(m->defw, m->defw + (size_t)m->ndefs * 4);
    h->anchors.assign(m->anchors, m->anchors + (size_t)m->ndefs * 2);

    // filters
    {
        if (h->f64 ? !m->filters_f64 : !m->filters_f32) return fail(h, PBD_ERR_INVALID, "model has no filters of the requested real type");
        std::vector<const void *> fp(m->nfilters);
        for (int f = 0; f < m->nfilters; ++f)
            fp[f] = h->f64 ? static_cast<const void *>(m->filters_f64 + m->filter_offset[f])
                           : static_cast<const void *>(m->filters_f32 + m->filter_offset[f]);
        int rc = upload_filters(h, m->nfilters, fp.data(), m->filter_ksize);
        if (rc != PBD_OK) return rc;
    }

    // validation + pointer slots + depth
    h->ptr_slot.assign(totparts, 0);
    std::vector<int> depth(totparts, 0);
    int NS = 0, maxdepth = 0;
    h->max_parts = 0;
    for (int c = 0; c < h->NC; ++c) {
        const int p0 = h->part_offset[c], np = h->part_offset[c + 1] - p0;
        if (np < 1) return fail(h, PBD_ERR_INVALID, "component %d has no parts", c);
        h->max_parts = std::max(h->max_parts, np);
        std::set<int> seen;
        for (int p = 0; p < np; ++p) {
            const int gp = p0 + p, par = h->parentid[gp];
            const int K = h->mix_offset[gp + 1] - h->mix_offset[gp];
            if (K < 1 || K > kMaxMix) return fail(h, PBD_ERR_UNSUPPORTED, "part %d has %d mixtures (1..%d supported)", p, K, kMaxMix);
            h->max_mix = std::max(h->max_mix, K);
            if ((p == 0) != (par < 0) || par >= p) return fail(h, PBD_ERR_INVALID, "part %d: parent %d breaks topological order", p, par);
            for (int mm = 0; mm < K; ++mm) {
                const int f = h->filterid[h->mix_offset[gp] + mm];
                if (f < 0 || f >= h->F) return fail(h, PBD_ERR_INVALID, "filter id %d out of range", f);
                if (!seen.insert(f).second) h->seq_mode = true;   // accumulators keyed by filter id interact: see below
            }
            h->ptr_slot[gp] = NS;
            if (p > 0) {
                const int gpar = p0 + par;
                const int L = h->mix_offset[gpar + 1] - h->mix_offset[gpar];
                NS += L;
                depth[gp] = depth[gpar] + 1;
                maxdepth = std::max(maxdepth, depth[gp]);
                for (int mm = 0; mm < K; ++mm) {
                    const int gm = h->mix_offset[gp] + mm;
                    const int d = h->defid[gm], b = h->biasid[gm];
                    if (d < 0 || d >= m->ndefs) return fail(h, PBD_ERR_INVALID, "defid %d out of range", d);
                    if (b < 0 || b + L > m->nbias) return fail(h, PBD_ERR_INVALID, "biasid %d out of range", b);
                    if (h->defw[(size_t)d * 4 + 0] == 0.f || h->defw[(size_t)d * 4 + 2] == 0.f)
                        return fail(h, PBD_ERR_INVALID, "deformation %d has a zero quadratic term", d);
                }
            } else {
                const int b = h->biasid[h->mix_offset[gp]];
                if (b < 0 || b >= m->nbias) return fail(h, PBD_ERR_INVALID, "root biasid %d out of range", b);
            }
        }
    }
    if (h->max_parts > kWalkMaxParts) return fail(h, PBD_ERR_UNSUPPORTED, "%d parts per component (max %d)", h->max_parts, kWalkMaxParts);
    h->NS = NS;

    // children (descending index) per part
    std::vector<std::vector<int>> children(totparts);
    for (int c = 0; c < h->NC; ++c) {
        const int p0 = h->part_offset[c], np = h->part_offset[c + 1] - p0;
        for (int p = np - 1; p > 0; --p) children[p0 + h->parentid[p0 + p]].push_back(p0 + p);
    }

    h->groups.clear();
    h->JGmax = 0;
    h->NM = totmix;
    h->totmix = totmix;
    auto dt_job = [&](int gm, bool from_acc, int plane) {
        DtJob j{};
        j.from_acc = from_acc ? 1 : 0;
        j.plane = plane;
        j.gm = gm;
        const int d = h->defid[gm];
        const float *w = &h->defw[(size_t)d * 4];
        j.ax = (double)(-w[0]); j.bx = (double)(-w[1]); j.ay = (double)(-w[2]); j.by = (double)(-w[3]);
        j.osx = h->anchors[(size_t)d * 2]; j.osy = h->anchors[(size_t)d * 2 + 1];
        return j;
    };
    std::vector<std::vector<char>> touched(h->NC, std::vector<char>(h->F, 0));   // sequential schedule: accumulator (c, f) exists
    if (h->seq_mode) {
        // The reference's own order (src/DynamicProgram.cpp:95): parts nparts-1 .. 1, one step per part; the
        // accumulated scores live in planes keyed by (component, filter id) as its `ncscores` (:93,115-119,154-156).
        // Step s takes part np-1-s of every component (components are independent).
        h->NM = h->NC * h->F;
        for (int sidx = 0; sidx + 1 < h->max_parts; ++sidx) {
            Group g;
            for (int c = 0; c < h->NC; ++c) {
                const int p0 = h->part_offset[c], np = h->part_offset[c + 1] - p0;
                const int p = np - 1 - sidx;
                if (p < 1) continue;
                const int gp = p0 + p, gpar = p0 + h->parentid[gp];
                const int K = h->mix_offset[gp + 1] - h->mix_offset[gp], L = h->mix_offset[gpar + 1] - h->mix_offset[gpar];
                SeqCombineJob sj{};
                sj.job_begin = (int)g.jobs.size(); sj.nmix = K; sj.slot = h->ptr_slot[gp]; sj.npar = L;
                for (int mm = 0; mm < K; ++mm) {
                    const int gm = h->mix_offset[gp] + mm, f = h->filterid[gm];
                    g.jobs.push_back(dt_job(gm, touched[c][f] != 0, touched[c][f] ? c * h->F + f : f));   // score_in, :115-119
                    sj.bias_off[mm] = h->biasid[gm];
                }
                for (int pm = 0; pm < L; ++pm) {
                    const int fp = h->filterid[h->mix_offset[gpar] + pm];
                    sj.target[pm] = c * h->F + fp; sj.filter[pm] = fp;
                    sj.init[pm] = touched[c][fp] ? 0 : 1;
                    touched[c][fp] = 1;
                }
                g.sjobs.push_back(sj);
            }
            h->JGmax = std::max(h->JGmax, (int)g.jobs.size());
            h->groups.push_back(std::move(g));
        }
    }
    // depth groups, deepest first: DT jobs of the parts at depth `dep`, combine jobs of their parents
    for (int dep = h->seq_mode ? 0 : maxdepth; dep >= 1; --dep) {
        Group g;
        std::vector<int> job_begin_of(totparts, -1);
        for (int c = 0; c < h->NC; ++c) {
            const int p0 = h->part_offset[c], np = h->part_offset[c + 1] - p0;
            for (int p = 1; p < np; ++p) {
                const int gp = p0 + p;
                if (depth[gp] != dep) continue;
                const int K = h->mix_offset[gp + 1] - h->mix_offset[gp];
                job_begin_of[gp] = (int)g.jobs.size();
                for (int mm = 0; mm < K; ++mm) {
                    const int gm = h->mix_offset[gp] + mm;
                    g.jobs.push_back(dt_job(gm, !children[gp].empty(), children[gp].empty() ? h->filterid[gm] : gm));
                }
            }
            // parents at depth dep-1 whose children (all at depth dep) were just listed
            for (int p = 0; p < np; ++p) {
                const int gpar = p0 + p;
                if (depth[gpar] != dep - 1 || children[gpar].empty()) continue;
                const int L = h->mix_offset[gpar + 1] - h->mix_offset[gpar];
                CombineJob cj{};
                cj.npar = L; cj.acc_plane = h->mix_offset[gpar];
                for (int pm = 0; pm < L; ++pm) cj.filter[pm] = h->filterid[h->mix_offset[gpar] + pm];
                cj.child_begin = (int)g.childs.size();
                for (int ch : children[gpar]) {   // already in descending index order
                    ChildDesc cd{};
                    cd.job_begin = job_begin_of[ch];
                    cd.nmix = h->mix_offset[ch + 1] - h->mix_offset[ch];
                    cd.slot = h->ptr_slot[ch];
                    for (int mm = 0; mm < cd.nmix; ++mm) cd.bias_off[mm] = h->biasid[h->mix_offset[ch] + mm];
                    g.childs.push_back(cd);
                }
                cj.child_end = (int)g.childs.size();
                g.cjobs.push_back(cj);
            }
        }
        h->JGmax = std::max(h->JGmax, (int)g.jobs.size());
        h->groups.push_back(std::move(g));
    }
    for (auto &g : h->groups) {
        HIPCHK(h, g.d_jobs.upload(g.jobs));
        // the usual deformation (w1 = w3 = +0.0f, so b = -0.0): the passes then run without the b terms
        auto neg_zero = [](double v) { return v == 0.0 && std::signbit(v); };
        g.bz_x = g.bz_y = 1;
        for (const DtJob &j : g.jobs) {
            if (!(neg_zero(j.bx) && j.ax != 0.0)) g.bz_x = 0;
            if (!(neg_zero(j.by) && j.ay != 0.0)) g.bz_y = 0;
        }
        HIPCHK(h, g.d_childs.upload(g.childs));
        HIPCHK(h, g.d_cjobs.upload(g.cjobs));
        HIPCHK(h, g.d_sjobs.upload(g.sjobs));
    }
    h->rjobs.assign(h->NC, RootJob{});
    h->walk.clear(); h->walk_off.assign(h->NC + 1, 0);
    for (int c = 0; c < h->NC; ++c) {
        const int p0 = h->part_offset[c], np = h->part_offset[c + 1] - p0;
        RootJob &r = h->rjobs[c];
        r.nmix = h->mix_offset[p0 + 1] - h->mix_offset[p0];
        r.from_acc = 0;
        for (int mm = 0; mm < r.nmix; ++mm) {
            const int f = h->filterid[h->mix_offset[p0] + mm];
            const bool acc = h->seq_mode ? touched[c][f] != 0 : !children[p0].empty();
            if (acc) r.from_acc |= 1 << mm;
            r.plane[mm] = !acc ? f : h->seq_mode ? c * h->F + f : h->mix_offset[p0] + mm;
        }
        r.bias = h->biasw[h->biasid[h->mix_offset[p0]]];
        h->walk_off[c] = (int)h->walk.size();
        for (int p = 0; p < np; ++p) {
            PartWalk w{};
            w.parent = h->parentid[p0 + p];
            w.slot = h->ptr_slot[p0 + p];
            w.mix0 = h->mix_offset[p0 + p];
            const int K = h->mix_offset[p0 + p + 1] - h->mix_offset[p0 + p];
            for (int mm = 0; mm < K; ++mm) w.ksize[mm] = m->filter_ksize[h->filterid[h->mix_offset[p0 + p] + mm]];
            h->walk.push_back(w);
        }
    }
    h->walk_off[h->NC] = (int)h->walk.size();
    HIPCHK(h, h->d_rjobs.upload(h->rjobs));
    HIPCHK(h, h->d_walk.upload(h->walk));
    HIPCHK(h, h->d_walk_off.upload(h->walk_off));
    HIPCHK(h, h->d_biasw.upload(h->biasw));
    return PBD_OK;
}

// ---- stages --------------------------------------------------------------------------------------
// alloc_* size the grow-only workspace for `nframes`; launch_* enqueue the kernels for frames
// [f0, f0+nb) on stream `st` (no allocation, no synchronisation inside).
int alloc_features(pbd_handle *h, Plan &P, int nframes, int cn)
{
    HIPCHK(h, h->pyr.ensure((size_t)nframes * P.pix_per_frame * cn * depth_size(h->cur_depth) + 32));   // slack: 8-bit pixels are read with 4- and 16-byte loads
    HIPCHK(h, h->gmag.ensure((size_t)nframes * P.pix_per_frame * h->rs));
    HIPCHK(h, h->gori.ensure((size_t)nframes * P.pix_per_frame + 16));
    HIPCHK(h, h->hist.ensure((size_t)nframes * P.blk_per_frame * 18 * h->rs));
    HIPCHK(h, h->norm.ensure((size_t)nframes * P.blk_per_frame * h->rs));
    HIPCHK(h, h->feat.ensure(std::max<size_t>((size_t)nframes * P.cell_per_frame * 32 * h->rs, 16)));
    return PBD_OK;
}

void launch_features(pbd_handle *h, Plan &P, const void *d_frames, int cn, int f0, int nb, hipStream_t st)
{
    PyrParams pp{};
    pp.lv = P.d_lv.d; pp.nlevels = P.nlevels; pp.interval = std::min(P.interval, P.nlevels); pp.cn = cn; pp.frame0 = f0;
    pp.pix_per_frame = P.pix_per_frame; pp.pyr = h->pyr.as<uint8_t>(); pp.frames = static_cast<const uint8_t *>(d_frames);
    pp.rows = P.rows; pp.cols = P.cols; pp.tabx = P.d_tabx.d; pp.taby = P.d_taby.d;
    pp.depth = h->cur_depth; pp.tabxf = P.d_tabxf.d; pp.tabyf = P.d_tabyf.d;
    {
        ProfScope ps(h, PBD_K_RESIZE, st);
        launch_resize(pp, nb, P.npix_resized, st);
    }
    for (int first = P.interval; first < P.nlevels; first += P.interval) {
        const int last = std::min(first + P.interval, P.nlevels);
        const long long base = P.lv[first].img_off;
        const long long end = (last < P.nlevels) ? P.lv[last].img_off : P.pix_per_frame;
        ProfScope ps(h, PBD_K_PYRDOWN, st);
        launch_pyrdown_range(pp, nb, first, last, base, end - base, st);
    }
    HogParams hp{};
    hp.lv = P.d_lv.d; hp.nlevels = P.nlevels; hp.cn = cn; hp.sbin = h->sbin; hp.frame0 = f0;
    hp.pix_per_frame = P.pix_per_frame; hp.blk_per_frame = P.blk_per_frame; hp.cell_per_frame = P.cell_per_frame;
    hp.pyr = h->pyr.as<uint8_t>(); hp.depth = h->cur_depth; hp.coord = h->d_coord.p;
    hp.gmag = h->gmag.p; hp.gori = h->gori.as<uint8_t>();
    hp.hist = h->hist.p; hp.norm = h->norm.p; hp.feat = h->feat.p;
    hp.htiles = P.d_htiles.d; hp.nhtiles = P.nhtiles;
    {
        ProfScope ps(h, PBD_K_HOG_HIST, st);
        launch_hog_hist(hp, nb, h->f64, st);
    }
    {
        ProfScope ps(h, PBD_K_HOG_FEAT, st);
        launch_hog_feat(hp, nb, h->f64, st);
    }
    h->feat_c31_zero = true;
}

int ensure_seg_tiles(pbd_handle *h, Plan &P, int nb)
{
    if (nb < 1 || P.segtiles.count(nb)) return PBD_OK;
    std::vector<ConvSegTile> tiles;
    build_seg_tiles(P.lv, nb, tiles);
    DevTable<ConvSegTile> t;
    HIPCHK(h, t.upload(tiles));
    P.segtiles[nb] = t;
    return PBD_OK;
}

int alloc_conv(pbd_handle *h, Plan &P, int nframes)
{
    if (!h->filters_set) return fail(h, PBD_ERR_STATE, "pdf() before setFilters()");
    HIPCHK(h, h->resp.ensure(std::max<size_t>((size_t)nframes * P.cell_per_frame * h->F * h->resp_es, 16) + 32));   // + slack: 16-half chunk reads
    return PBD_OK;
}

void launch_conv_stage(pbd_handle *h, Plan &P, int f0, int nb, hipStream_t st)
{
    ConvParams cp{};
    cp.lv = P.d_lv.d; cp.tiles = P.d_tiles.d; cp.ntiles = P.ntiles;
    cp.shaped = P.d_shaped.d;
    for (int k = 0; k < 3; ++k) cp.nshaped[k] = P.nshaped[k];
    cp.segtiles = nullptr; cp.nsegtiles = 0;
    if (!h->f64 && h->cfg.conv_mode != PBD_CONV_MFMA && h->cfg.conv_mode != PBD_CONV_MFMA_F16) {
        const auto it = P.segtiles.find(nb);     // built by ensure_seg_tiles before the first launch of this many frames
        if (it != P.segtiles.end()) { cp.segtiles = it->second.d; cp.nsegtiles = (int)it->second.n; }
    }
    cp.F = h->F; cp.frame0 = f0;
    cp.cell_per_frame = P.cell_per_frame;
    cp.feat = h->feat.p; cp.resp = h->resp.p;
    cp.fma = h->cfg.conv_mode == PBD_CONV_FMA;
    cp.c31_zero = h->feat_c31_zero ? 1 : 0;
    ProfScope ps(h, PBD_K_CONV, st);
    for (const pbd_handle::ConvClass &C : h->conv_classes) {     // one launch per filter size (one class in every known model)
        cp.nf = C.nf; cp.Fpad = C.Fpad; cp.ksize = C.K; cp.wts = C.wts.p; cp.fmap = C.fmap.d;
        const int ngroups = C.Fpad / kConvQ;
        // few workgroups (single frame): split the filter groups over more workgroups to fill the chip
        const long long wgs = (long long)P.ntiles * nb;
        cp.groups_per_block = wgs >= 1024 ? ngroups : std::max(1, (int)(ngroups * wgs / 1024));
        {   // Channels of the haloed tile staged in LDS at a time (generic kernel).  All 32 staged once is the least work, but 110 KB
            // (T = double, 5 x 5) leaves ONE workgroup -- one wave per SIMD -- on a CU and the kernel then waits for its own LDS
            // reads: the largest block within 36 KB (four or more workgroups per CU) is taken, restaged per filter group.
            // Measured, one 640x480 frame, T = double: 9.84 ms with 32 channels -> see profiles/README.md.
            static const int env_cb = getenv("PBD_CONV_CBLOCK") ? atoi(getenv("PBD_CONV_CBLOCK")) : 0;
            const size_t plane = (size_t)(((kConvTH + C.K - 1) * (kConvTW + C.K - 1)) | 1) * h->rs;
            cp.cblock = 32;
            while (cp.cblock > 1 && cp.cblock * plane > (size_t)36 * 1024) cp.cblock /= 2;
            if (env_cb > 0 && env_cb <= 32 && (size_t)env_cb * plane <= (size_t)160 * 1024) cp.cblock = env_cb;
        }
        cp.wts3 = C.wts3.p;
        cp.unit_f0 = C.unit_f0.d; cp.unit_ql = C.unit_ql.d; cp.unit_woff = C.unit_woff.d; cp.nunits = C.nunits;
        cp.c31tab = C.c31tab.d; cp.c31stride = C.c31stride;
        cp.units_per_block = wgs >= 1024 ? std::max(C.nunits, 1) : std::max(1, (int)((long long)C.nunits * wgs / 1024));
        if (h->cfg.conv_mode == PBD_CONV_MFMA || h->cfg.conv_mode == PBD_CONV_MFMA_F16)
            launch_conv_mfma(cp, h->d_wrec.p, h->cfg.conv_mode == PBD_CONV_MFMA_F16, nb, st);
        else launch_conv(cp, nb, h->f64, st);
    }
}

// frames per DP chunk so that the chunk scratch stays within the budget
int dp_chunk_frames(pbd_handle *h, Plan &P, int want)
{
    const size_t per_frame = (size_t)P.cell_per_frame * std::max(h->JGmax, 1);
    const size_t stk_per_frame = (size_t)P.stk_per_jf * std::max(h->JGmax, 1);
    // bytes of scratch per chunk (12 B / cell-job + 16 B / two stack entries); PBD_DP_BUDGET_MB lets the tests
    // force several chunks on a small batch
    const char *env_budget = getenv("PBD_DP_BUDGET_MB");
    const size_t budget = env_budget && atoll(env_budget) > 0 ? (size_t)atoll(env_budget) << 20 : (size_t)8 << 30;
    int chunk = std::max(want, 1);
    while (chunk > 1 && (per_frame * (6 + 2 * h->rs) + stk_per_frame * (h->f64 ? kStkPairF64 : kStkPairF32)) * chunk > budget) chunk = (chunk + 1) / 2;
    return chunk;
}

int alloc_dp(pbd_handle *h, Plan &P, int nframes, int chunk)
{
    const size_t cpf = (size_t)P.cell_per_frame;
    const int NSa = std::max(h->NS, 1);
    HIPCHK(h, h->acc.ensure(std::max<size_t>((size_t)nframes * cpf * std::max(h->NM, 1) * h->rs, 16)));
    const size_t pes = P.ptr8 ? 1 : 2;        // bytes per position
    HIPCHK(h, h->Ik.ensure(std::max<size_t>((size_t)nframes * cpf * NSa, 16)));
    HIPCHK(h, h->rootv.ensure(std::max<size_t>((size_t)nframes * cpf * h->NC * h->rs, 16)));
    HIPCHK(h, h->rooti.ensure(std::max<size_t>((size_t)nframes * cpf * h->NC * sizeof(int), 16)));
    const size_t per_frame = cpf * std::max(h->JGmax, 1);
    const size_t stk_per_frame = (size_t)P.stk_per_jf * std::max(h->JGmax, 1);
    HIPCHK(h, h->tmp.ensure(std::max<size_t>(per_frame * chunk * h->rs, 16)));
    HIPCHK(h, h->dt.ensure(std::max<size_t>(per_frame * chunk * h->rs, 16) + 32));            // + slack: the combine step reads whole cell groups
    // the transform's pointer planes are kept for the whole batch (one plane per (part, mixture)): the walk composes Ix / Iy from them
    HIPCHK(h, h->IxRaw.ensure(std::max<size_t>((size_t)nframes * cpf * std::max(h->totmix, 1) * pes, 16) + 32));
    HIPCHK(h, h->IyRaw.ensure(std::max<size_t>((size_t)nframes * cpf * std::max(h->totmix, 1) * pes, 16) + 32));
    HIPCHK(h, h->stk.ensure(std::max<size_t>(stk_per_frame * chunk * (h->f64 ? kStkPairF64 : kStkPairF32), 16)));
    return PBD_OK;
}

// dynamic program for frames [f0, f0+nb), nb <= the chunk size given to alloc_dp
void launch_dp_chunk(pbd_handle *h, Plan &P, int f0, int nb, hipStream_t st)
{
    DpParams dp{};
    dp.lv = P.d_lv.d; dp.nlevels = P.nlevels; dp.F = h->F; dp.NS = h->NS; dp.NC = h->NC; dp.NM = h->NM;
    dp.cell_per_frame = P.cell_per_frame; dp.quad_per_frame = P.quad_per_frame; dp.max_mix = h->max_mix;
    dp.resp = h->resp.p; dp.resp_half = h->resp_half ? 1 : 0; dp.acc = h->acc.p;
    dp.Ik = h->Ik.as<uint8_t>(); dp.NJ = h->totmix; dp.ptr8 = P.ptr8 ? 1 : 0;
    dp.tmp = h->tmp.p; dp.dt = h->dt.p;
    dp.IxRaw = h->IxRaw.p; dp.IyRaw = h->IyRaw.p;
    dp.stk = h->stk.p; dp.stk_per_jf = P.stk_per_jf;
    dp.stk_row_off = P.d_stk_row_off.d; dp.stk_col_off = P.d_stk_col_off.d;
    dp.biasw = h->d_biasw.d;
    dp.row2level = P.d_row2level.d; dp.rowoff = P.d_rowoff.d; dp.col2level = P.d_col2level.d; dp.coloff = P.d_coloff.d;
    dp.nrows_flat = P.nrows_flat; dp.ncols_flat = P.ncols_flat; dp.longest = P.longest;
    dp.rootv = h->rootv.p; dp.rooti = h->rooti.as<int>(); dp.rjobs = h->d_rjobs.d;
    dp.frame0 = f0;
    for (auto &g : h->groups) {
        dp.JG = (int)g.jobs.size();
        dp.jobs = g.d_jobs.d; dp.cjobs = g.d_cjobs.d; dp.childs = g.d_childs.d;
        dp.bz_x = g.bz_x; dp.bz_y = g.bz_y;
        { ProfScope ps(h, PBD_K_DT_ROWS, st); launch_dt_rows(dp, nb, h->f64, st); }
        { ProfScope ps(h, PBD_K_DT_COLS, st); launch_dt_cols(dp, nb, h->f64, st); }
        dp.sjobs = g.d_sjobs.d;
        {
            ProfScope ps(h, PBD_K_DP_COMBINE, st);
            if (h->seq_mode) launch_dp_combine_seq(dp, (int)g.sjobs.size(), nb, h->f64, st);
            else launch_dp_combine(dp, (int)g.cjobs.size(), nb, h->f64, st);
        }
    }
    { ProfScope ps(h, PBD_K_DP_ROOT, st); launch_dp_root(dp, nb, h->f64, st); }
}

// single-stream wrappers used by the staged entry points
int run_features(pbd_handle *h, Plan &P, int nframes, int cn)
{
    int rc = alloc_features(h, P, nframes, cn);
    if (rc != PBD_OK) return rc;
    launch_features(h, P, h->frames.p, cn, 0, nframes, h->stream);
    HIPCHK(h, hipGetLastError());
    h->have_features = true;
    return PBD_OK;
}

int run_conv(pbd_handle *h, Plan &P, int nframes)
{
    int rc = alloc_conv(h, P, nframes);
    if (rc != PBD_OK) return rc;
    if ((rc = ensure_seg_tiles(h, P, nframes)) != PBD_OK) return rc;
    launch_conv_stage(h, P, 0, nframes, h->stream);
    HIPCHK(h, hipGetLastError());
    h->have_resp = true;
    return PBD_OK;
}

int run_dp(pbd_handle *h, Plan &P, int nframes)
{
    const int chunk = dp_chunk_frames(h, P, nframes);
    int rc = alloc_dp(h, P, nframes, chunk);
    if (rc != PBD_OK) return rc;
    for (int f0 = 0; f0 < nframes; f0 += chunk) launch_dp_chunk(h, P, f0, std::min(chunk, nframes - f0), h->stream);
    HIPCHK(h, hipGetLastError());
    h->have_dp = true;
    return PBD_OK;
}

// ---- argmin: find (ordered compaction) + walk into a device payload, then one D2H ---------------------------------
// enqueues the find and walk kernels for the `nframes` frames of the device-resident DP result; the candidate list is
// written to d_payload = int32[1 + capacity * stride] (see pbd_handle::CandBuf).  No host synchronisation.
int enqueue_argmin(pbd_handle *h, Plan &P, int nframes, const float *d_scales, int frame_offset, int32_t *d_payload,
                   int capacity, hipStream_t st)
{
    ArgminParams ap{};
    ap.lv = P.d_lv.d; ap.nlevels = P.nlevels; ap.NS = h->NS; ap.NC = h->NC; ap.nframes = nframes;
    ap.cell_per_frame = P.cell_per_frame;
    ap.rootv = h->rootv.p; ap.rooti = h->rooti.as<int>();
    ap.IxRaw = h->IxRaw.p; ap.IyRaw = h->IyRaw.p; ap.NJ = h->totmix; ap.Ik = h->Ik.as<uint8_t>(); ap.ptr8 = P.ptr8 ? 1 : 0;
    ap.thresh = h->thresh; ap.scales = d_scales;
    ap.walk = h->d_walk.d; ap.walk_off = h->d_walk_off.d;
    ap.max_parts = h->max_parts; ap.stride = 8 + 4 * h->max_parts; ap.capacity = std::max(capacity, 0);
    ap.payload = d_payload; ap.frame_offset = frame_offset;
    ap.ntotal = (long long)nframes * P.cell_per_frame * h->NC;
    ap.nblk = (int)std::max<long long>((ap.ntotal + argmin_find_span() - 1) / argmin_find_span(), 1);
    HIPCHK(h, h->find_blk.ensure((size_t)ap.nblk * sizeof(int)));
    ap.blk = h->find_blk.as<int>();
    ProfScope ps(h, PBD_K_ARGMIN, st);
    launch_argmin_find(ap, h->f64, st);
    launch_argmin_walk(ap, h->f64, st);
    return PBD_OK;
}

int candbuf_host(pbd_handle *h, pbd_handle::CandBuf &cb, size_t words)
{
    if (cb.host_words >= words) return PBD_OK;
    if (cb.host) { (void)hipHostFree(cb.host); cb.host = nullptr; cb.host_words = 0; }
    const size_t want = words + words / 4 + 256;
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void **>(&cb.host), want * sizeof(int32_t), hipHostMallocDefault));
    cb.host_words = want;
    return PBD_OK;
}

// find + walk into cb.payload and the speculative read-back of [count | first records], all on `st`
int enqueue_argmin_readback(pbd_handle *h, Plan &P, int nframes, const float *d_scales, pbd_handle::CandBuf &cb, hipStream_t st)
{
    const int stride = 8 + 4 * h->max_parts, cap = std::max(h->cfg.max_candidates, 1);
    HIPCHK(h, cb.payload.ensure(((size_t)cap * stride + 1) * sizeof(int32_t)));
    int rc = enqueue_argmin(h, P, nframes, d_scales, 0, cb.payload.as<int32_t>(), cap, st);
    if (rc != PBD_OK) return rc;
    cb.copied = std::min(h->cand_guess, cap);
    const size_t words = 1 + (size_t)cb.copied * stride;
    // the mirror is sized for twice the guess: growing it (hipHostFree + hipHostMalloc) synchronises the device
    if (cb.host_words < words && (rc = candbuf_host(h, cb, 1 + (size_t)std::min(2 * (long long)cb.copied, (long long)cap) * stride)) != PBD_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(cb.host, cb.payload.p, words * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    return PBD_OK;
}

// after the read-back has completed: hand the records to the caller (the reference's order is nondeterministic,
// src/DynamicProgram.cpp:246-251; here it is (frame, level, component, y, x), produced on the device)
int argmin_deliver(pbd_handle *h, pbd_handle::CandBuf &cb, hipStream_t st, int32_t *cand, int capacity, int *ncand)
{
    const int stride = 8 + 4 * h->max_parts, cap = std::max(h->cfg.max_candidates, 1);
    const int found = cb.host[0];
    const int n = std::min(found, cap);
    if (n > cb.copied) {     // more candidates than the speculative copy covered: fetch the list again, whole
        const size_t words = 1 + (size_t)n * stride;
        const int rc = candbuf_host(h, cb, words);
        if (rc != PBD_OK) return rc;
        HIPCHK(h, hipMemcpyAsync(cb.host, cb.payload.p, words * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        cb.copied = n;
    }
    h->cand_guess = std::min(cap, std::max(n + n / 4 + 256, 1024));
    const int nout = std::min(n, std::max(capacity, 0));
    if (nout > 0) memcpy(cand, cb.host + 1, (size_t)nout * stride * sizeof(int32_t));
    *ncand = nout;
    if (found > cap || n > capacity)
        return fail(h, PBD_ERR_CAPACITY, "%d candidates found, capacity %d (config max_candidates %d)", found, capacity, cap);
    return PBD_OK;
}

int run_argmin(pbd_handle *h, Plan &P, int nframes, const float *d_scales, int32_t *cand, int capacity, int *ncand)
{
    const int rc = enqueue_argmin_readback(h, P, nframes, d_scales, h->cb, h->stream);
    if (rc != PBD_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return argmin_deliver(h, h->cb, h->stream, cand, capacity, ncand);
}

// enqueues pyramid -> HOG -> convolution -> dynamic program for the batch (no host synchronisation); *plan_out = its plan
int enqueue_detect(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int cn, Plan **plan_out)
{
    if (nframes < 1 || nframes > h->cfg.max_batch)
        return fail(h, PBD_ERR_INVALID, "nframes %d outside 1..max_batch %d", nframes, h->cfg.max_batch);
    if (cn != 1 && cn != 3) return fail(h, PBD_ERR_INVALID, "channels %d (1 or 3, src/HOGFeatures.cpp:171)", cn);
    if (!h->bank_matches_model)
        return fail(h, PBD_ERR_STATE, "the filter bank set by setFilters() (%d filters) does not cover the model's filter ids", h->F);
    Plan *P = nullptr;
    int rc = get_image_plan(h, rows, cols, &P);
    if (rc != PBD_OK) return rc;
    h->cur = P; h->cur_frames = nframes; h->cur_cn = cn;
    h->have_features = h->have_resp = h->have_dp = false;
    // Software pipeline over chunks of frames: features + convolution of chunk c run on `stream`, the
    // dynamic program of chunk c-1 on `stream2`.  The convolution is VALU-bound and the distance
    // transform latency-bound, so the two overlap well; buffers are indexed by frame, so chunks never
    // alias, and the DP scratch belongs to stream2 alone.
    // (measured on MI355X: with the convolution holding 127 KB of LDS per CU the DP kernels get too few
    //  waves to profit, so the pipeline is off unless PBD_PIPELINE_CHUNKS asks for it)
    static const int env_chunks = getenv("PBD_PIPELINE_CHUNKS") ? atoi(getenv("PBD_PIPELINE_CHUNKS")) : 1;
    const int want = env_chunks > 1 ? (nframes + env_chunks - 1) / env_chunks : nframes;
    const int chunk = dp_chunk_frames(h, *P, want);
    if ((rc = alloc_features(h, *P, nframes, cn)) != PBD_OK) return rc;
    if ((rc = alloc_conv(h, *P, nframes)) != PBD_OK) return rc;
    if ((rc = ensure_seg_tiles(h, *P, std::min(chunk, nframes))) != PBD_OK) return rc;
    if (nframes % chunk && (rc = ensure_seg_tiles(h, *P, nframes % chunk)) != PBD_OK) return rc;
    if ((rc = alloc_dp(h, *P, nframes, chunk)) != PBD_OK) return rc;
    const int nchunks = (nframes + chunk - 1) / chunk;
    while ((int)h->chunk_events.size() < nchunks + 1) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->chunk_events.push_back(e);
    }
    const bool two = nchunks > 1 && h->stream2 != nullptr;
    hipStream_t sdp = two ? h->stream2 : h->stream;
    if (two) {   // stream2 must not start before earlier work on `stream` (frame upload, previous call)
        HIPCHK(h, hipEventRecord(h->chunk_events[nchunks], h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream2, h->chunk_events[nchunks], 0));
    }
    for (int c = 0; c < nchunks; ++c) {
        const int f0 = c * chunk, nb = std::min(chunk, nframes - f0);
        launch_features(h, *P, d_frames, cn, f0, nb, h->stream);
        launch_conv_stage(h, *P, f0, nb, h->stream);
        if (two) {
            HIPCHK(h, hipEventRecord(h->chunk_events[c], h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->stream2, h->chunk_events[c], 0));
        }
        launch_dp_chunk(h, *P, f0, nb, sdp);
    }
    if (two) {   // join: argmin runs on `stream` after the last DP chunk
        HIPCHK(h, hipEventRecord(h->chunk_events[nchunks], h->stream2));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->chunk_events[nchunks], 0));
    }
    HIPCHK(h, hipGetLastError());
    h->have_features = h->have_resp = h->have_dp = true;
    *plan_out = P;
    return PBD_OK;
}

int detect_device(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int cn, int32_t *cand,
                  int capacity, int *ncand)
{
    if (!h || !cand || !ncand) return PBD_ERR_INVALID;
    if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
    Plan *P = nullptr;
    const int rc = enqueue_detect(h, nframes, d_frames, rows, cols, cn, &P);
    if (rc != PBD_OK) return rc;
    return run_argmin(h, *P, nframes, P->d_scales.d, cand, capacity, ncand);
}

int upload_frames(pbd_handle *h, int nframes, const void *const *imgs, int rows, int cols, int cn, size_t stride_bytes)
{
    const size_t row_bytes = (size_t)cols * cn * depth_size(h->cur_depth);
    if (stride_bytes < row_bytes) return fail(h, PBD_ERR_INVALID, "stride %zu < row bytes %zu", stride_bytes, row_bytes);
    // 32F / 64F images: a NaN or Inf pixel is refused.  The reference computes *something* deterministic from one (NaN
    // gradients, NaN histogram bins, NaN responses whose envelope read-out order then matters); the distance transform here
    // walks the envelope top-down, which equals the reference's bottom-up walk only for strictly increasing finite
    // intersections -- so non-finite input is defined as an error instead of being allowed to differ silently.
    if (h->cur_depth == kDepth32F || h->cur_depth == kDepth64F) {
        const size_t n = (size_t)cols * cn;
        for (int i = 0; i < nframes; ++i)
            for (int y = 0; y < rows; ++y) {
                const char *row = static_cast<const char *>(imgs[i]) + (size_t)y * stride_bytes;
                bool ok = true;
                if (h->cur_depth == kDepth32F) { const float *p = reinterpret_cast<const float *>(row); for (size_t k = 0; k < n; ++k) ok = ok && std::isfinite(p[k]); }
                else { const double *p = reinterpret_cast<const double *>(row); for (size_t k = 0; k < n; ++k) ok = ok && std::isfinite(p[k]); }
                if (!ok) return fail(h, PBD_ERR_INVALID, "frame %d, row %d holds a NaN or Inf pixel", i, y);
            }
    }
    HIPCHK(h, h->frames.ensure((size_t)nframes * rows * row_bytes + 4));
    for (int i = 0; i < nframes; ++i)
        HIPCHK(h, hipMemcpy2DAsync(h->frames.as<uint8_t>() + (size_t)i * rows * row_bytes, row_bytes, imgs[i], stride_bytes,
                                   row_bytes, rows, hipMemcpyHostToDevice, h->stream));
    return PBD_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char *pbd_version(void) { return "pbd-hip 0.1 (gfx950)"; }

// diagnostics, not part of include/pbd.h
int pbd_debug_conv_occupancy(int nw) { return nw == 5 ? conv_mfma_occupancy(false) : nw == 6 ? conv_mfma_occupancy(true) : conv_occupancy(nw); }
// the convolution's tile cover of one rows x cols level (host-only, no GPU needed): out[i] = {shape, y0, x0}
int pbd_debug_cover_level(int rows, int cols, int *out, int capacity)
{
    return guarded(nullptr, [&]() -> int {
        std::vector<ConvTile> shaped[3];
        cover_level(0, rows, cols, shaped);
        int n = 0;
        for (int k = 0; k < 3; ++k)
            for (const ConvTile &t : shaped[k]) {
                if (n < capacity) { out[3 * n] = k; out[3 * n + 1] = t.y0; out[3 * n + 2] = t.x0; }
                ++n;
            }
        return n;
    });
}

// the exact convolution's strip-sequence tiles for `nb` frames of the given feature-map sizes (host-only, no GPU needed):
// out[i] = {nseg, len0, len1, len2, then per segment frame, level, strip, x0} = 16 ints per tile; returns the tile count
int pbd_debug_seg_tiles(int nlevels, const int *rows, const int *cols, int nb, int *out, int capacity)
{
    return guarded(nullptr, [&]() -> int {
        std::vector<LevelDesc> lv(nlevels);
        for (int l = 0; l < nlevels; ++l) { memset(&lv[l], 0, sizeof lv[l]); lv[l].rows = rows[l]; lv[l].cols = cols[l]; }
        std::vector<ConvSegTile> tiles;
        build_seg_tiles(lv, nb, tiles);
        static_assert(sizeof(ConvSegTile) == 16 * sizeof(int), "a tile record is 16 ints");
        for (size_t i = 0; i < tiles.size() && (int)i < capacity; ++i) memcpy(out + 16 * i, &tiles[i], sizeof(ConvSegTile));
        return (int)tiles.size();
    });
}

// runs a body that throws inside the ABI guard (host-only): 0 = std::bad_alloc, 1 = std::length_error from an absurd
// std::vector size, 2 = another std::exception; returns the status code the guard produced
int pbd_debug_guard_selftest(int kind)
{
    return guarded(nullptr, [&]() -> int {
        if (kind == 0) throw std::bad_alloc();
        if (kind == 1) { std::vector<int32_t> v; v.resize(v.max_size() + (size_t)1); return (int)v.size(); }
        if (kind == 2) throw std::runtime_error("selftest");
        return PBD_OK;
    });
}

const char *pbd_last_error(const pbd_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pbd_create(const pbd_model *model, const pbd_config *config, pbd_handle **out)
{
    return guarded(nullptr, [&]() -> int {
        if (!model || !config || !out) return fail(nullptr, PBD_ERR_INVALID, "null argument");
        *out = nullptr;
        if (config->real_type != PBD_REAL_F32 && config->real_type != PBD_REAL_F64)
            return fail(nullptr, PBD_ERR_UNSUPPORTED, "real_type %d: PBD_REAL_F32 or PBD_REAL_F64", config->real_type);
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev < 1)
            return fail(nullptr, PBD_ERR_HIP, "no HIP device available (%s); this library has no CPU path",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        if (config->device < 0 || config->device >= ndev) return fail(nullptr, PBD_ERR_INVALID, "device %d of %d", config->device, ndev);
        e = hipSetDevice(config->device);
        if (e != hipSuccess) return fail(nullptr, PBD_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
        std::unique_ptr<pbd_handle, void (*)(pbd_handle *)> h(new pbd_handle, pbd_destroy);   // releases device memory on every exit
        h->cfg = *config;
        h->f64 = config->real_type == PBD_REAL_F64;
        h->rs = h->f64 ? sizeof(double) : sizeof(float);
        h->resp_half = config->conv_mode == PBD_CONV_MFMA_F16 && !h->f64;
        h->resp_es = h->resp_half ? 2 : h->rs;
        if (h->cfg.max_batch < 1) h->cfg.max_batch = 1;
        if (h->cfg.max_candidates < 1) h->cfg.max_candidates = 65536;
        if (config->stream) {
            h->stream = reinterpret_cast<hipStream_t>(config->stream);
        } else {
            e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
            if (e != hipSuccess) return fail(nullptr, PBD_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
            h->own_stream = true;
        }
        e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
        if (e != hipSuccess) return fail(nullptr, PBD_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        int rc = build_model(h.get(), model);
        if (rc != PBD_OK) {
            g_create_error = h->err;
            return rc;
        }
        *out = h.release();
        return PBD_OK;
    });
}

void pbd_destroy(pbd_handle *h)
{
    if (!h) return;
    try {
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    h->prof.release();
    if (h->stream_copy) (void)hipStreamSynchronize(h->stream_copy);
    if (h->stream_d2h) (void)hipStreamSynchronize(h->stream_d2h);
    for (pbd_handle::Slot &S : h->slot) {
        if (S.pinned) (void)hipHostFree(S.pinned);
        if (S.cb.host) (void)hipHostFree(S.cb.host);
        if (S.copied) (void)hipEventDestroy(S.copied);
        if (S.done) (void)hipEventDestroy(S.done);
        S.frames.release(); S.cb.payload.release();
    }
    if (h->cb.host) (void)hipHostFree(h->cb.host);
    h->cb.payload.release();
    if (h->stream_copy) (void)hipStreamDestroy(h->stream_copy);
    if (h->stream_d2h) (void)hipStreamDestroy(h->stream_d2h);
    for (auto e : h->chunk_events) (void)hipEventDestroy(e);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    for (DevBuf *b : {&h->frames, &h->pyr, &h->gmag, &h->gori, &h->hist, &h->norm, &h->feat, &h->resp, &h->acc, &h->Ik, &h->rootv,
                      &h->rooti, &h->tmp, &h->dt, &h->IxRaw, &h->IyRaw, &h->stk, &h->find_blk,
                      &h->scales_tmp})
        b->release();
    for (auto &c : h->conv_classes) { c.wts.release(); c.fmap.release(); c.wts3.release(); c.unit_f0.release(); c.unit_ql.release(); c.unit_woff.release(); c.c31tab.release(); }
    h->d_wrec.release(); h->d_biasw.release(); h->d_coord.release(); h->d_walk_off.release();
    h->d_rjobs.release(); h->d_walk.release();
    for (auto &g : h->groups) { g.d_jobs.release(); g.d_childs.release(); g.d_cjobs.release(); g.d_sjobs.release(); }
    h->plans.clear();
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    } catch (...) {
    }
    delete h;
}

int pbd_candidate_stride(const pbd_handle *h) { return h ? 8 + 4 * h->max_parts : 0; }
int pbd_binsize(const pbd_handle *h) { return h ? h->sbin : 0; }
int pbd_num_ptr_slots(const pbd_handle *h) { return h ? h->NS : 0; }
int pbd_ptr_slot(const pbd_handle *h, int component, int part)
{
    if (!h || component < 0 || component >= h->NC) return -1;
    const int p0 = h->part_offset[component];
    if (part < 0 || p0 + part >= h->part_offset[component + 1]) return -1;
    return h->ptr_slot[p0 + part];
}

int pbd_set_level_shard(pbd_handle *h, int rank, int world)
{
    return guarded(h, [&]() -> int {
        if (!h) return PBD_ERR_INVALID;
        if (world < 1 || rank < 0 || rank >= world) return fail(h, PBD_ERR_INVALID, "level shard %d of %d", rank, world);
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        if (rank == h->shard_rank && world == h->shard_world) return PBD_OK;
        (void)hipSetDevice(h->cfg.device);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->stream2) HIPCHK(h, hipStreamSynchronize(h->stream2));
        h->shard_rank = rank; h->shard_world = world;
        // image plans depend on the shard
        h->cur = nullptr; h->have_features = h->have_resp = h->have_dp = false;
        for (size_t i = 0; i < h->plans.size();)
            if (h->plans[i]->kind == 0) h->plans.erase(h->plans.begin() + i); else ++i;
        return PBD_OK;
    });
}

int pbd_pyramid_plan(pbd_handle *h, int rows, int cols, int *nlevels, int *img_rows, int *img_cols, int *feat_rows,
                     int *feat_cols, float *scales)
{
    return guarded(h, [&]() -> int {
        if (!h || !nlevels) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        Plan *P = nullptr;
        int rc = get_image_plan(h, rows, cols, &P);
        if (rc != PBD_OK) return rc;
        *nlevels = P->nlevels;
        for (int l = 0; l < P->nlevels; ++l) {
            if (img_rows) img_rows[l] = P->lv[l].img_rows;
            if (img_cols) img_cols[l] = P->lv[l].img_cols;
            if (feat_rows) feat_rows[l] = P->lv[l].rows;
            if (feat_cols) feat_cols[l] = P->lv[l].cols;
            if (scales) scales[l] = P->scales[l];
        }
        return PBD_OK;
    });
}

int pbd_features_pyramid(pbd_handle *h, const void *img, int rows, int cols, int channels, size_t stride_bytes,
                         int depth_code, void *const *feat)
{
    return guarded(h, [&]() -> int {
        if (!h || !img || !feat) return PBD_ERR_INVALID;
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        (void)hipSetDevice(h->cfg.device);
        if (!depth_size(depth_code))
            return fail(h, PBD_ERR_UNSUPPORTED, "image depth code %d: 0 (8U), 2 (16U), 5 (32F) or 6 (64F), src/HOGFeatures.cpp:136-146", depth_code);
        h->cur_depth = depth_code;
        if (channels != 1 && channels != 3) return fail(h, PBD_ERR_INVALID, "channels %d (1 or 3, src/HOGFeatures.cpp:171)", channels);
        Plan *P = nullptr;
        int rc = get_image_plan(h, rows, cols, &P);
        if (rc != PBD_OK) return rc;
        if ((rc = upload_frames(h, 1, &img, rows, cols, channels, stride_bytes)) != PBD_OK) return rc;
        h->cur = P; h->cur_frames = 1; h->cur_cn = channels;
        h->have_features = h->have_resp = h->have_dp = false;
        if ((rc = run_features(h, *P, 1, channels)) != PBD_OK) return rc;
        for (int l = 0; l < P->nlevels; ++l) {
            const LevelDesc &d = P->lv[l];
            const size_t n = (size_t)d.rows * d.cols * 32;
            if (n && feat[l])
                HIPCHK(h, hipMemcpyAsync(feat[l], h->feat.as<char>() + (size_t)d.cell_off * 32 * h->rs, n * h->rs,
                                         hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PBD_OK;
    });
}

int pbd_get_pyramid_image(pbd_handle *h, int frame, int level, uint8_t *dst)
{
    return guarded(h, [&]() -> int {
        if (!h || !dst) return PBD_ERR_INVALID;
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        (void)hipSetDevice(h->cfg.device);
        if (!h->cur || h->cur->kind != 0 || !h->have_features) return fail(h, PBD_ERR_STATE, "no pyramid has been computed");
        Plan &P = *h->cur;
        if (frame < 0 || frame >= h->cur_frames || level < 0 || level >= P.nlevels) return fail(h, PBD_ERR_INVALID, "frame/level out of range");
        const LevelDesc &d = P.lv[level];
        const size_t es = depth_size(h->cur_depth);
        HIPCHK(h, hipMemcpyAsync(dst, h->pyr.as<uint8_t>() + ((size_t)frame * P.pix_per_frame + d.img_off) * h->cur_cn * es,
                                 (size_t)d.img_rows * d.img_cols * h->cur_cn * es, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PBD_OK;
    });
}

int pbd_conv_set_filters(pbd_handle *h, int nfilters, const void *const *filters, const int *ksize)
{
    return guarded(h, [&]() -> int {
        if (!h || !filters || !ksize) return PBD_ERR_INVALID;
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        (void)hipSetDevice(h->cfg.device);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->stream2) HIPCHK(h, hipStreamSynchronize(h->stream2));
        const int rc = upload_filters(h, nfilters, filters, ksize);
        if (rc != PBD_OK) return rc;
        h->have_resp = h->have_dp = false;   // staged results of the old bank are gone
        return revalidate_bank(h);
    });
}

int pbd_conv_pdf(pbd_handle *h, int nlevels, const void *const *feat, const int *rows, const int *cols, void *const *resp)
{
    return guarded(h, [&]() -> int {
        if (!h || !feat || !rows || !cols || !resp) return PBD_ERR_INVALID;
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        (void)hipSetDevice(h->cfg.device);
        Plan *P = nullptr;
        int rc = get_dims_plan(h, nlevels, rows, cols, &P);
        if (rc != PBD_OK) return rc;
        HIPCHK(h, h->feat.ensure(std::max<size_t>((size_t)P->cell_per_frame * 32 * h->rs, 16)));
        for (int l = 0; l < nlevels; ++l) {
            const size_t n = (size_t)rows[l] * cols[l] * 32;
            if (n) HIPCHK(h, hipMemcpyAsync(h->feat.as<char>() + (size_t)P->lv[l].cell_off * 32 * h->rs, feat[l], n * h->rs,
                                            hipMemcpyHostToDevice, h->stream));
        }
        h->cur = P; h->cur_frames = 1; h->have_features = true; h->have_resp = h->have_dp = false;
        h->feat_c31_zero = false;
        if ((rc = run_conv(h, *P, 1)) != PBD_OK) return rc;
        std::vector<uint16_t> halfbuf;
        for (int l = 0; l < nlevels; ++l) {
            const size_t n = (size_t)rows[l] * cols[l] * h->F;
            if (!n) continue;
            const char *src = h->resp.as<char>() + (size_t)P->lv[l].cell_off * h->F * h->resp_es;
            if (h->resp_half) {      // fp16 on the device, T = float at the seam
                halfbuf.resize(n);
                HIPCHK(h, hipMemcpyAsync(halfbuf.data(), src, n * 2, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                float *dst = static_cast<float *>(resp[l]);
                for (size_t i = 0; i < n; ++i) dst[i] = host_h2f(halfbuf[i]);
            } else {
                HIPCHK(h, hipMemcpyAsync(resp[l], src, n * h->rs, hipMemcpyDeviceToHost, h->stream));
            }
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PBD_OK;
    });
}

int pbd_dp_min(pbd_handle *h, int nlevels, const int *rows, const int *cols, const void *const *resp, int32_t *const *Ix,
               int32_t *const *Iy, int32_t *const *Ik, void *const *rootv, int32_t *const *rooti)
{
    return guarded(h, [&]() -> int {
        if (!h || !rows || !cols || !resp) return PBD_ERR_INVALID;
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        (void)hipSetDevice(h->cfg.device);
        Plan *P = nullptr;
        if (!h->bank_matches_model)
            return fail(h, PBD_ERR_STATE, "the filter bank set by setFilters() (%d filters) does not cover the model's filter ids", h->F);
        int rc = get_dims_plan(h, nlevels, rows, cols, &P);
        if (rc != PBD_OK) return rc;
        HIPCHK(h, h->resp.ensure(std::max<size_t>((size_t)P->cell_per_frame * h->F * h->resp_es, 16) + 32));
        std::vector<uint16_t> halfbuf;
        for (int l = 0; l < nlevels; ++l) {
            const size_t n = (size_t)rows[l] * cols[l] * h->F;
            if (!n) continue;
            char *dst = h->resp.as<char>() + (size_t)P->lv[l].cell_off * h->F * h->resp_es;
            if (h->resp_half) {      // the device side of this mode reads fp16 responses (exact for what pbd_conv_pdf returned)
                halfbuf.resize(n);
                const float *src = static_cast<const float *>(resp[l]);
                for (size_t i = 0; i < n; ++i) halfbuf[i] = host_f2h(src[i]);
                HIPCHK(h, hipMemcpy(dst, halfbuf.data(), n * 2, hipMemcpyHostToDevice));
            } else {
                HIPCHK(h, hipMemcpyAsync(dst, resp[l], n * h->rs, hipMemcpyHostToDevice, h->stream));
            }
        }
        h->cur = P; h->cur_frames = 1; h->have_resp = true; h->have_dp = false;
        if ((rc = run_dp(h, *P, 1)) != PBD_OK) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<uint8_t> t8;
        for (int l = 0; l < nlevels; ++l) {
            const size_t hw = (size_t)rows[l] * cols[l];
            if (!hw) continue;
            const size_t n = hw * h->NS, off = (size_t)P->lv[l].cell_off * h->NS;
            if (n) {
                t8.resize(n);
                // the device keeps the winning mixture per slot (Ik) and the transform's own pointer planes per (part, mixture);
                // the reference's Ix / Iy of a slot are composed here: Ix = IxRaw[k][y][x], Iy = IyRaw[k][y][Ix], k = Ik
                HIPCHK(h, hipMemcpy(t8.data(), h->Ik.as<uint8_t>() + off, n, hipMemcpyDeviceToHost));
                std::vector<uint8_t> ik(t8.begin(), t8.begin() + n);
                if (Ik && Ik[l]) for (size_t i = 0; i < n; ++i) Ik[l][i] = ik[i];
                if ((Ix && Ix[l]) || (Iy && Iy[l])) {
                    const size_t nj = hw * h->totmix, joff = (size_t)P->lv[l].cell_off * h->totmix;
                    std::vector<int> px(nj), py(nj);
                    auto fetch_planes = [&](const DevBuf &src, std::vector<int> &dst) -> hipError_t {
                        if (P->ptr8) {
                            std::vector<uint8_t> b(nj);
                            hipError_t e = hipMemcpy(b.data(), src.as<uint8_t>() + joff, nj, hipMemcpyDeviceToHost);
                            if (e != hipSuccess) return e;
                            for (size_t i = 0; i < nj; ++i) dst[i] = b[i];
                            return hipSuccess;
                        }
                        std::vector<int16_t> b(nj);
                        hipError_t e = hipMemcpy(b.data(), src.as<int16_t>() + joff, nj * 2, hipMemcpyDeviceToHost);
                        if (e != hipSuccess) return e;
                        for (size_t i = 0; i < nj; ++i) dst[i] = b[i];
                        return hipSuccess;
                    };
                    HIPCHK(h, fetch_planes(h->IxRaw, px));
                    HIPCHK(h, fetch_planes(h->IyRaw, py));
                    const int Wl = cols[l];
                    const int totparts = (int)h->parentid.size();
                    for (int gp = 0; gp < totparts; ++gp) {
                        bool root = false;
                        for (int c = 0; c < h->NC; ++c) root = root || gp == h->part_offset[c];
                        if (root) continue;
                        int c = 0;
                        while (c + 1 < h->NC && h->part_offset[c + 1] <= gp) ++c;
                        const int gpar = h->part_offset[c] + h->parentid[gp];
                        const int L = h->mix_offset[gpar + 1] - h->mix_offset[gpar];
                        for (int pm = 0; pm < L; ++pm) {
                            const size_t so = (size_t)(h->ptr_slot[gp] + pm) * hw;
                            for (size_t cell = 0; cell < hw; ++cell) {
                                const size_t plane = (size_t)(h->mix_offset[gp] + ik[so + cell]) * hw;
                                const int x = px[plane + (cell % Wl) * (size_t)rows[l] + cell / Wl];      // IxRaw is kept transposed
                                if (Ix && Ix[l]) Ix[l][so + cell] = x;
                                if (Iy && Iy[l]) Iy[l][so + cell] = py[plane + (cell / Wl) * Wl + x];
                            }
                        }
                    }
                }
            }
            const size_t roff = (size_t)P->lv[l].cell_off * h->NC;
            if (rootv && rootv[l]) HIPCHK(h, hipMemcpy(rootv[l], h->rootv.as<char>() + roff * h->rs, hw * h->NC * h->rs, hipMemcpyDeviceToHost));
            if (rooti && rooti[l]) HIPCHK(h, hipMemcpy(rooti[l], h->rooti.as<int>() + roff, hw * h->NC * sizeof(int), hipMemcpyDeviceToHost));
        }
        return PBD_OK;
    });
}

int pbd_dp_argmin(pbd_handle *h, const float *scales, int32_t *cand, int capacity, int *ncand)
{
    return guarded(h, [&]() -> int {
        if (!h || !scales || !cand || !ncand) return PBD_ERR_INVALID;
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        (void)hipSetDevice(h->cfg.device);
        if (!h->cur || !h->have_dp) return fail(h, PBD_ERR_STATE, "argmin() before min()");
        Plan &P = *h->cur;
        HIPCHK(h, h->scales_tmp.ensure(sizeof(float) * PBD_MAX_LEVELS));
        HIPCHK(h, hipMemcpyAsync(h->scales_tmp.p, scales, sizeof(float) * P.nlevels, hipMemcpyHostToDevice, h->stream));
        return run_argmin(h, P, h->cur_frames, h->scales_tmp.as<float>(), cand, capacity, ncand);
    });
}

int pbd_detect(pbd_handle *h, const void *img, int rows, int cols, int channels, size_t stride_bytes, int32_t *cand,
               int capacity, int *ncand)
{
    return pbd_detect_batch(h, 1, &img, rows, cols, channels, stride_bytes, cand, capacity, ncand);
}

int pbd_detect_batch(pbd_handle *h, int nframes, const void *const *imgs, int rows, int cols, int channels,
                     size_t stride_bytes, int32_t *cand, int capacity, int *ncand)
{
    return guarded(h, [&]() -> int {
        if (!h || !imgs || !cand || !ncand) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        if (nframes < 1 || nframes > h->cfg.max_batch)
            return fail(h, PBD_ERR_INVALID, "nframes %d outside 1..max_batch %d", nframes, h->cfg.max_batch);
        if (channels != 1 && channels != 3) return fail(h, PBD_ERR_INVALID, "channels %d (1 or 3, src/HOGFeatures.cpp:171)", channels);
        h->cur_depth = kDepth8U;
        int rc = upload_frames(h, nframes, imgs, rows, cols, channels, stride_bytes);
        if (rc != PBD_OK) return rc;
        return detect_device(h, nframes, h->frames.p, rows, cols, channels, cand, capacity, ncand);
    });
}

// what submit() does once the frames of the batch are (being) made resident at d_frames: the whole path, the candidates'
// read-back and the slot's completion event, all enqueued on the handle's stream without waiting
static int submit_enqueue(pbd_handle *h, pbd_handle::Slot &S, int nframes, const void *d_frames, int rows, int cols, int channels)
{
    if (!S.done) HIPCHK(h, hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
    Plan *P = nullptr;
    h->cur_depth = kDepth8U;
    int rc = enqueue_detect(h, nframes, d_frames, rows, cols, channels, &P);
    if (rc != PBD_OK) return rc;
    if ((rc = enqueue_argmin_readback(h, *P, nframes, P->d_scales.d, S.cb, h->stream)) != PBD_OK) return rc;
    HIPCHK(h, hipEventRecord(S.done, h->stream));
    HIPCHK(h, hipGetLastError());
    S.plan = P; S.nframes = nframes;
    h->nsubmitted += 1;
    return PBD_OK;
}

int pbd_detect_batch_submit(pbd_handle *h, int nframes, const void *const *imgs, int rows, int cols, int channels,
                            size_t stride_bytes)
{
    return guarded(h, [&]() -> int {
        if (!h || !imgs) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        if (h->nsubmitted - h->nwaited >= 2) return fail(h, PBD_ERR_STATE, "two batches are already in flight: call pbd_detect_batch_wait first");
        if (nframes < 1 || nframes > h->cfg.max_batch)
            return fail(h, PBD_ERR_INVALID, "nframes %d outside 1..max_batch %d", nframes, h->cfg.max_batch);
        if (channels != 1 && channels != 3) return fail(h, PBD_ERR_INVALID, "channels %d (1 or 3, src/HOGFeatures.cpp:171)", channels);
        const size_t row_bytes = (size_t)cols * channels, frame_bytes = row_bytes * rows, bytes = frame_bytes * nframes;
        if (stride_bytes < row_bytes) return fail(h, PBD_ERR_INVALID, "stride %zu < row bytes %zu", stride_bytes, row_bytes);
        pbd_handle::Slot &S = h->slot[h->nsubmitted & 1];
        if (!h->stream_copy) HIPCHK(h, hipStreamCreateWithFlags(&h->stream_copy, hipStreamNonBlocking));
        if (!S.copied) HIPCHK(h, hipEventCreateWithFlags(&S.copied, hipEventDisableTiming));
        if (S.pinned_cap < bytes) {
            if (S.pinned) { (void)hipHostFree(S.pinned); S.pinned = nullptr; S.pinned_cap = 0; }
            HIPCHK(h, hipHostMalloc(&S.pinned, bytes + bytes / 8, hipHostMallocDefault));
            S.pinned_cap = bytes + bytes / 8;
        }
        HIPCHK(h, S.frames.ensure(bytes));
        // host staging (this is what overlaps the kernels of the batch submitted before), then one asynchronous copy
        for (int i = 0; i < nframes; ++i) {
            char *dst = static_cast<char *>(S.pinned) + (size_t)i * frame_bytes;
            const char *src = static_cast<const char *>(imgs[i]);
            if (stride_bytes == row_bytes) memcpy(dst, src, frame_bytes);
            else for (int y = 0; y < rows; ++y) memcpy(dst + (size_t)y * row_bytes, src + (size_t)y * stride_bytes, row_bytes);
        }
        HIPCHK(h, hipMemcpyAsync(S.frames.p, S.pinned, bytes, hipMemcpyHostToDevice, h->stream_copy));
        HIPCHK(h, hipEventRecord(S.copied, h->stream_copy));
        HIPCHK(h, hipStreamWaitEvent(h->stream, S.copied, 0));
        return submit_enqueue(h, S, nframes, S.frames.p, rows, cols, channels);
    });
}

int pbd_detect_batch_device_submit(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int channels)
{
    return guarded(h, [&]() -> int {
        if (!h || !d_frames) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        if (h->nsubmitted - h->nwaited >= 2) return fail(h, PBD_ERR_STATE, "two batches are already in flight: call pbd_detect_batch_wait first");
        return submit_enqueue(h, h->slot[h->nsubmitted & 1], nframes, d_frames, rows, cols, channels);
    });
}

int pbd_detect_batch_wait(pbd_handle *h, int32_t *cand, int capacity, int *ncand)
{
    return guarded(h, [&]() -> int {
        if (!h || !cand || !ncand) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        if (h->nsubmitted == h->nwaited) return fail(h, PBD_ERR_STATE, "no batch in flight");
        pbd_handle::Slot &S = h->slot[h->nwaited & 1];
        h->nwaited += 1;                       // the slot is released whatever happens below
        HIPCHK(h, hipEventSynchronize(S.done));
        if (!h->stream_d2h) HIPCHK(h, hipStreamCreateWithFlags(&h->stream_d2h, hipStreamNonBlocking));
        // (a list longer than the speculative copy is fetched on a stream of its own: the compute stream may already hold
        //  the next batch, whose kernels this copy must not queue behind -- and they do not touch this slot's payload)
        return argmin_deliver(h, S.cb, h->stream_d2h, cand, capacity, ncand);
    });
}

// Device-resident output (new surface, for multi-GPU jobs and device pipelines): the whole path with the candidate list
// left ON THE DEVICE in the caller's buffer; asynchronous.  See include/pbd.h.
int pbd_detect_batch_device_out(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int channels,
                                int frame_offset, int32_t *d_payload, int capacity)
{
    return guarded(h, [&]() -> int {
        if (!h || !d_frames || !d_payload) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        if (capacity < 0) return fail(h, PBD_ERR_INVALID, "capacity %d", capacity);
        h->cur_depth = kDepth8U;
        Plan *P = nullptr;
        int rc = enqueue_detect(h, nframes, d_frames, rows, cols, channels, &P);
        if (rc != PBD_OK) return rc;
        if ((rc = enqueue_argmin(h, *P, nframes, P->d_scales.d, frame_offset, d_payload, capacity, h->stream)) != PBD_OK) return rc;
        HIPCHK(h, hipGetLastError());
        return PBD_OK;
    });
}

int pbd_argmin_device_out(pbd_handle *h, int frame_offset, int32_t *d_payload, int capacity)
{
    return guarded(h, [&]() -> int {
        if (!h || !d_payload) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        if (!h->cur || !h->have_dp || h->cur->kind != 0) return fail(h, PBD_ERR_STATE, "no detect result is resident on the device");
        if (capacity < 0) return fail(h, PBD_ERR_INVALID, "capacity %d", capacity);
        const int rc = enqueue_argmin(h, *h->cur, h->cur_frames, h->cur->d_scales.d, frame_offset, d_payload, capacity, h->stream);
        if (rc != PBD_OK) return rc;
        HIPCHK(h, hipGetLastError());
        return PBD_OK;
    });
}

void *pbd_stream(const pbd_handle *h) { return h ? reinterpret_cast<void *>(h->stream) : nullptr; }

int pbd_detect_batch_device(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int channels,
                            int32_t *cand, int capacity, int *ncand)
{
    return guarded(h, [&]() -> int {
        if (!h || !d_frames) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        h->cur_depth = kDepth8U;
        return detect_device(h, nframes, d_frames, rows, cols, channels, cand, capacity, ncand);
    });
}

int pbd_detect_typed(pbd_handle *h, const void *img, int rows, int cols, int channels, size_t stride_bytes, int depth_code,
                     int32_t *cand, int capacity, int *ncand)
{
    return guarded(h, [&]() -> int {
        if (!h || !img || !cand || !ncand) return PBD_ERR_INVALID;
        (void)hipSetDevice(h->cfg.device);
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        if (!depth_size(depth_code))
            return fail(h, PBD_ERR_UNSUPPORTED, "image depth code %d: 0 (8U), 2 (16U), 5 (32F) or 6 (64F), src/HOGFeatures.cpp:136-146", depth_code);
        if (channels != 1 && channels != 3) return fail(h, PBD_ERR_INVALID, "channels %d (1 or 3, src/HOGFeatures.cpp:171)", channels);
        h->cur_depth = depth_code;
        int rc = upload_frames(h, 1, &img, rows, cols, channels, stride_bytes);
        if (rc != PBD_OK) return rc;
        return detect_device(h, 1, h->frames.p, rows, cols, channels, cand, capacity, ncand);
    });
}

int pbd_get_stage(pbd_handle *h, int stage, int frame, int level, void *dst, size_t dst_bytes)
{
    return guarded(h, [&]() -> int {
        if (!h || !dst) return PBD_ERR_INVALID;
        if (h->nsubmitted != h->nwaited) return fail(h, PBD_ERR_STATE, "a submitted batch has not been waited for");
        (void)hipSetDevice(h->cfg.device);
        if (!h->cur) return fail(h, PBD_ERR_STATE, "nothing has been computed");
        Plan &P = *h->cur;
        if (frame < 0 || frame >= h->cur_frames || level < 0 || level >= P.nlevels) return fail(h, PBD_ERR_INVALID, "frame/level out of range");
        const LevelDesc &d = P.lv[level];
        const size_t hw = (size_t)d.rows * d.cols, cpf = (size_t)P.cell_per_frame;
        const void *src = nullptr;
        size_t bytes = 0;
        switch (stage) {
        case PBD_STAGE_FEATURES:
            if (!h->have_features) return fail(h, PBD_ERR_STATE, "features not computed");
            src = h->feat.as<char>() + ((size_t)frame * cpf + d.cell_off) * 32 * h->rs; bytes = hw * 32 * h->rs; break;
        case PBD_STAGE_RESPONSES:
            if (!h->have_resp) return fail(h, PBD_ERR_STATE, "responses not computed");
            src = h->resp.as<char>() + ((size_t)frame * cpf + d.cell_off) * h->F * h->resp_es; bytes = hw * h->F * h->rs; break;
        case PBD_STAGE_ROOTV:
            if (!h->have_dp) return fail(h, PBD_ERR_STATE, "dp not computed");
            src = h->rootv.as<char>() + ((size_t)frame * cpf + d.cell_off) * h->NC * h->rs; bytes = hw * h->NC * h->rs; break;
        case PBD_STAGE_ROOTI:
            if (!h->have_dp) return fail(h, PBD_ERR_STATE, "dp not computed");
            src = h->rooti.as<int>() + ((size_t)frame * cpf + d.cell_off) * h->NC; bytes = hw * h->NC * sizeof(int); break;
        default: return fail(h, PBD_ERR_INVALID, "unknown stage %d", stage);
        }
        if (dst_bytes < bytes) return fail(h, PBD_ERR_INVALID, "destination holds %zu bytes, need %zu", dst_bytes, bytes);
        if (stage == PBD_STAGE_RESPONSES && h->resp_half && bytes) {     // fp16 on the device -> T = float
            const size_t n = bytes / sizeof(float);
            std::vector<uint16_t> halfbuf(n);
            HIPCHK(h, hipMemcpyAsync(halfbuf.data(), src, n * 2, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            float *out = static_cast<float *>(dst);
            for (size_t i = 0; i < n; ++i) out[i] = host_h2f(halfbuf[i]);
            return PBD_OK;
        }
        if (bytes) HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return PBD_OK;
    });
}

int pbd_profile_enable(pbd_handle *h, int on)
{
    return guarded(h, [&]() -> int {
        if (!h) return PBD_ERR_INVALID;
        h->prof.flush();
        h->prof.on = on == 2 ? 2 : (on != 0 ? 1 : 0);
        return PBD_OK;
    });
}
int pbd_profile_reset(pbd_handle *h)
{
    return guarded(h, [&]() -> int {
        if (!h) return PBD_ERR_INVALID;
        h->prof.flush();
        for (int k = 0; k < PBD_K_COUNT; ++k) { h->prof.total[k] = 0; h->prof.launches[k] = 0; }
        return PBD_OK;
    });
}
int pbd_profile_read(pbd_handle *h, int k, double *total_ms, int *launches)
{
    return guarded(h, [&]() -> int {
        if (!h || k < 0 || k >= PBD_K_COUNT) return PBD_ERR_INVALID;
        h->prof.flush();
        if (total_ms) *total_ms = h->prof.total[k];
        if (launches) *launches = h->prof.launches[k];
        return PBD_OK;
    });
}
const char *pbd_kernel_name(int k)
{
    static const char *names[PBD_K_COUNT] = {"k_resize", "k_pyrdown", "k_hog_hist", "k_hog_feat", "k_conv", "k_dt_rows",
                                             "k_dt_cols", "k_dp_combine", "k_dp_root", "k_argmin"};
    return (k >= 0 && k < PBD_K_COUNT) ? names[k] : "?";
}
int pbd_synchronize(pbd_handle *h)
{
    return guarded(h, [&]() -> int {
        if (!h) return PBD_ERR_INVALID;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->stream2) HIPCHK(h, hipStreamSynchronize(h->stream2));
        return PBD_OK;
    });
}

}  // extern "C"
