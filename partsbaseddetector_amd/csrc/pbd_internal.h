// pbd_internal.h -- shared declarations of the HIP implementation behind include/pbd.h.
// gfx950 (MI355X) only.  Not part of the public interface.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pbd.h"

namespace pbd {

// One pyramid level as the kernels see it.  Offsets are per frame; a frame's slice of a buffer
// starts at frame * (per-frame total).
struct LevelDesc {
    int img_rows, img_cols;   // level image (pixels)
    int blk_rows, blk_cols;   // HOG blocks = round(dim / sbin)             (src/HOGFeatures.cpp:174)
    int rows, cols;           // feature / response map = blocks - 2         (:175)
    int src_level;            // pyrDown source level (level - interval), -1 for resized levels
    int tab_x, tab_y;         // offsets into the resize tables (levels < interval)
    long long img_off;        // pixel offset of the level image (multiply by channels for bytes)
    long long blk_off;        // block offset
    long long cell_off;       // cell offset
    long long quad_off;       // offset in units of 4 consecutive cells of a level (combine step)
};

// resize tables (cv::resize INTER_LINEAR 8U fixed point)
struct ResizeTabX { int sx; short a0, a1; };
struct ResizeTabY { int y0, y1; short b0, b1; };
// the same mapping with float coefficients (16U / 32F / 64F images); last: sx is the last source column (D = S[sx])
struct ResizeTabXf { int sx, last; float a0, a1; };
struct ResizeTabYf { int y0, y1; float b0, b1; };
// image depths HOGFeatures::pyramid accepts (src/HOGFeatures.cpp:136-146), OpenCV's depth codes
enum { kDepth8U = 0, kDepth16U = 2, kDepth32F = 5, kDepth64F = 6 };
inline size_t depth_size(int depth) { return depth == kDepth8U ? 1 : depth == kDepth16U ? 2 : depth == kDepth32F ? 4 : depth == kDepth64F ? 8 : 0; }

// bilinear cell weights of a pixel coordinate (src/HOGFeatures.cpp:252-259), depends on sbin only
template <typename R> struct HogCoordT { int ip; R v0, v1; };
typedef HogCoordT<float> HogCoord;
typedef HogCoordT<double> HogCoordD;

struct ConvTile { int level; int y0, x0; };
// Tile of the exact 5 x 5 convolution: 64 consecutive positions of the sequence "every strip of four rows of every level of
// every frame of the launch, left to right", as up to kConvMaxSeg runs inside one strip each (pbd_kernels_conv.hip)
constexpr int kConvMaxSeg = 3;
struct ConvSeg { int frame, level, strip, x0; };            // frame: index within the launch
struct ConvSegTile { int nseg; int len[kConvMaxSeg]; ConvSeg seg[kConvMaxSeg]; };
// tile of HOG blocks handled by one workgroup of the fused gradient + histogram kernel: kHogTBX x hog_tile_rows(sbin)
constexpr int kHogTBX = 16;
inline int hog_tile_rows(int sbin) { return sbin <= 4 ? 16 : 8; }

// distance-transform job = (part, mixture) of one tree-depth group.  Its input is ONE plane: the raw
// response of its filter for a leaf part, or the accumulated score (response + children's messages,
// written by the combine step of the deeper group) otherwise.
struct DtJob {
    int plane;                // filter id (from_acc == 0) or global mixture index (from_acc == 1)
    int from_acc;
    int gm;                   // global (part, mixture) index: the job's plane in the persistent pointer buffers IxRaw / IyRaw
    int osx, osy;             // anchor
    double ax, bx, ay, by;    // Quadratic(-w0,-w1), Quadratic(-w2,-w3)  (src/DynamicProgram.cpp:125-127)
};

// mixtures per part the job tables and the register arrays of the combine / root kernels are sized for (the reference has no
// limit, include/Parts.hpp:51-261; Person_26parts has 6, Face_68parts 1)
constexpr int kMaxMix = 16;

// one child part of a combine job
struct ChildDesc {
    int job_begin;            // first DtJob (index within the group) of this child
    int nmix;                 // child mixtures K
    int slot;                 // back-pointer slot of (child, parent mixture 0) = ptr_slot[child]
    int bias_off[kMaxMix];          // biasid[child][mm], mm < K (add the parent mixture)
};

// combine job = one PARENT part: for every parent mixture m
//   acc[m] = response(parent, m); for each child in DESCENDING index order: acc[m] += max_mm(dt[child][mm] + bias)
// (the order of the reference's in-place `parent.score += maxv`, src/DynamicProgram.cpp:95,154-156)
struct CombineJob {
    int child_begin, child_end;   // range in the ChildDesc list, descending child index
    int npar;                     // parent mixtures L
    int acc_plane;                // global mixture index of (parent, mixture 0)
    int filter[kMaxMix];                // response plane of (parent, m)
};

struct RootJob {              // one per component
    int nmix;
    int plane[kMaxMix];             // response plane (filter id) or accumulated-score plane per root mixture
    int from_acc;             // bit mm set: plane[mm] is an accumulated-score plane
    float bias;
};

// Sequential schedule (a filter id used more than once inside a component): the reference keys the accumulated
// scores by FILTER id (src/DynamicProgram.cpp:93,115-119,154-156), so parts sharing a filter see each other's
// contributions in processing order (parts nparts-1 .. 1).  One step = one part per component, and this job is the
// part's contribution to its parent's accumulators, applied in place in parent-mixture order.
struct SeqCombineJob {
    int job_begin;            // first DtJob of the part within the step
    int nmix;                 // the part's mixtures K
    int slot;                 // back-pointer slot of (part, parent mixture 0)
    int bias_off[kMaxMix];          // biasid[part][mm]
    int npar;                 // parent mixtures L
    int target[kMaxMix];            // accumulated-score plane (component * F + filter id of (parent, pm))
    int filter[kMaxMix];            // filter id of (parent, pm): the plane the accumulator starts from
    int init[kMaxMix];              // 1: the accumulator has not been touched yet -> start from the raw response (:155)
};

struct PartWalk {             // argmin tree walk, one per part of a component
    int parent;               // local parent index
    int slot;                 // ptr_slot
    int mix0;                 // global (part, mixture) index of the part's mixture 0
    int ksize[kMaxMix];             // filter size per mixture (xsize == ysize == rows, include/Parts.hpp:185-187)
};

constexpr int kWalkMaxParts = 160;   // parts per component the back-tracking walk holds in LDS (Face_68parts: 68, Person_26parts: 26)
// bytes of one spilled PAIR of envelope-stack entries {T sa, sb, za; unsigned vv;} (natural alignment of T)
constexpr size_t kStkPairF32 = 16, kStkPairF64 = 32;
constexpr int kConvTW = 32, kConvTH = 8, kConvQ = 8;
constexpr int kConvMaxK = 31;        // largest filter side of the generic convolution kernel (the reference has no limit)
#ifndef PBD_CONV3_NW
#define PBD_CONV3_NW 8
#endif
constexpr int kConv3NW = PBD_CONV3_NW;   // waves per workgroup of k_conv3

// ---- launch parameter blocks ---------------------------------------------------------------
struct PyrParams {
    const LevelDesc *lv;
    int nlevels, interval, cn;
    int frame0;                   // first frame of this launch (grid index 0)
    long long pix_per_frame;      // pixels (not bytes) of all level images of one frame
    uint8_t *pyr;                 // [frames][pix_per_frame*cn] elements of the image depth
    const uint8_t *frames;        // [frames][rows*cols*cn] dense
    int rows, cols;
    const ResizeTabX *tabx;
    const ResizeTabY *taby;
    int depth;                    // kDepth8U (fixed-point resampling, the tables above) or 16U / 32F / 64F (tables below)
    const ResizeTabXf *tabxf;
    const ResizeTabYf *tabyf;
};

struct HogParams {
    const LevelDesc *lv;
    int nlevels, cn, sbin;
    int frame0;
    long long pix_per_frame, blk_per_frame, cell_per_frame;
    const uint8_t *pyr;
    int depth;                    // image depth of `pyr`
    const void *coord;            // HogCoordT<R>[]
    void *gmag;                   // R [frames][pix_per_frame] gradient magnitude per image pixel
    uint8_t *gori;                // [frames][pix_per_frame] snapped orientation 0..17
    void *hist;                   // R [frames][18][blk_per_frame]
    void *norm;                   // R [frames][blk_per_frame]
    void *feat;                   // R [frames][cell_per_frame*32]
    const ConvTile *htiles;       // block tiles of the fused gradient + histogram kernel ({level, by0, bx0})
    int nhtiles;
};

struct ConvParams {
    const LevelDesc *lv;
    const ConvTile *tiles;        // uniform 32 x 8 tiling (generic and MFMA kernels)
    int ntiles;
    const ConvTile *shaped;       // mixed-shape tiling of the exact 5x5 kernel: 32x8 tiles, then 16x16, then 8x32
    int nshaped[3];
    const ConvSegTile *segtiles;  // the exact / FMA 5x5 kernel's cover of the whole launch (all frames)
    int nsegtiles;
    int F;                        // response planes per cell block (all filters of the bank)
    int nf, Fpad, ksize;          // this launch: filters of one size class, padded to kConvQ, their size
    const int *fmap;              // class-local filter index -> response plane (NULL: identity, the single-class case)
    int groups_per_block;         // filter groups (of kConvQ) handled by one workgroup
    int cblock;                   // generic kernel: channels of the haloed tile staged in LDS at a time (32, or less for large filters)
    // k_conv3: the bank cut into units of 2 / 4 / 6 / 8 filters, weights per unit [32][25][QL]
    const void *wts3;
    const int *unit_f0, *unit_ql; // first filter (class-local index) and filters of a unit
    const int *unit_woff;         // float offset of a unit's weights in wts3
    int nunits, units_per_block;
    int c31_zero;                 // the features come from this library's HOG: channel 31 is 0 in every cell of the image
    const float *c31tab;          // k_conv3: [81 border cases][c31stride] ordered sums of the out-of-image taps' channel-31 weights
    int c31stride;
    int frame0;
    long long cell_per_frame;
    const void *feat;             // R [frames][cell_per_frame*32]
    const void *wts;              // R; 5x5 float kernel: [group][32][tap][8]; generic: [32][k*k][Fpad]
    void *resp;                   // R [frames][cell_per_frame*F], level-major then filter planes (fp16 in PBD_CONV_MFMA_F16 mode)
    int fma;
};

struct DpParams {
    const LevelDesc *lv;
    int nlevels;
    int F, NS, NC, NM;            // filters, pointer slots, components, (part, mixture) pairs
    long long cell_per_frame;
    int frame0;                   // first frame of this chunk (absolute index into resp/msg/ptr buffers)
    const void *resp;             // R, or fp16 when resp_half (PBD_CONV_MFMA_F16: BASELINE configs[4] "fp16 responses")
    int bz_x, bz_y;               // every job of the launch has a linear coefficient of exactly -0.0 (and a != 0) along x / y
    int resp_half;
    void *acc;                    // R [frames][cell_per_frame*NM] accumulated scores of non-leaf parts
    uint8_t *Ik;                  // [frames][cell_per_frame*NS] winning child mixture per (part, parent mixture) slot
    int NJ;                       // planes per cell block of IxRaw / IyRaw (= (part, mixture) pairs of the model)
    int ptr8;
    // group scratch, indexed by chunk-local frame
    int JG;                       // jobs in this group
    void *tmp, *dt;               // R [chunk][cell_per_frame*JG]
    long long quad_per_frame;
    int max_mix;                  // largest number of mixtures of any part of the model (<= kMaxMix)
    int lane_shift;               // distance-transform passes: 64 >> lane_shift rows (columns) per wave (set per launch)
    // the transform's own pointers, row-major, KEPT for the whole batch ([frames][cell_per_frame*NJ], plane = DtJob::gm; uint8
    // when ptr8, else int16): IxRaw from the rows pass (stored TRANSPOSED, [x][y], as that pass writes it), IyRaw[y][x] from the columns pass.  The reference's Ix / Iy of a
    // (part, parent mixture) slot are Ix = IxRaw[k][y][x], Iy = IyRaw[k][y][Ix] with k = Ik (include/DistanceTransform.hpp:233-244,
    // src/DynamicProgram.cpp:146-152); only the candidates' walks and pbd_dp_min's read-back ever need them, so they are
    // composed there instead of for every cell
    void *IxRaw, *IyRaw;
    void *stk;                    // [chunk][JG][stk_per_jf] records of two entries, wave-private, lane-interleaved
    long long stk_per_jf;         // records per (job, frame)
    const long long *stk_row_off; // per rows-pass wave (64 flat rows): first entry
    const long long *stk_col_off; // per columns-pass wave
    const DtJob *jobs;
    const ChildDesc *childs;
    const CombineJob *cjobs;
    const SeqCombineJob *sjobs;   // sequential schedule only
    const float *biasw;
    const int *row2level; const int *rowoff;   // flat row -> level, level -> first flat row
    const int *col2level; const int *coloff;
    int nrows_flat, ncols_flat;
    int longest;                  // longest row / column of the plan (LDS of the cooperative passes)
    void *rootv; int *rooti;      // R / int [frames][cell_per_frame*NC]
    const RootJob *rjobs;
};

struct ArgminParams {
    const LevelDesc *lv;
    int nlevels, NS, NC, nframes;
    long long cell_per_frame;
    const void *rootv; const int *rooti;   // rootv: R
    const void *IxRaw, *IyRaw; const uint8_t *Ik;     // see DpParams
    int NJ;
    int ptr8;
    float thresh;
    const float *scales;          // [nlevels]
    const PartWalk *walk;         // all components concatenated
    const int *walk_off;          // [NC+1]
    int max_parts, stride, capacity;
    // the candidate list as it leaves the device ("payload"): word 0 = number of roots FOUND (may exceed `capacity`), then
    // min(found, capacity) records of `stride` words in (frame, level, component, y, x) order
    int32_t *payload;
    int *blk; int nblk;           // hits per block of the find kernels, then their exclusive prefix sums
    long long ntotal;             // nframes * cell_per_frame * NC root cells
    int frame_offset;             // added to the `frame` field of every record (frames sharded over GPUs: global frame id)
};

// ---- kernel launches and their timing -------------------------------------------------------
// Every kernel of the library is launched through PBD_LAUNCH.  While a profiling scope is open on the calling thread
// (pbd_profile_enable; bench.py's roofline figures) the launch carries a start / stop event pair of its own
// (hipExtLaunchKernelGGL): the timestamps are those of the kernel's dispatch packet, so nothing is inserted into the stream
// between kernels.  (Bracketing a kernel with hipEventRecord puts a marker packet on either side of it: 10.5 us per
// kernel boundary in the rocprofv3 trace of round 3, 0.45 ms of every 64-frame step -- profiles/r03_hd/README.md.)
struct ProfHook {
    void *ctx;
    void (*take)(void *ctx, hipEvent_t *start, hipEvent_t *stop);
};
extern thread_local ProfHook *g_prof_hook;

template <typename F, typename... Args>
inline void launch_k(F kernel, const dim3 &grid, const dim3 &block, unsigned lds, hipStream_t s, Args... args)
{
    hipEvent_t a = nullptr, b = nullptr;
    if (g_prof_hook) g_prof_hook->take(g_prof_hook->ctx, &a, &b);
    hipExtLaunchKernelGGL(kernel, grid, block, lds, s, a, b, 0, args...);
}
#define PBD_LAUNCH(kernel, grid, block, lds, stream, ...) ::pbd::launch_k(kernel, grid, block, lds, stream, __VA_ARGS__)

// ---- launchers (pbd_kernels_*.hip) ---------------------------------------------------------
void launch_resize(const PyrParams &p, int nframes, long long npix_resized, hipStream_t s);
void launch_pyrdown_range(const PyrParams &p, int nframes, int first_level, int last_level, long long base,
                          long long npix, hipStream_t s);
// `f64` selects the reference's T=double instantiation (every real-typed buffer then holds doubles)
void launch_hog_hist(const HogParams &p, int nframes, bool f64, hipStream_t s);
void launch_hog_feat(const HogParams &p, int nframes, bool f64, hipStream_t s);
void launch_conv(const ConvParams &p, int nframes, bool f64, hipStream_t s);
int conv_occupancy(int nw);
int conv_mfma_occupancy(bool f16);
// matrix-core path (pbd_kernels_conv_mfma.hip); wrec: [pass][tap][160 filters][144 B] bf16 hi/lo records
// PBD_CONV_MFMA_F16: 80 B fp16 records, one MFMA per product tile
void launch_conv_mfma(const ConvParams &p, const void *wrec, bool f16, int nframes, hipStream_t s);
constexpr int kMfmaFilterBlock = 160, kMfmaRecBytes = 144, kMfmaRecBytesF16 = 80;
void launch_dt_rows(const DpParams &p, int nframes, bool f64, hipStream_t s);
void launch_dt_cols(const DpParams &p, int nframes, bool f64, hipStream_t s);
void launch_dp_combine(const DpParams &p, int ncjobs, int nframes, bool f64, hipStream_t s);
void launch_dp_combine_seq(const DpParams &p, int nsjobs, int nframes, bool f64, hipStream_t s);
void launch_dp_root(const DpParams &p, int nframes, bool f64, hipStream_t s);
void launch_argmin_find(const ArgminParams &p, bool f64, hipStream_t s);
int argmin_find_span();       // root cells per block of the find kernels (sizes ArgminParams::blk)
void launch_argmin_walk(const ArgminParams &p, bool f64, hipStream_t s);

}  // namespace pbd
