/*
 * pbd_oracle.h -- CPU restatement ("oracle") of the PartsBasedDetector detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (partsbaseddetector_amd/, include/, the C-ABI
 * library) may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or fixtures for this
 * path, and it cannot be compiled in this image (it needs OpenCV and Boost,
 * which are absent; writing stand-in headers for them is not permitted).  This restatement follows
 * the reference sources line by line (citations below) and is cross-checked in tests/ by
 * independent formulations (brute-force max-plus distance transform, scipy correlation,
 * scatter-form HOG), but no output of the reference itself pins it.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 * Suffix _f32 = reference template parameter T=float (src/demo.cpp:85), _f64 = T=double
 * (cells/detect.cpp:93, ros/Node.hpp:121).
 */
#ifndef PBD_ORACLE_H_
#define PBD_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBDO_MAX_LEVELS 128

/* Flattened model: include/Model.hpp:49-122 + include/Parts.hpp:51-261 as plain arrays.
 * gp = part_offset[c] + p is the global part index; gm = mix_offset[gp] + m the global
 * (part, mixture) index. */
typedef struct pbdo_model {
    int ncomponents;
    int nfilters;
    int flen;                 /* channels per cell (32) */
    const int *filter_ksize;  /* [nfilters] filter is ksize x ksize x flen, channel fastest (Mat(k, k*flen)) */
    const int64_t *filter_offset; /* [nfilters] offset (in elements) into filters_f32 / filters_f64 */
    const float *filters_f32;
    const double *filters_f64;
    int nbias;
    const float *biasw;       /* [nbias]   (vectorf in the reference for both T) */
    int ndefs;
    const float *defw;        /* [ndefs*4] (vector2Df) */
    const int *anchors;       /* [ndefs*2] x,y (cv::Point, 0-based) */
    const int *part_offset;   /* [ncomponents+1] */
    const int *parentid;      /* [totparts], root = -1 */
    const int *mix_offset;    /* [totparts+1] */
    const int *filterid;      /* [totmix] */
    const int *biasid;        /* [totmix] biasid_[c][p][mm]; root: only entry 0 is meaningful */
    const int *defid;         /* [totmix] defid_[c][p][mm]; root: unused */
    float thresh;
    int sbin;
    int interval;             /* Model::nscales_ is really the interval (src/FileStorageModel.cpp:105) */
    int norient;
} pbdo_model;

/* One detection; mirrors include/Candidate.hpp:56-80 (parts_ rects, confidence_[0], component_)
 * plus the root location/level it was back-tracked from (used to give candidates a total order,
 * the reference's own order being nondeterministic: src/DynamicProgram.cpp:246-251). */
typedef struct pbdo_candidate_hdr {
    int component;
    int level;
    int root_x;
    int root_y;
    float score;
    int nparts;
} pbdo_candidate_hdr;

/* ---- pyramid geometry: src/HOGFeatures.cpp:95-127, include/HOGFeatures.hpp:74-81 ---- */
int pbdo_pyramid_plan(int rows, int cols, int sbin, int interval,
                      int *lvl_rows, int *lvl_cols, float *scales); /* returns nscales (<= PBDO_MAX_LEVELS) */

/* ---- resampling (arithmetic lives in OpenCV, restated from SURVEY.md Appendix E; third-party, unpinned) ---- */
void pbdo_resize_linear_u8(const uint8_t *src, int srows, int scols, int cn, size_t sstride,
                           uint8_t *dst, int drows, int dcols, size_t dstride);
void pbdo_pyrdown_u8(const uint8_t *src, int srows, int scols, int cn, size_t sstride,
                     uint8_t *dst, size_t dstride); /* dst is ((srows+1)/2) x ((scols+1)/2) */
/* all pyramid images, densely packed one after another into `out` (level l at out + img_offset[l],
 * row stride lvl_cols[l]*cn); returns nscales */
int pbdo_pyramid_images_u8(const uint8_t *im, int rows, int cols, int cn, size_t stride,
                           int sbin, int interval, uint8_t *out, int64_t *img_offset,
                           int *lvl_rows, int *lvl_cols, float *scales);

/* image depths HOGFeatures::pyramid accepts (src/HOGFeatures.cpp:136-146), OpenCV's depth codes */
enum { PBDO_8U = 0, PBDO_16U = 2, PBDO_32F = 5, PBDO_64F = 6 };
size_t pbdo_depth_size(int depth);
/* depth-generic forms: strides, `out` and img_offset in ELEMENTS of the depth (third-party resampling, unpinned) */
void pbdo_resize_linear(const void *src, int depth, int srows, int scols, int cn, size_t sstride, void *dst, int drows,
                        int dcols, size_t dstride);
void pbdo_pyrdown(const void *src, int depth, int srows, int scols, int cn, size_t sstride, void *dst, size_t dstride);
int pbdo_pyramid_images(const void *im, int depth, int rows, int cols, int cn, size_t stride, int sbin, int interval,
                        void *out, int64_t *img_offset, int *lvl_rows, int *lvl_cols, float *scales);
void pbdo_hog_features_d_f32(const void *im, int depth, int rows, int cols, int cn, size_t stride, int sbin, int norient,
                             int flen, float *feat);
void pbdo_hog_features_d_f64(const void *im, int depth, int rows, int cols, int cn, size_t stride, int sbin, int norient,
                             int flen, double *feat);
int pbdo_features_d_f32(const pbdo_model *m, const void *im, int depth, int rows, int cols, int cn, size_t stride,
                        float *feat, int64_t *feat_offset, int *out_rows, int *out_cols, float *scales);
int pbdo_features_d_f64(const pbdo_model *m, const void *im, int depth, int rows, int cols, int cn, size_t stride,
                        double *feat, int64_t *feat_offset, int *out_rows, int *out_cols, float *scales);
int pbdo_detect_d_f32(const pbdo_model *m, const void *im, int depth, int rows, int cols, int cn, size_t stride,
                      pbdo_candidate_hdr *hdr, int *rects, int max_parts, int capacity, double *stage_ms);
int pbdo_detect_d_f64(const pbdo_model *m, const void *im, int depth, int rows, int cols, int cn, size_t stride,
                      pbdo_candidate_hdr *hdr, int *rects, int max_parts, int capacity, double *stage_ms);


/* ---- HOG: src/HOGFeatures.cpp:168-341 ---- */
void pbdo_hog_dims(int rows, int cols, int sbin, int *out_rows, int *out_cols);
void pbdo_hog_features_f32(const uint8_t *im, int rows, int cols, int cn, size_t stride,
                           int sbin, int norient, int flen, float *feat);
void pbdo_hog_features_f64(const uint8_t *im, int rows, int cols, int cn, size_t stride,
                           int sbin, int norient, int flen, double *feat);

/* ---- convolution: src/SpatialConvolutionEngine.cpp:70-94,133-159; src/filter.cpp:3808-3924 ---- */
void pbdo_conv_f32(const float *feat, int H, int W, int flen, const float *filt, int k, float *resp);
void pbdo_conv_f64(const double *feat, int H, int W, int flen, const double *filt, int k, double *resp);

/* ---- distance transform: include/DistanceTransform.hpp:89-105,152-245 ---- */
void pbdo_dt_f32(const float *in, int M, int N, double ax, double bx, double ay, double by,
                 int osx, int osy, float *out, int *Ix, int *Iy);
void pbdo_dt_f64(const double *in, int M, int N, double ax, double bx, double ay, double by,
                 int osx, int osy, double *out, int *Ix, int *Iy);

/* ---- dynamic program for one (level, component): src/DynamicProgram.cpp:67-173 ----
 * responses: nfilters maps of H x W (resp[f] = responses + f*H*W).
 * Ix/Iy/Ik: for every non-root part gp of the component and parent mixture m a H x W int map at
 *           ((mix_offset_par(gp,m)) * H*W) where the slot index is ptr_slot[gp] + m (see pbdo_ptr_slots).
 * rootv/rooti: H x W. */
int  pbdo_ptr_slots(const pbdo_model *m, int *ptr_slot /*[totparts]*/); /* returns total slots */
void pbdo_dp_min_f32(const pbdo_model *m, int c, const float *responses, int H, int W,
                     int *Ix, int *Iy, int *Ik, float *rootv, int *rooti);
void pbdo_dp_min_f64(const pbdo_model *m, int c, const double *responses, int H, int W,
                     int *Ix, int *Iy, int *Ik, double *rootv, int *rooti);

/* ---- back-tracking for one (level, component): src/DynamicProgram.cpp:190-255 ----
 * Appends candidates (raster order of root hits) to hdr/rects (rects: nparts x {x,y,w,h} per
 * candidate, stride max_parts*4). Returns the number appended, or -1 on capacity overflow. */
int pbdo_dp_argmin_f32(const pbdo_model *m, int c, int level, float scale, int H, int W,
                       const int *Ix, const int *Iy, const int *Ik,
                       const float *rootv, const int *rooti,
                       pbdo_candidate_hdr *hdr, int *rects, int max_parts, int capacity);
int pbdo_dp_argmin_f64(const pbdo_model *m, int c, int level, float scale, int H, int W,
                       const int *Ix, const int *Iy, const int *Ik,
                       const double *rootv, const int *rooti,
                       pbdo_candidate_hdr *hdr, int *rects, int max_parts, int capacity);

/* ---- whole path: src/PartsBasedDetector.cpp:69-95 ----
 * im: rows x cols x cn uint8 (BGR interleaved when cn==3). Candidates sorted by (level, component,
 * root_y, root_x). Returns candidate count or -1 on overflow. stage_ms (may be NULL) receives wall
 * milliseconds for {pyramid images, HOG, conv, dp min, argmin}. */
int pbdo_detect_f32(const pbdo_model *m, const uint8_t *im, int rows, int cols, int cn, size_t stride,
                    pbdo_candidate_hdr *hdr, int *rects, int max_parts, int capacity, double *stage_ms);
int pbdo_detect_f64(const pbdo_model *m, const uint8_t *im, int rows, int cols, int cn, size_t stride,
                    pbdo_candidate_hdr *hdr, int *rects, int max_parts, int capacity, double *stage_ms);

/* staged access for tests: features / responses of every level, packed */
int pbdo_features_f32(const pbdo_model *m, const uint8_t *im, int rows, int cols, int cn, size_t stride,
                      float *feat, int64_t *feat_offset, int *out_rows, int *out_cols, float *scales);
void pbdo_responses_f32(const pbdo_model *m, const float *feat, int H, int W, float *resp /*nfilters*H*W*/);

int pbdo_num_threads(void);
void pbdo_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif /* PBD_ORACLE_H_ */
