/*
 * pbd.h -- C ABI of the MI355X-native PartsBasedDetector detection hot path.
 *
 * Drop-in boundary: each entry point states the reference interface it replaces (paths relative to
 * the reference repository).  Plain C, no OpenCV/STL/torch types; every pointer is a host pointer
 * unless the name says `_device`.  Functions return PBD_OK (0) or a negative pbd_status; the text of
 * the last failure on a handle is available from pbd_last_error().  No exception crosses this ABI.
 *
 * Layouts at the seam are the reference's:
 *   image    rows x cols x channels, uint8, interleaved BGR (channels 3) or grey (channels 1)
 * T below is the handle's real type: float for PBD_REAL_F32 (src/demo.cpp:85), double for PBD_REAL_F64
 * (cells/detect.cpp:93, ros/Node.hpp:121); real-typed buffers cross the ABI as void pointers to T.
 *   feature  Mat(H, W*flen) single-channel T, channel fastest        (src/HOGFeatures.cpp:180,288)
 *   filter   Mat(k, k*flen) T, same interleave                       (src/MatlabIOModel.cpp:115-123)
 *   response Mat(H, W) T, one per (level, filter): responses[level][filter]
 *
 * One handle = one host thread = one GPU (as the reference's detector is not re-entrant:
 * src/HOGFeatures.cpp:99-107).  The library fails if the HIP runtime or device is unavailable; there
 * is no CPU fallback.
 */
#ifndef PBD_H_
#define PBD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBD_MAX_LEVELS 128

typedef enum pbd_status {
    PBD_OK = 0,
    PBD_ERR_INVALID = -1,     /* bad argument / model fails validation (reference: assert / CV_Error) */
    PBD_ERR_UNSUPPORTED = -2, /* e.g. image depth other than 8-bit (src/HOGFeatures.cpp:136-146 default branch) */
    PBD_ERR_HIP = -3,         /* HIP runtime error */
    PBD_ERR_CAPACITY = -4,    /* candidate capacity exceeded; output truncated */
    PBD_ERR_STATE = -5,       /* call order violated (e.g. pdf before setFilters) */
    PBD_ERR_NOMEM = -6
} pbd_status;

/* Flattened Model + Parts tables (include/Model.hpp:49-122, include/Parts.hpp:51-261).
 * gp = part_offset[c] + p is a global part index; gm = mix_offset[gp] + m a global (part, mixture)
 * index.  Copied at pbd_create (the reference's Parts copies the pools: include/Parts.hpp:232-235). */
typedef struct pbd_model {
    int ncomponents;
    int nfilters;
    int flen;                      /* channels per HOG cell (32) */
    const int *filter_ksize;       /* [nfilters]; filter f is k x k x flen */
    const int64_t *filter_offset;  /* [nfilters] element offset of filter f in filters_f32/_f64 */
    const float *filters_f32;      /* used when real_type == PBD_REAL_F32 */
    const double *filters_f64;     /* used when real_type == PBD_REAL_F64 (may be NULL otherwise) */
    int nbias;
    const float *biasw;            /* [nbias] */
    int ndefs;
    const float *defw;             /* [ndefs*4] */
    const int *anchors;            /* [ndefs*2] (x, y), 0-based */
    const int *part_offset;        /* [ncomponents+1] */
    const int *parentid;           /* [totparts], root -1; parent index < child index */
    const int *mix_offset;         /* [totparts+1] */
    const int *filterid;           /* [totmix] */
    const int *biasid;             /* [totmix]: biasid_[c][p][mm]; bias(mm)[m] = biasw[biasid + m] (Parts.hpp:172-175) */
    const int *defid;              /* [totmix]: defid_[c][p][mm] (root: ignored) */
    float thresh;
    int sbin;
    int interval;                  /* Model::nscales_ (src/FileStorageModel.cpp:105) */
    int norient;                   /* 18 */
} pbd_model;

enum { PBD_REAL_F32 = 0, PBD_REAL_F64 = 1 };
enum { PBD_CONV_EXACT = 0,   /* multiply and add rounded separately in the reference's order: bit-identical responses */
       PBD_CONV_FMA = 1,     /* fused multiply-add: responses within 1e-4, not bit-identical */
       PBD_CONV_MFMA = 2,    /* matrix cores, bf16 hi/lo operand split (3 MFMAs per product tile), fp32 accumulation:
                                responses within 1e-4 (~1e-6 observed), not bit-identical; 5x5 filters, PBD_REAL_F32 only */
       PBD_CONV_MFMA_F16 = 3 }; /* matrix cores, operands rounded once to fp16 (1 MFMA per product tile), fp32 accumulation:
                                the 1e-4 bar does not hold (~1e-3 observed on unit-scale scores); same restrictions.
                                In this mode the RESPONSES live on the device as fp16 (BASELINE configs[4] "fp16 responses"):
                                pbd_conv_pdf returns them widened to float, |v| >= 65520 saturates to +-inf, and pbd_dp_min
                                ROUNDS THE CALLER'S float scores to fp16 before the dynamic program (exact for scores that
                                came from pbd_conv_pdf; arbitrary scores lose precision: tests/test_gpu_parity.py::
                                test_mfma_f16_dp_min_rounds_its_input pins this) */

typedef struct pbd_config {
    int device;            /* HIP device ordinal */
    int real_type;         /* PBD_REAL_F32 (src/demo.cpp:85) or PBD_REAL_F64 (cells/detect.cpp:93) */
    int conv_mode;         /* PBD_CONV_EXACT / PBD_CONV_FMA / PBD_CONV_MFMA / PBD_CONV_MFMA_F16 */
    int max_batch;         /* frames per pbd_detect_batch* call (>= 1) */
    int max_candidates;    /* candidate capacity per batch */
    void *stream;          /* hipStream_t to run on (NULL: the library creates its own) */
} pbd_config;

/* One detection = one reference Candidate (include/Candidate.hpp:56-80): parts_ as x,y,w,h,
 * confidence_[0] = score (other confidences are 0: src/DynamicProgram.cpp:241-244), component_.
 * A record is pbd_candidate_stride() int32 words: this header followed by max_parts x {x,y,w,h}. */
typedef struct pbd_candidate_hdr {
    int32_t frame;      /* index within the batch */
    int32_t component;
    int32_t level;
    int32_t root_x;
    int32_t root_y;
    float score;
    int32_t nparts;
    int32_t reserved;
} pbd_candidate_hdr;

typedef struct pbd_handle pbd_handle;

/* ---- lifetime.  Replaces PartsBasedDetector<T>::distributeModel (src/PartsBasedDetector.cpp:102-127). */
int pbd_create(const pbd_model *model, const pbd_config *config, pbd_handle **out);
void pbd_destroy(pbd_handle *h);
const char *pbd_last_error(const pbd_handle *h); /* h may be NULL: last error of a failed pbd_create */
const char *pbd_version(void);
int pbd_candidate_stride(const pbd_handle *h);  /* int32 words per candidate record */

/* ---- IFeatures (include/IFeatures.hpp:49-73), implemented by HOGFeatures<T> (src/HOGFeatures.cpp). */
int pbd_binsize(const pbd_handle *h);                      /* IFeatures::binsize */
/* Plans the pyramid for a rows x cols frame: level image sizes, feature map sizes and scales
 * (src/HOGFeatures.cpp:95-127,174-175).  Replaces the size/scale logic of HOGFeatures::pyramid and
 * IFeatures::nscales()/scales().  Arrays hold PBD_MAX_LEVELS entries. */
int pbd_pyramid_plan(pbd_handle *h, int rows, int cols, int *nlevels, int *img_rows, int *img_cols,
                     int *feat_rows, int *feat_cols, float *scales);
/* IFeatures::pyramid(im, pyrafeatures): feat[l] receives feat_rows[l] x (feat_cols[l]*flen) values of T.
 * stride_bytes: byte distance between image rows (cv::Mat::step).  depth_code = cv::Mat::depth() of the image: 0 (CV_8U),
 * 2 (CV_16U), 5 (CV_32F) or 6 (CV_64F) -- the four features<IT> instantiations of src/HOGFeatures.cpp:136-146; any other
 * depth fails with PBD_ERR_UNSUPPORTED as the reference's CV_Error.  A 32F / 64F image holding a NaN or Inf pixel is
 * refused with PBD_ERR_INVALID (here and in pbd_detect_typed): non-finite input is an error, not a silently different result. */
int pbd_features_pyramid(pbd_handle *h, const void *img, int rows, int cols, int channels,
                         size_t stride_bytes, int depth_code, void *const *feat);
/* the resampled level images of the last pbd_features_pyramid / pbd_detect call (for tests) */
int pbd_get_pyramid_image(pbd_handle *h, int frame, int level, uint8_t *dst);

/* Level sharding (new surface; the reference has no multi-device mode): pyramid levels are independent through HOG,
 * convolution and the dynamic program (src/DynamicProgram.cpp:115-119 reads only scores[n] of the same level), so ONE
 * frame can be split over `world` GPUs -- handle `rank` then computes only its share of the levels (longest-processing-
 * time assignment over the level sizes, the same on every rank) and returns only their candidates; the union over the
 * ranks is the full result.  pbd_pyramid_plan reports 0 x 0 feature maps for the levels of other ranks.  (1, 0..): off. */
int pbd_set_level_shard(pbd_handle *h, int rank, int world);

/* ---- IConvolutionEngine (include/IConvolutionEngine.hpp:44-68), SpatialConvolutionEngine. */
/* setFilters(filters): filters[f] is ksize[f] x (ksize[f]*flen) values of T.  pbd_create already
 * installs the model's filters; this replaces them (src/SpatialConvolutionEngine.cpp:133-159). */
int pbd_conv_set_filters(pbd_handle *h, int nfilters, const void *const *filters, const int *ksize);
/* pdf(features, responses): resp[l] receives nfilters planes of rows[l] x cols[l]
 * (responses[l][f] at resp[l] + f*rows[l]*cols[l]) (src/SpatialConvolutionEngine.cpp:106-124). */
int pbd_conv_pdf(pbd_handle *h, int nlevels, const void *const *feat, const int *rows, const int *cols,
                 void *const *resp);

/* ---- DynamicProgram<T> (include/DynamicProgram.hpp:74-75). */
int pbd_num_ptr_slots(const pbd_handle *h);   /* back-pointer maps per (level): sum over non-root parts of parent mixtures */
int pbd_ptr_slot(const pbd_handle *h, int component, int part); /* slot of (part, parent mixture 0) */
/* min(parts, scores, Ix, Iy, Ik, rootv, rooti) (src/DynamicProgram.cpp:67-173).
 * resp[l]: nfilters planes; Ix/Iy/Ik[l]: pbd_num_ptr_slots planes of int32 (plane slot(part)+m =
 * reference Ix[l][c][part][m]); rootv[l]/rooti[l]: ncomponents planes. */
int pbd_dp_min(pbd_handle *h, int nlevels, const int *rows, const int *cols, const void *const *resp,
               int32_t *const *Ix, int32_t *const *Iy, int32_t *const *Ik, void *const *rootv,
               int32_t *const *rooti);
/* argmin(parts, rootv, rooti, scales, Ix, Iy, Ik, candidates) (src/DynamicProgram.cpp:190-255) on
 * the device-resident result of the last pbd_dp_min / pbd_detect*.  Candidates are written sorted by
 * (frame, level, component, root_y, root_x): the raster order of the reference's Math::find per (level, component)
 * (:208-216); its order ACROSS levels is nondeterministic (#pragma omp critical, :246-251).  The order is produced on the
 * device (ordered compaction, no sort); when more than `capacity` are found, the first `capacity` of that order are
 * returned with PBD_ERR_CAPACITY. */
int pbd_dp_argmin(pbd_handle *h, const float *scales, int32_t *cand, int capacity, int *ncand);

/* ---- PartsBasedDetector<T>::detect (include/PartsBasedDetector.hpp:172-173, src/PartsBasedDetector.cpp:69-95).
 * `depth` of the 3-argument overload is ignored by the reference (:91-93) and has no parameter here. */
int pbd_detect(pbd_handle *h, const void *img, int rows, int cols, int channels, size_t stride_bytes,
               int32_t *cand, int capacity, int *ncand);
/* The same for an image of any accepted depth (depth_code as in pbd_features_pyramid); pbd_detect is depth_code 0.
 * The batch entry points take 8-bit frames. */
int pbd_detect_typed(pbd_handle *h, const void *img, int rows, int cols, int channels, size_t stride_bytes,
                     int depth_code, int32_t *cand, int capacity, int *ncand);
/* New surface (the reference has no batch API): nframes equally-sized frames, results identical to
 * nframes pbd_detect calls, candidate `frame` field = index in the batch. */
int pbd_detect_batch(pbd_handle *h, int nframes, const void *const *imgs, int rows, int cols, int channels,
                     size_t stride_bytes, int32_t *cand, int capacity, int *ncand);
/* Pipelined form of pbd_detect_batch (new surface, as the batch API): submit() copies the frames into a pinned
 * staging buffer of the handle, starts their transfer on a copy stream and enqueues the whole path behind it, without
 * waiting; wait() returns the candidates of the oldest submitted batch (same records and order as pbd_detect_batch).
 * Up to two batches may be in flight -- submit(k+1), then wait(k) -- so that the host-side staging and the PCIe
 * transfer of batch k+1 overlap the kernels of batch k.  The synchronous entry points and the staged read-back
 * refuse to run (PBD_ERR_STATE) while a batch is in flight. */
int pbd_detect_batch_submit(pbd_handle *h, int nframes, const void *const *imgs, int rows, int cols, int channels,
                            size_t stride_bytes);
int pbd_detect_batch_wait(pbd_handle *h, int32_t *cand, int capacity, int *ncand);
/* submit() for frames already resident in device memory (d_frames as in pbd_detect_batch_device; they must stay
 * untouched until the matching wait() returns): nothing is copied in, the candidates' read-back of batch k overlaps the
 * kernels of batch k+1. */
int pbd_detect_batch_device_submit(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int channels);
/* Same, frames already resident in device memory: d_frames = nframes contiguous rows*cols*channels
 * images.  cand is a HOST buffer. */
int pbd_detect_batch_device(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int channels,
                            int32_t *cand, int capacity, int *ncand);

/* Device-resident output (new surface: multi-GPU jobs, device pipelines).  The candidate list stays on the device in the
 * caller's buffer d_payload = int32[1 + capacity * pbd_candidate_stride()] ("payload"):
 *   word 0      number of candidates FOUND -- may exceed `capacity`; then only the first `capacity` of the order are present
 *   word 1...   min(found, capacity) records, sorted by (frame, level, component, root_y, root_x); `frame` = frame_offset +
 *               index in the batch (frames sharded over GPUs: the global frame id)
 * i.e. exactly what a rank contributes to the one gather of Candidate lists per batch (partsbaseddetector_amd/dist.py hands
 * this buffer to all_gather_into_tensor as it is; nothing passes through the host).  ASYNCHRONOUS: the call returns once
 * the work is enqueued on pbd_stream(); order later work behind that stream (or call pbd_synchronize).
 * pbd_argmin_device_out re-emits the list of the batch still resident from the last pbd_detect* call -- used after a
 * capacity overflow (word 0 > capacity) with a larger buffer; the dynamic program is not run again. */
int pbd_detect_batch_device_out(pbd_handle *h, int nframes, const void *d_frames, int rows, int cols, int channels,
                                int frame_offset, int32_t *d_payload, int capacity);
int pbd_argmin_device_out(pbd_handle *h, int frame_offset, int32_t *d_payload, int capacity);
/* the hipStream_t every kernel of this handle runs on (pbd_config.stream, or the library's own) */
void *pbd_stream(const pbd_handle *h);

/* ---- staged read-back of the last pbd_detect* call (tests, profiling) ---- */
enum { PBD_STAGE_FEATURES = 0, PBD_STAGE_RESPONSES = 1, PBD_STAGE_ROOTV = 2, PBD_STAGE_ROOTI = 3 };
int pbd_get_stage(pbd_handle *h, int stage, int frame, int level, void *dst, size_t dst_bytes);

/* ---- per-kernel timing with HIP events on the library's stream (bench.py roofline) ---- */
enum { PBD_K_RESIZE = 0, PBD_K_PYRDOWN, PBD_K_HOG_HIST, PBD_K_HOG_FEAT, PBD_K_CONV, PBD_K_DT_ROWS,
       PBD_K_DT_COLS, PBD_K_DP_COMBINE, PBD_K_DP_ROOT, PBD_K_ARGMIN, PBD_K_COUNT };
/* on = 1: every kernel launch carries a start / stop event pair (the runtime isolates a timed dispatch: about 1 ms per
 * 64-frame step of ~45 launches); on = 2: only the convolution (one launch per step: free); 0: off */
int pbd_profile_enable(pbd_handle *h, int on);
int pbd_profile_reset(pbd_handle *h);
/* total_ms / launches accumulated since the last reset for kernel id k */
int pbd_profile_read(pbd_handle *h, int k, double *total_ms, int *launches);
const char *pbd_kernel_name(int k);
int pbd_synchronize(pbd_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* PBD_H_ */
