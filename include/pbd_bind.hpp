// pbd_bind.hpp -- the bodies of every C++ binding of the C ABI (pbd.h), written ONCE over a small traits type.
//
// Two hosts instantiate these templates:
//   * include/pbd_host.hpp            Tr = pbdhost::HostTraits<T>   (plain buffers, no OpenCV) -- compiled into
//                                     host/pbd_demo and run on the GPU by tests/test_host_demo.py, T = float and double
//   * include/pbd_opencv_adapters.hpp Tr = pbd_adapters::CvTraits<T> (cv::Mat)                  -- what a maintainer of the
//                                     reference adds; it supplies ONLY the cv::Mat traits and class shells
// so the code that crosses the ABI in the OpenCV adapters is the code the tests ran.  No OpenCV, no Boost, C++03-clean.
//
// What a traits type Tr provides (T = the detector's real type, `PartsBasedDetector<T>`: float as src/demo.cpp:85, double as
// cells/detect.cpp:93 / ros/Node.hpp:121):
//   typedef T Real;  typedef ... Mat;  typedef ... IMat (int32 matrix);  typedef ... Image;  typedef ... Candidate;
//   static void        create(Mat &m, int rows, int cols);        rows x cols values of T, continuous
//   static T          *ptr(Mat &m);                                first element
//   static const T    *cptr(const Mat &m);
//   static int         rows(const Mat &m);   static int cols(const Mat &m);
//   static void        icreate(IMat &m, int rows, int cols);   static int32_t *iptr(IMat &m);
//   static Mat         real_continuous(const Mat &m);              m as continuous T data (the same matrix if it already is)
//   static Mat         rows_view(Mat &m, int r0, int r1);          rows [r0, r1) of m (a view where the host has views)
//   static const void *img_data(const Image &);  img_rows / img_cols / img_channels -> int;  img_step -> size_t (bytes
//                      between rows, cv::Mat::step);  img_depth -> int (cv::Mat::depth(): 0 8U, 2 16U, 5 32F, 6 64F)
//   static void        fail(int status, const std::string &text);  throws the host's exception type; never returns
//   static void        candidate(std::vector<Candidate> &out, const pbd_candidate_hdr &hd, const int32_t *rects);
//                      appends one Candidate: hd.nparts boxes {x, y, w, h}; confidence hd.score for part 0, 0 for the others
//                      (src/DynamicProgram.cpp:241-244)
// and, for FlatModel, over the host's model class (reference: include/Model.hpp:49-122 -- the accessor names are its):
//   typedef ... FilterMat;   static int filter_rows(const FilterMat &);
//   static void        filter_values(const FilterMat &w, std::vector<double> &out);   appends rows x (rows*flen) values
#ifndef PBD_BIND_HPP_
#define PBD_BIND_HPP_

#include <stdint.h>

#include <string>
#include <vector>

#include "pbd.h"

namespace pbdbind {

template <typename T> struct RealCode;
template <> struct RealCode<float> { enum { value = PBD_REAL_F32 }; };
template <> struct RealCode<double> { enum { value = PBD_REAL_F64 }; };

template <class Tr>
inline void check(pbd_handle *h, int rc)
{   // the C ABI never throws; the host's own error channel does (reference: assert / CV_Error -> cv::Exception)
    if (rc != PBD_OK) Tr::fail(rc, std::string("pbd: ") + pbd_last_error(h));
}

// Model -> pbd_model: the flattened Parts tables (include/Parts.hpp:103-189) + the filter pool converted to T
// (src/PartsBasedDetector.cpp:114-117).  ModelT is the reference's Model or pbdhost::Model.
template <class Tr>
struct FlatModel {
    std::vector<int> ksize, part_offset, parentid, mix_offset, filterid, biasid, defid, anchors;
    std::vector<int64_t> foff;
    std::vector<float> filters32, biasw, defw;
    std::vector<double> filters64;
    pbd_model m;

    template <class ModelT>
    explicit FlatModel(ModelT &model)
    {
        int64_t off = 0;
        for (size_t f = 0; f < model.filters().size(); ++f) {
            const typename Tr::FilterMat &w = model.filters()[f];
            ksize.push_back(Tr::filter_rows(w));
            foff.push_back(off);
            const size_t before = filters64.size();
            Tr::filter_values(w, filters64);
            off += (int64_t)(filters64.size() - before);
        }
        filters32.resize(filters64.size());
        for (size_t i = 0; i < filters64.size(); ++i) filters32[i] = (float)filters64[i];   // convertTo(CV_32F): one rounding
        biasw.assign(model.bias().begin(), model.bias().end());
        for (size_t d = 0; d < model.def().size(); ++d) {
            for (int i = 0; i < 4; ++i) defw.push_back(model.def()[d][i]);
            anchors.push_back(model.anchors()[d].x);
            anchors.push_back(model.anchors()[d].y);
        }
        part_offset.push_back(0);
        mix_offset.push_back(0);
        for (size_t c = 0; c < model.filterid().size(); ++c) {
            for (size_t p = 0; p < model.filterid()[c].size(); ++p) {
                parentid.push_back(model.parentid()[c][p]);
                const std::vector<int> &fid = model.filterid()[c][p], &bid = model.biasid()[c][p], &did = model.defid()[c][p];
                for (size_t mm = 0; mm < fid.size(); ++mm) {
                    filterid.push_back(fid[mm]);
                    biasid.push_back(mm < bid.size() ? bid[mm] : -1);
                    defid.push_back(p > 0 && mm < did.size() ? did[mm] : -1);
                }
                mix_offset.push_back(mix_offset.back() + (int)fid.size());
            }
            part_offset.push_back(part_offset.back() + (int)model.filterid()[c].size());
        }
        m.ncomponents = (int)model.filterid().size();
        m.nfilters = (int)ksize.size();
        m.flen = model.flen();
        m.filter_ksize = ksize.empty() ? NULL : &ksize[0];
        m.filter_offset = foff.empty() ? NULL : &foff[0];
        m.filters_f32 = filters32.empty() ? NULL : &filters32[0];
        m.filters_f64 = filters64.empty() ? NULL : &filters64[0];
        m.nbias = (int)biasw.size();
        m.biasw = biasw.empty() ? NULL : &biasw[0];
        m.ndefs = (int)(defw.size() / 4);
        m.defw = defw.empty() ? NULL : &defw[0];
        m.anchors = anchors.empty() ? NULL : &anchors[0];
        m.part_offset = &part_offset[0];
        m.parentid = parentid.empty() ? NULL : &parentid[0];
        m.mix_offset = &mix_offset[0];
        m.filterid = filterid.empty() ? NULL : &filterid[0];
        m.biasid = biasid.empty() ? NULL : &biasid[0];
        m.defid = defid.empty() ? NULL : &defid[0];
        m.thresh = model.thresh();
        m.sbin = model.binsize();
        m.interval = model.nscales();      // Model::nscales_ is the interval (src/FileStorageModel.cpp:105)
        m.norient = model.norient();
    }
private:
    FlatModel(const FlatModel &);
    FlatModel &operator=(const FlatModel &);
};

// pbd_create / pbd_destroy: what distributeModel owns (src/PartsBasedDetector.cpp:102-127)
template <class Tr, class ModelT>
inline pbd_handle *create(ModelT &model, int device, int conv_mode, int max_batch, int max_candidates)
{
    FlatModel<Tr> fm(model);
    pbd_config cfg = {device, RealCode<typename Tr::Real>::value, conv_mode, max_batch, max_candidates, NULL};
    pbd_handle *h = NULL;
    const int rc = pbd_create(&fm.m, &cfg, &h);
    if (rc != PBD_OK) Tr::fail(rc, std::string("pbd_create: ") + pbd_last_error(NULL));
    return h;
}

// IFeatures::pyramid(im, pyrafeatures) + scales() (include/IFeatures.hpp:49-73, src/HOGFeatures.cpp:95-127)
template <class Tr>
inline void pyramid(pbd_handle *h, const typename Tr::Image &im, std::vector<typename Tr::Mat> &pyrafeatures,
                    std::vector<float> &scales)
{
    int n = 0, fr[PBD_MAX_LEVELS], fc[PBD_MAX_LEVELS];
    float sc[PBD_MAX_LEVELS];
    check<Tr>(h, pbd_pyramid_plan(h, Tr::img_rows(im), Tr::img_cols(im), &n, NULL, NULL, fr, fc, sc));
    const int flen = 32;                                        // Mat(H, W*flen), src/HOGFeatures.cpp:180
    pyrafeatures.resize(n);
    std::vector<void *> ptrs(n);
    for (int l = 0; l < n; ++l) {
        Tr::create(pyrafeatures[l], fr[l], fc[l] * flen);
        ptrs[l] = Tr::ptr(pyrafeatures[l]);
    }
    // the depth codes of the C ABI are cv::Mat::depth() itself: 8U / 16U / 32F / 64F are the four features<IT>
    // instantiations (src/HOGFeatures.cpp:136-146); any other depth -> PBD_ERR_UNSUPPORTED (the reference's default: branch)
    check<Tr>(h, pbd_features_pyramid(h, Tr::img_data(im), Tr::img_rows(im), Tr::img_cols(im), Tr::img_channels(im),
                                      Tr::img_step(im), Tr::img_depth(im), n ? &ptrs[0] : NULL));
    scales.assign(sc, sc + n);
}

// IConvolutionEngine::setFilters (include/IConvolutionEngine.hpp:67, src/SpatialConvolutionEngine.cpp:133-159)
template <class Tr>
inline void set_filters(pbd_handle *h, const std::vector<typename Tr::Mat> &filters)
{
    const size_t F = filters.size();
    std::vector<typename Tr::Mat> real(F);                      // converted to T / made continuous, alive across the call
    std::vector<const void *> ptrs(F);
    std::vector<int> ks(F);
    for (size_t f = 0; f < F; ++f) {
        real[f] = Tr::real_continuous(filters[f]);
        ptrs[f] = Tr::cptr(real[f]);
        ks[f] = Tr::rows(real[f]);
    }
    check<Tr>(h, pbd_conv_set_filters(h, (int)F, F ? &ptrs[0] : NULL, F ? &ks[0] : NULL));
}

// IConvolutionEngine::pdf (include/IConvolutionEngine.hpp:56, src/SpatialConvolutionEngine.cpp:106-124):
// responses[level][filter] = H x W, here views of one packed matrix per level (nfilters*H rows x W)
template <class Tr>
inline void pdf(pbd_handle *h, size_t nfilters, const std::vector<typename Tr::Mat> &features,
                std::vector<std::vector<typename Tr::Mat> > &responses)
{
    const int M = (int)features.size(), flen = 32;
    std::vector<typename Tr::Mat> cont(M), packed(M);
    std::vector<const void *> fp(M);
    std::vector<void *> rp(M);
    std::vector<int> rows(M), cols(M);
    for (int m = 0; m < M; ++m) {
        cont[m] = Tr::real_continuous(features[m]);
        rows[m] = Tr::rows(cont[m]);
        cols[m] = Tr::cols(cont[m]) / flen;
        fp[m] = Tr::cptr(cont[m]);
        Tr::create(packed[m], (int)nfilters * rows[m], cols[m]);
        rp[m] = Tr::ptr(packed[m]);
    }
    check<Tr>(h, pbd_conv_pdf(h, M, M ? &fp[0] : NULL, M ? &rows[0] : NULL, M ? &cols[0] : NULL, M ? &rp[0] : NULL));
    responses.assign(M, std::vector<typename Tr::Mat>(nfilters));
    for (int m = 0; m < M; ++m)
        for (size_t n = 0; n < nfilters; ++n) responses[m][n] = Tr::rows_view(packed[m], (int)n * rows[m], (int)(n + 1) * rows[m]);
}

// DynamicProgram<T>::min (include/DynamicProgram.hpp:74, src/DynamicProgram.cpp:67-173): scores[level][filter] in,
// rootv[level][component] (T) and rooti[level][component] (int32, as a T-independent IMat) out; the back-pointers stay on
// the device for argmin().
template <class Tr>
inline void dp_min(pbd_handle *h, int nfilters, int ncomponents, const std::vector<std::vector<typename Tr::Mat> > &scores,
                   std::vector<std::vector<typename Tr::Mat> > &rootv, std::vector<std::vector<typename Tr::IMat> > &rooti)
{
    typedef typename Tr::Real T;
    const int M = (int)scores.size();
    std::vector<int> rows(M), cols(M);
    std::vector<std::vector<T> > packed(M), rv(M);
    std::vector<std::vector<int32_t> > ri(M);
    std::vector<const void *> sp(M);
    std::vector<void *> rvp(M);
    std::vector<int32_t *> rip(M);
    for (int m = 0; m < M; ++m) {
        rows[m] = Tr::rows(scores[m][0]);
        cols[m] = Tr::cols(scores[m][0]);
        const size_t hw = (size_t)rows[m] * cols[m];
        packed[m].resize(hw * nfilters + 1);
        for (int f = 0; f < nfilters; ++f) {
            const typename Tr::Mat c = Tr::real_continuous(scores[m][f]);
            const T *src = Tr::cptr(c);
            for (size_t i = 0; i < hw; ++i) packed[m][f * hw + i] = src[i];
        }
        rv[m].resize(hw * ncomponents + 1);
        ri[m].resize(hw * ncomponents + 1);
        sp[m] = &packed[m][0]; rvp[m] = &rv[m][0]; rip[m] = &ri[m][0];
    }
    check<Tr>(h, pbd_dp_min(h, M, M ? &rows[0] : NULL, M ? &cols[0] : NULL, M ? &sp[0] : NULL, NULL, NULL, NULL,
                            M ? &rvp[0] : NULL, M ? &rip[0] : NULL));
    rootv.assign(M, std::vector<typename Tr::Mat>(ncomponents));
    rooti.assign(M, std::vector<typename Tr::IMat>(ncomponents));
    for (int m = 0; m < M; ++m)
        for (int c = 0; c < ncomponents; ++c) {
            const size_t hw = (size_t)rows[m] * cols[m];
            Tr::create(rootv[m][c], rows[m], cols[m]);
            Tr::icreate(rooti[m][c], rows[m], cols[m]);
            T *dv = Tr::ptr(rootv[m][c]);
            int32_t *di = Tr::iptr(rooti[m][c]);
            for (size_t i = 0; i < hw; ++i) { dv[i] = rv[m][c * hw + i]; di[i] = ri[m][c * hw + i]; }
        }
}

// pbd_candidate records -> the host's Candidates (include/Candidate.hpp:56-80)
template <class Tr>
inline void unpack_candidates(pbd_handle *h, const std::vector<int32_t> &buf, int n, std::vector<typename Tr::Candidate> &out)
{
    const int stride = pbd_candidate_stride(h);
    for (int i = 0; i < n; ++i) {
        const int32_t *r = &buf[(size_t)i * stride];
        pbd_candidate_hdr hd;
        for (int w = 0; w < 8; ++w) reinterpret_cast<int32_t *>(&hd)[w] = r[w];
        Tr::candidate(out, hd, r + 8);
    }
}

// DynamicProgram<T>::argmin (include/DynamicProgram.hpp:75, src/DynamicProgram.cpp:190-255) on the device-resident result
template <class Tr>
inline void dp_argmin(pbd_handle *h, const std::vector<float> &scales, std::vector<typename Tr::Candidate> &candidates,
                      int capacity)
{
    std::vector<int32_t> buf((size_t)capacity * pbd_candidate_stride(h) + 1);
    int n = 0;
    check<Tr>(h, pbd_dp_argmin(h, scales.empty() ? NULL : &scales[0], &buf[0], capacity, &n));
    unpack_candidates<Tr>(h, buf, n, candidates);
}

// PartsBasedDetector<T>::detect (include/PartsBasedDetector.hpp:172-173, src/PartsBasedDetector.cpp:69-95): the whole path
// on the GPU, only Candidates come back.  Any accepted image depth (pbd_detect_typed).
template <class Tr>
inline void detect(pbd_handle *h, const typename Tr::Image &im, std::vector<typename Tr::Candidate> &candidates,
                   int capacity)
{
    std::vector<int32_t> buf((size_t)capacity * pbd_candidate_stride(h) + 1);
    int n = 0;
    check<Tr>(h, pbd_detect_typed(h, Tr::img_data(im), Tr::img_rows(im), Tr::img_cols(im), Tr::img_channels(im),
                                  Tr::img_step(im), Tr::img_depth(im), &buf[0], capacity, &n));
    unpack_candidates<Tr>(h, buf, n, candidates);
}

}  // namespace pbdbind
#endif  // PBD_BIND_HPP_
