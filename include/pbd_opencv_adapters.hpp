// pbd_opencv_adapters.hpp -- header-only adapters that drop the HIP path into the reference's C++ host.
//
// Compiled only inside the reference tree (it needs OpenCV and the reference's own headers); nothing in
// this repository includes it.  See INTEGRATION.md for the two-line change in
// src/PartsBasedDetector.cpp:108,111 that installs these classes.
//
//   HipHOGFeatures        : IFeatures            (include/IFeatures.hpp:49-73)
//   HipConvolutionEngine  : IConvolutionEngine   (include/IConvolutionEngine.hpp:44-68)
//   hipDetect()           : whole PartsBasedDetector<float>::detect on the GPU (src/PartsBasedDetector.cpp:69-95)
#pragma once

#include <opencv2/core/core.hpp>

#include <stdexcept>
#include <string>
#include <vector>

#include "Candidate.hpp"
#include "IConvolutionEngine.hpp"
#include "IFeatures.hpp"
#include "Model.hpp"
#include "pbd.h"

namespace pbd_adapters {

inline void check(pbd_handle *h, int rc)
{   // the C ABI never throws; the reference signals errors with cv::Exception (CV_Error)
    if (rc != PBD_OK) CV_Error(CV_StsError, std::string("pbd: ") + pbd_last_error(h));
}

// Model -> pbd_model (flattened Parts tables, include/Parts.hpp:172-187)
struct FlatModel {
    std::vector<int> ksize, part_offset, parentid, mix_offset, filterid, biasid, defid, anchors;
    std::vector<int64_t> foff;
    std::vector<float> filters, biasw, defw;
    pbd_model m;
    explicit FlatModel(Model &model)
    {
        const int flen = model.flen();
        int64_t off = 0;
        for (size_t f = 0; f < model.filters().size(); ++f) {
            cv::Mat w;
            model.filters()[f].convertTo(w, CV_32F);          // src/PartsBasedDetector.cpp:114-117
            ksize.push_back(w.rows);
            foff.push_back(off);
            for (int r = 0; r < w.rows; ++r) filters.insert(filters.end(), w.ptr<float>(r), w.ptr<float>(r) + w.cols);
            off += (int64_t)w.rows * w.cols;
        }
        biasw = model.bias();
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
                const vectori &fid = model.filterid()[c][p], &bid = model.biasid()[c][p], &did = model.defid()[c][p];
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
        m.flen = flen;
        m.filter_ksize = ksize.data(); m.filter_offset = foff.data();
        m.filters_f32 = filters.data(); m.filters_f64 = NULL;
        m.nbias = (int)biasw.size(); m.biasw = biasw.data();
        m.ndefs = (int)(defw.size() / 4); m.defw = defw.data(); m.anchors = anchors.data();
        m.part_offset = part_offset.data(); m.parentid = parentid.data(); m.mix_offset = mix_offset.data();
        m.filterid = filterid.data(); m.biasid = biasid.data(); m.defid = defid.data();
        m.thresh = model.thresh(); m.sbin = model.binsize(); m.interval = model.nscales(); m.norient = model.norient();
    }
};

class Handle {
public:
    pbd_handle *h;
    explicit Handle(Model &model, int device = 0, int max_batch = 1) : h(NULL)
    {
        FlatModel fm(model);
        pbd_config cfg = {device, PBD_REAL_F32, PBD_CONV_EXACT, max_batch, 1 << 18, NULL};
        if (pbd_create(&fm.m, &cfg, &h) != PBD_OK) CV_Error(CV_StsError, std::string("pbd_create: ") + pbd_last_error(NULL));
    }
    ~Handle() { pbd_destroy(h); }
private:
    Handle(const Handle &);
    Handle &operator=(const Handle &);
};

class HipHOGFeatures : public IFeatures {
    pbd_handle *h_;
    size_t binsize_;
    vectorf scales_;
public:
    explicit HipHOGFeatures(pbd_handle *h) : h_(h), binsize_(pbd_binsize(h)) {}
    size_t binsize(void) const { return binsize_; }
    size_t nscales(void) const { return scales_.size(); }
    vectorf scales(void) const { return scales_; }
    void pyramid(const cv::Mat &im, vectorMat &pyrafeatures)
    {
        int n = 0, fr[PBD_MAX_LEVELS], fc[PBD_MAX_LEVELS];
        float sc[PBD_MAX_LEVELS];
        check(h_, pbd_pyramid_plan(h_, im.rows, im.cols, &n, NULL, NULL, fr, fc, sc));
        pyrafeatures.resize(n);
        std::vector<float *> ptrs(n);
        for (int l = 0; l < n; ++l) {
            pyrafeatures[l].create(fr[l], fc[l] * 32, CV_32F);      // Mat(H, W*flen), src/HOGFeatures.cpp:180
            ptrs[l] = pyrafeatures[l].ptr<float>(0);
        }
        // the depth codes of the C ABI are cv::Mat::depth() itself: CV_8U / CV_16U / CV_32F / CV_64F are the four
        // features<IT> instantiations (src/HOGFeatures.cpp:136-146); any other depth -> PBD_ERR_UNSUPPORTED (the
        // reference's default: branch, :141-145)
        check(h_, pbd_features_pyramid(h_, im.data, im.rows, im.cols, im.channels(), im.step, im.depth(), ptrs.data()));
        scales_.assign(sc, sc + n);
    }
};

class HipConvolutionEngine : public IConvolutionEngine {
    pbd_handle *h_;
    size_t nfilters_;
public:
    explicit HipConvolutionEngine(pbd_handle *h) : h_(h), nfilters_(0) {}
    void setFilters(const vectorMat &filters)
    {
        std::vector<cv::Mat> f32(filters.size());
        std::vector<const float *> ptrs(filters.size());
        std::vector<int> ks(filters.size());
        for (size_t f = 0; f < filters.size(); ++f) {
            filters[f].convertTo(f32[f], CV_32F);
            if (!f32[f].isContinuous()) f32[f] = f32[f].clone();
            ptrs[f] = f32[f].ptr<float>(0);
            ks[f] = f32[f].rows;
        }
        check(h_, pbd_conv_set_filters(h_, (int)filters.size(), ptrs.data(), ks.data()));
        nfilters_ = filters.size();
    }
    void pdf(const vectorMat &features, vector2DMat &responses)
    {
        const int M = (int)features.size();
        std::vector<const float *> fp(M);
        std::vector<float *> rp(M);
        std::vector<int> rows(M), cols(M);
        std::vector<cv::Mat> cont(M), packed(M);
        for (int m = 0; m < M; ++m) {
            cont[m] = features[m].isContinuous() ? features[m] : features[m].clone();
            rows[m] = cont[m].rows; cols[m] = cont[m].cols / 32;
            fp[m] = cont[m].ptr<float>(0);
            packed[m].create((int)nfilters_ * rows[m], cols[m], CV_32F);
            rp[m] = packed[m].ptr<float>(0);
        }
        check(h_, pbd_conv_pdf(h_, M, fp.data(), rows.data(), cols.data(), rp.data()));
        responses.assign(M, vectorMat(nfilters_));
        for (int m = 0; m < M; ++m)        // responses[level][filter], each a view of the packed planes
            for (size_t n = 0; n < nfilters_; ++n) responses[m][n] = packed[m].rowRange((int)n * rows[m], (int)(n + 1) * rows[m]);
    }
};

// Whole detect() on the GPU: pyramid -> pdf -> min -> argmin, only Candidates come back.
inline void hipDetect(pbd_handle *h, const cv::Mat &im, vectorCandidate &candidates)
{
    const int stride = pbd_candidate_stride(h), cap = 1 << 16;
    std::vector<int32_t> buf((size_t)cap * stride);
    int n = 0;
    check(h, pbd_detect_typed(h, im.data, im.rows, im.cols, im.channels(), im.step, im.depth(), buf.data(), cap, &n));
    for (int i = 0; i < n; ++i) {
        const int32_t *r = &buf[(size_t)i * stride];
        const pbd_candidate_hdr *hd = reinterpret_cast<const pbd_candidate_hdr *>(r);
        Candidate c;
        c.setComponent(hd->component);
        for (int p = 0; p < hd->nparts; ++p)
            c.addPart(cv::Rect(r[8 + 4 * p], r[9 + 4 * p], r[10 + 4 * p], r[11 + 4 * p]), p == 0 ? hd->score : 0.0f);
        candidates.push_back(c);
    }
}

}  // namespace pbd_adapters
