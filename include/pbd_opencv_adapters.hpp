// pbd_opencv_adapters.hpp -- header-only adapters that drop the HIP path into the reference's C++ host.
//
// Compiled inside the reference tree (it needs OpenCV and the reference's own headers).  This header supplies ONLY the
// cv::Mat traits and the class shells; every line that crosses the C ABI lives in pbd_bind.hpp, the same templates
// include/pbd_host.hpp instantiates with plain buffers and tests/test_host_demo.py runs on the GPU (T = float and double).
// tests/test_adapters_compile.py compiles this very file (both T) against the declarations in tests/adapter_doubles/.
// See INTEGRATION.md for the two-line change in src/PartsBasedDetector.cpp:108,111 that installs these classes.
//
//   HipHOGFeatures<T>        : IFeatures            (include/IFeatures.hpp:49-73)
//   HipConvolutionEngine<T>  : IConvolutionEngine   (include/IConvolutionEngine.hpp:44-68)
//   hipDetect<T>()           : whole PartsBasedDetector<T>::detect on the GPU (src/PartsBasedDetector.cpp:69-95)
//   T = float (src/demo.cpp:85) or double (cells/detect.cpp:93, ros/Node.hpp:121)
#pragma once

#include <opencv2/core/core.hpp>

#include <string>
#include <vector>

#include "Candidate.hpp"
#include "IConvolutionEngine.hpp"
#include "IFeatures.hpp"
#include "Model.hpp"
#include "pbd.h"
#include "pbd_bind.hpp"

namespace pbd_adapters {

template <typename T>
struct CvTraits {
    typedef T Real;
    typedef cv::Mat Mat;
    typedef cv::Mat IMat;
    typedef cv::Mat FilterMat;
    typedef cv::Mat Image;
    typedef ::Candidate Candidate;
    static int type() { return cv::DataType<T>::type; }                     // CV_32F / CV_64F, as the reference's DataType<T>::type
    static void create(Mat &m, int rows, int cols) { m.create(rows, cols, type()); }   // cv::Mat::create is continuous
    static T *ptr(Mat &m) { return m.ptr<T>(0); }
    static const T *cptr(const Mat &m) { return m.ptr<T>(0); }
    static int rows(const Mat &m) { return m.rows; }
    static int cols(const Mat &m) { return m.cols; }
    static void icreate(IMat &m, int rows, int cols) { m.create(rows, cols, CV_32S); }
    static int32_t *iptr(IMat &m) { return m.ptr<int32_t>(0); }
    static Mat real_continuous(const Mat &m)
    {
        Mat r = m;
        if (r.depth() != cv::DataType<T>::depth) m.convertTo(r, type());    // src/PartsBasedDetector.cpp:114-117
        if (!r.isContinuous()) r = r.clone();
        return r;
    }
    static Mat rows_view(Mat &m, int r0, int r1) { return m.rowRange(r0, r1); }        // a view: no copy
    static const void *img_data(const Image &im) { return im.data; }
    static int img_rows(const Image &im) { return im.rows; }
    static int img_cols(const Image &im) { return im.cols; }
    static int img_channels(const Image &im) { return im.channels(); }
    static size_t img_step(const Image &im) { return im.step; }
    static int img_depth(const Image &im) { return im.depth(); }
    static void fail(int /*rc*/, const std::string &text) { CV_Error(CV_StsError, text); }   // the reference's error channel
    static void candidate(std::vector<Candidate> &out, const pbd_candidate_hdr &hd, const int32_t *r)
    {
        Candidate c;
        c.setComponent(hd.component);
        for (int p = 0; p < hd.nparts; ++p)
            c.addPart(cv::Rect(r[4 * p], r[4 * p + 1], r[4 * p + 2], r[4 * p + 3]), p == 0 ? hd.score : 0.0f);
        out.push_back(c);
    }
    static int filter_rows(const FilterMat &w) { return w.rows; }
    static void filter_values(const FilterMat &w, std::vector<double> &out)
    {
        cv::Mat d;
        w.convertTo(d, CV_64F);
        for (int r = 0; r < d.rows; ++r) out.insert(out.end(), d.ptr<double>(r), d.ptr<double>(r) + d.cols);
    }
};

// owns the pbd_handle (member of the detector: boost::scoped_ptr<pbd_adapters::Handle<T> > hip_)
template <typename T>
class Handle {
public:
    pbd_handle *h;
    explicit Handle(Model &model, int device = 0, int conv_mode = PBD_CONV_EXACT, int max_batch = 1)
        : h(pbdbind::create<CvTraits<T> >(model, device, conv_mode, max_batch, 1 << 18)) {}
    ~Handle() { pbd_destroy(h); }
private:
    Handle(const Handle &);
    Handle &operator=(const Handle &);
};

template <typename T>
class HipHOGFeatures : public IFeatures {
    pbd_handle *h_;
    size_t binsize_;
    vectorf scales_;
public:
    explicit HipHOGFeatures(pbd_handle *h) : h_(h), binsize_(pbd_binsize(h)) {}
    size_t binsize(void) const { return binsize_; }
    size_t nscales(void) const { return scales_.size(); }
    vectorf scales(void) const { return scales_; }
    void pyramid(const cv::Mat &im, vectorMat &pyrafeatures) { pbdbind::pyramid<CvTraits<T> >(h_, im, pyrafeatures, scales_); }
};

template <typename T>
class HipConvolutionEngine : public IConvolutionEngine {
    pbd_handle *h_;
    size_t nfilters_;
public:
    explicit HipConvolutionEngine(pbd_handle *h) : h_(h), nfilters_(0) {}
    void setFilters(const vectorMat &filters)
    {
        pbdbind::set_filters<CvTraits<T> >(h_, filters);
        nfilters_ = filters.size();
    }
    void pdf(const vectorMat &features, vector2DMat &responses) { pbdbind::pdf<CvTraits<T> >(h_, nfilters_, features, responses); }
};

// Whole detect() on the GPU: pyramid -> pdf -> min -> argmin, only Candidates come back.
template <typename T>
inline void hipDetect(pbd_handle *h, const cv::Mat &im, vectorCandidate &candidates)
{
    pbdbind::detect<CvTraits<T> >(h, im, candidates, 1 << 16);
}

}  // namespace pbd_adapters
