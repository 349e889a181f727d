// Translation unit of tests/test_adapters_compile.py: every member of every adapter class, for both real types.
#include "pbd_opencv_adapters.hpp"

template struct pbd_adapters::CvTraits<float>;
template struct pbd_adapters::CvTraits<double>;
template class pbd_adapters::Handle<float>;
template class pbd_adapters::Handle<double>;
template class pbd_adapters::HipHOGFeatures<float>;
template class pbd_adapters::HipHOGFeatures<double>;
template class pbd_adapters::HipConvolutionEngine<float>;
template class pbd_adapters::HipConvolutionEngine<double>;
template void pbd_adapters::hipDetect<float>(pbd_handle *, const cv::Mat &, vectorCandidate &);
template void pbd_adapters::hipDetect<double>(pbd_handle *, const cv::Mat &, vectorCandidate &);
// the staged DynamicProgram binding is used by pbd_host.hpp only, but must type-check with cv::Mat as well
template void pbdbind::dp_min<pbd_adapters::CvTraits<double> >(pbd_handle *, int, int, const vector2DMat &, vector2DMat &, vector2DMat &);
template void pbdbind::dp_argmin<pbd_adapters::CvTraits<float> >(pbd_handle *, const std::vector<float> &, vectorCandidate &, int);

// the members the reference's distributeModel would hold, T = double as cells/detect.cpp:93
void install(Model &model, IFeatures *&features, IConvolutionEngine *&conv, pbd_adapters::Handle<double> *&hip)
{
    hip = new pbd_adapters::Handle<double>(model);
    features = new pbd_adapters::HipHOGFeatures<double>(hip->h);
    conv = new pbd_adapters::HipConvolutionEngine<double>(hip->h);
}
