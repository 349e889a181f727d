// demo.cpp -- command-line harness with the flow of the reference's src/demo.cpp:55-117:
//   model file -> FileStorageModel::deserialize -> PartsBasedDetector<T>::distributeModel -> read image ->
//   detect -> "Number of candidates" -> Candidate::sort [-> nonMaximaSuppression] -> list the best ones.
// The GUI part of the reference's demo (Visualize, highgui) is out of scope.
//
//   pbd_demo model.yml image.(ppm|pgm) [--double] [--nms OVERLAP] [--top N] [--staged]
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "pbd_host.hpp"

using namespace pbdhost;

template <typename T>
static int run(FileStorageModel &model, const Image &im, bool staged, float nms, int top)
{
    PartsBasedDetector<T> pbd;
    pbd.distributeModel(model);
    std::vector<Candidate> candidates;
    if (staged) {
        // the four calls of PartsBasedDetector<T>::detect (src/PartsBasedDetector.cpp:73-89) through the engine mirrors
        HOGFeatures<T> features(pbd.handle());
        SpatialConvolutionEngine<T> conv(pbd.handle(), model.filtersw_.size());
        DynamicProgram<T> dp(pbd.handle(), (int)model.filtersw_.size());
        std::vector<MatT<T> > pyramid;
        features.pyramid(im, pyramid);
        std::vector<std::vector<MatT<T> > > pdf, rootv;
        std::vector<std::vector<MatT<int> > > rooti;
        conv.pdf(pyramid, pdf);
        dp.min(pdf, rootv, rooti, model.ncomponents());
        dp.argmin(features.scales(), candidates);
    } else {
        pbd.detect(im, candidates);
    }
    std::printf("Number of candidates: %zu\n", candidates.size());
    Candidate::sort(candidates);
    if (nms >= 0) {
        Candidate::nonMaximaSuppression(im.rows, im.cols, candidates, nms);
        std::printf("After NMS: %zu\n", candidates.size());
    }
    for (size_t i = 0; i < candidates.size() && (int)i < top; ++i) {
        const Candidate &c = candidates[i];
        std::printf("cand %d %d %d %d %.9g", c.level, c.component(), c.root_y, c.root_x, (double)c.score());
        for (size_t p = 0; p < c.parts().size(); ++p)
            std::printf(" %d,%d,%d,%d", c.parts()[p].x, c.parts()[p].y, c.parts()[p].width, c.parts()[p].height);
        std::printf("\n");
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "Usage: pbd_demo model_file image_file [--double] [--nms overlap] [--top n] [--staged]\n");
        return -1;
    }
    bool dbl = false, staged = false;
    float nms = -1.f;
    int top = 1 << 30;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--double")) dbl = true;
        else if (!std::strcmp(argv[i], "--staged")) staged = true;
        else if (!std::strcmp(argv[i], "--nms") && i + 1 < argc) nms = (float)std::atof(argv[++i]);
        else if (!std::strcmp(argv[i], "--top") && i + 1 < argc) top = std::atoi(argv[++i]);
    }
    try {
        FileStorageModel model;
        if (!model.deserialize(argv[1])) { std::fprintf(stderr, "Error deserializing file\n"); return -1; }
        std::vector<uint8_t> pix;
        Image im;
        if (!readPNM(argv[2], pix, im)) { std::fprintf(stderr, "Image not found, or invalid image format\n"); return -1; }
        return dbl ? run<double>(model, im, staged, nms, top) : run<float>(model, im, staged, nms, top);
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return -2;
    }
}
