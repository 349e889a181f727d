// demo.cpp -- command-line harness with the flow of the reference's src/demo.cpp:55-117:
//   model file -> FileStorageModel::deserialize -> PartsBasedDetector<T>::distributeModel -> read image ->
//   detect -> "Number of candidates" -> Candidate::sort [-> nonMaximaSuppression] -> list the best ones.
// The GUI part of the reference's demo (Visualize, highgui) is out of scope.
//
//   pbd_demo model.(yml|xml) image.(ppm|pgm) [--double] [--nms OVERLAP] [--top N] [--staged] [--stream HANDLES FRAMES]
//   pbd_demo model.(yml|xml) --dump-model      (no GPU needed: prints what FileStorageModel::deserialize read)
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "pbd_host.hpp"

using namespace pbdhost;

template <typename T>
static int run(FileStorageModel &model, const Image &im, bool staged, float nms, int top, int stream_k, int stream_n)
{
    PartsBasedDetector<T> pbd;
    std::vector<Candidate> candidates;
    if (stream_k > 0) {
        // the image stream_n times through a FrameStream of stream_k handles: every result must be the first one's
        FrameStream<T> fs(model, stream_k);
        std::vector<Candidate> first, cur;
        size_t got = 0;
        bool same = true;
        auto cmp = [&](const std::vector<Candidate> &a, const std::vector<Candidate> &b) {
            if (a.size() != b.size()) return false;
            for (size_t i = 0; i < a.size(); ++i) {
                if (a[i].level != b[i].level || a[i].root_x != b[i].root_x || a[i].root_y != b[i].root_y || a[i].score() != b[i].score() ||
                    a[i].parts().size() != b[i].parts().size()) return false;
                for (size_t p = 0; p < a[i].parts().size(); ++p)
                    if (a[i].parts()[p].x != b[i].parts()[p].x || a[i].parts()[p].y != b[i].parts()[p].y ||
                        a[i].parts()[p].width != b[i].parts()[p].width || a[i].parts()[p].height != b[i].parts()[p].height) return false;
            }
            return true;
        };
        auto collect = [&]() {
            fs.next(cur);
            if (got++ == 0) first = cur; else same = same && cmp(first, cur);
        };
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < stream_n; ++i) {
            while (fs.full()) collect();
            fs.submit(im);
        }
        while (fs.pending()) collect();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("stream: %d frames over %d handles, %.1f frames/s, results %s\n", stream_n, stream_k, stream_n / sec, same ? "identical" : "DIFFER");
        if (!same) return -3;
        candidates = first;
    } else if (staged) {
        pbd.distributeModel(model);
        // the four calls of PartsBasedDetector<T>::detect (src/PartsBasedDetector.cpp:73-89) through the engine mirrors
        HOGFeatures<T> features(pbd.handle());
        SpatialConvolutionEngine<T> conv(pbd.handle(), model.filtersw_.size());
        DynamicProgram<T> dp(pbd.handle(), (int)model.filtersw_.size());
        std::vector<MatT<T> > pyramid;
        features.pyramid(im, pyramid);
        std::vector<std::vector<MatT<T> > > pdf, rootv;
        std::vector<std::vector<MatT<int32_t> > > rooti;
        conv.pdf(pyramid, pdf);
        dp.min(pdf, rootv, rooti, model.ncomponents());
        dp.argmin(features.scales(), candidates);
    } else {
        pbd.distributeModel(model);
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

// every field FileStorageModel::deserialize fills, as text (doubles with 17 significant digits: exact round trip)
static int dump_model(const FileStorageModel &m)
{
    std::printf("name %s\ninterval %d\nthresh %.9g\nsbin %d\nnorient %d\nflen %d\n", m.name().c_str(), m.nscales(), (double)m.thresh(),
                m.binsize(), m.norient(), m.flen());
    for (size_t f = 0; f < m.filtersw_.size(); ++f) {
        std::printf("filter %zu %d %d", f, m.filtersw_[f].rows, m.filtersw_[f].cols);
        for (size_t i = 0; i < m.filtersw_[f].data.size(); ++i) std::printf(" %.17g", m.filtersw_[f].data[i]);
        std::printf("\n");
    }
    std::printf("biasw");
    for (size_t i = 0; i < m.biasw_.size(); ++i) std::printf(" %.9g", (double)m.biasw_[i]);
    std::printf("\nanchors");
    for (size_t i = 0; i < m.anchors_.size(); ++i) std::printf(" %d,%d", m.anchors_[i].x, m.anchors_[i].y);
    std::printf("\n");
    for (size_t d = 0; d < m.defw_.size(); ++d) {
        std::printf("def %zu", d);
        for (size_t i = 0; i < m.defw_[d].size(); ++i) std::printf(" %.9g", (double)m.defw_[d][i]);
        std::printf("\n");
    }
    for (size_t c = 0; c < m.filterid_.size(); ++c)
        for (size_t p = 0; p < m.filterid_[c].size(); ++p) {
            std::printf("part %zu %zu parent %d filterid", c, p, m.parentid_[c][p]);
            for (size_t i = 0; i < m.filterid_[c][p].size(); ++i) std::printf(" %d", m.filterid_[c][p][i]);
            std::printf(" biasid");
            for (size_t i = 0; i < m.biasid_[c][p].size(); ++i) std::printf(" %d", m.biasid_[c][p][i]);
            std::printf(" defid");
            for (size_t i = 0; i < m.defid_[c][p].size(); ++i) std::printf(" %d", m.defid_[c][p][i]);
            std::printf("\n");
        }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "Usage: pbd_demo model_file image_file [--double] [--nms overlap] [--top n] [--staged] [--stream handles frames]\n");
        return -1;
    }
    bool dbl = false, staged = false;
    float nms = -1.f;
    int top = 1 << 30, stream_k = 0, stream_n = 0;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--double")) dbl = true;
        else if (!std::strcmp(argv[i], "--staged")) staged = true;
        else if (!std::strcmp(argv[i], "--nms") && i + 1 < argc) nms = (float)std::atof(argv[++i]);
        else if (!std::strcmp(argv[i], "--top") && i + 1 < argc) top = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--stream") && i + 2 < argc) { stream_k = std::atoi(argv[++i]); stream_n = std::atoi(argv[++i]); }
    }
    try {
        FileStorageModel model;
        if (!model.deserialize(argv[1])) { std::fprintf(stderr, "Error deserializing file\n"); return -1; }
        if (!std::strcmp(argv[2], "--dump-model")) return dump_model(model);
        std::vector<uint8_t> pix;
        Image im;
        if (!readPNM(argv[2], pix, im)) { std::fprintf(stderr, "Image not found, or invalid image format\n"); return -1; }
        return dbl ? run<double>(model, im, staged, nms, top, stream_k, stream_n) : run<float>(model, im, staged, nms, top, stream_k, stream_n);
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return -2;
    }
}
