// pbd_host.hpp -- C++ host-side mirror of the reference's operator interface for the detection hot
// path, over the C ABI (pbd.h).  Header-only, no OpenCV, no Boost.  Class and method names follow the
// reference so that host code reads the same:
//
//   Model / FileStorageModel     include/Model.hpp:49-122, src/FileStorageModel.cpp:42-159 (YAML flavour)
//   HOGFeatures                  include/IFeatures.hpp:49-73 (binsize, nscales, scales, pyramid)
//   SpatialConvolutionEngine     include/IConvolutionEngine.hpp:44-68 (setFilters, pdf)
//   DynamicProgram               include/DynamicProgram.hpp:74-75 (min, argmin)
//   PartsBasedDetector           include/PartsBasedDetector.hpp:152-175 (distributeModel, detect, name)
//   Candidate                    include/Candidate.hpp:56-111,277-304 (score, sort, boundingBox, nonMaximaSuppression)
//
// Errors: the reference uses assert / CV_Error -> cv::Exception; here every failure of the C ABI is
// thrown as pbdhost::Error (a std::runtime_error) carrying pbd_last_error().
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "pbd.h"
#include "pbd_bind.hpp"   // the ABI-calling bodies, shared with the OpenCV adapters (pbd_opencv_adapters.hpp)

namespace pbdhost {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// ---------------------------------------------------------------------------------------------- basics
struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
    int area() const { return width * height; }
    bool empty() const { return width <= 0 || height <= 0; }
    Rect operator|(const Rect &b) const
    {   // cv::Rect union
        if (empty()) return b;
        if (b.empty()) return *this;
        const int x1 = std::min(x, b.x), y1 = std::min(y, b.y);
        return Rect(x1, y1, std::max(x + width, b.x + b.width) - x1, std::max(y + height, b.y + b.height) - y1);
    }
    Rect operator&(const Rect &b) const
    {   // cv::Rect intersection
        const int x1 = std::max(x, b.x), y1 = std::max(y, b.y);
        const int w = std::min(x + width, b.x + b.width) - x1, h = std::min(y + height, b.y + b.height) - y1;
        return (w <= 0 || h <= 0) ? Rect() : Rect(x1, y1, w, h);
    }
};

template <typename T>
struct MatT {   // dense row-major matrix (the role cv::Mat_<T> plays at the reference's seams)
    int rows, cols;
    std::vector<T> data;
    MatT() : rows(0), cols(0) {}
    MatT(int r, int c) : rows(r), cols(c), data((size_t)r * c) {}
    T *ptr(int r = 0) { return data.data() + (size_t)r * cols; }
    const T *ptr(int r = 0) const { return data.data() + (size_t)r * cols; }
};

struct Point {   // cv::Point at the reference's seams (Model::anchors, include/Model.hpp:66)
    int x, y;
    Point() : x(0), y(0) {}
    Point(int x_, int y_) : x(x_), y(y_) {}
};

struct Image {   // image view: rows x cols x channels, BGR interleaved when channels == 3
    const void *data;
    int rows, cols, channels;
    size_t step;   // bytes between rows (cv::Mat::step)
    int depth;     // cv::Mat::depth(): 0 = 8U, 2 = 16U, 5 = 32F, 6 = 64F (src/HOGFeatures.cpp:136-146)
    Image() : data(NULL), rows(0), cols(0), channels(0), step(0), depth(0) {}
};

class Candidate {
public:
    std::vector<Rect> parts_;
    std::vector<float> confidence_;
    int component_;
    int frame, level, root_x, root_y;   // where the candidate was back-tracked from
    Candidate() : component_(0), frame(0), level(0), root_x(0), root_y(0) {}
    const std::vector<Rect> &parts() const { return parts_; }
    const std::vector<float> &confidence() const { return confidence_; }
    void addPart(Rect r, float confidence) { parts_.push_back(r); confidence_.push_back(confidence); }
    float score() const { return confidence_.size() ? confidence_[0] : -std::numeric_limits<float>::infinity(); }
    int component() const { return component_; }
    static bool descending(const Candidate &a, const Candidate &b) { return a.score() > b.score(); }
    static void sort(std::vector<Candidate> &c) { std::sort(c.begin(), c.end(), descending); }
    Rect boundingBox() const
    {
        Rect hull = parts_[0];
        for (size_t n = 0; n < parts_.size(); ++n) hull = hull | parts_[n];
        return hull;
    }
    // greedy paint-the-canvas suppression, include/Candidate.hpp:277-304
    static void nonMaximaSuppression(int rows, int cols, std::vector<Candidate> &candidates, float overlap = 0.0f)
    {
        const Rect bounds(0, 0, cols, rows);
        std::vector<uint8_t> scratch((size_t)rows * cols, 0);
        size_t keep = 0;
        for (size_t n = 0; n < candidates.size(); ++n) {
            const Rect box = candidates[n].boundingBox() & bounds;
            double boxsum = 0;
            for (int y = box.y; y < box.y + box.height; ++y)
                for (int x = box.x; x < box.x + box.width; ++x) boxsum += scratch[(size_t)y * cols + x];
            if (boxsum / box.area() > overlap) continue;   // 0/0 = NaN compares false: an empty box is kept
            for (int y = box.y; y < box.y + box.height; ++y)
                for (int x = box.x; x < box.x + box.width; ++x) scratch[(size_t)y * cols + x] = 1;
            candidates[keep++] = candidates[n];
        }
        candidates.resize(keep);
    }
};

// ---------------------------------------------------------------------------------------------- model
class Model {
public:
    std::string name_;
    int nscales_ = 10;   // the interval (src/FileStorageModel.cpp:105)
    float thresh_ = 0;
    int binsize_ = 4, flen_ = 32, norient_ = 18;
    std::vector<MatT<double> > filtersw_;
    std::vector<float> biasw_;
    std::vector<Point> anchors_;
    std::vector<std::vector<float> > defw_;
    std::vector<std::vector<std::vector<int> > > biasid_, filterid_, defid_;
    std::vector<std::vector<int> > parentid_;

    // accessors named as the reference's (include/Model.hpp:99-121)
    std::vector<MatT<double> > &filters() { return filtersw_; }
    std::vector<float> &bias() { return biasw_; }
    std::vector<std::vector<float> > &def() { return defw_; }
    std::vector<Point> &anchors() { return anchors_; }
    std::vector<std::vector<std::vector<int> > > &filterid() { return filterid_; }
    std::vector<std::vector<std::vector<int> > > &biasid() { return biasid_; }
    std::vector<std::vector<std::vector<int> > > &defid() { return defid_; }
    std::vector<std::vector<int> > &parentid() { return parentid_; }
    std::string name() const { return name_; }
    float thresh() const { return thresh_; }
    int binsize() const { return binsize_; }
    int nscales() const { return nscales_; }
    int flen() const { return flen_; }
    int norient() const { return norient_; }
    int ncomponents() const { return (int)filterid_.size(); }
    virtual ~Model() {}
};

// YAML flavour of cv::FileStorage as FileStorageModel writes it (subset: block mappings/sequences, flow
// sequences, !!opencv-matrix).  See partsbaseddetector_amd/filestorage.py for the Python twin (+ XML).
class FileStorageModel : public Model {
    struct Node {
        std::string scalar;
        std::vector<Node> seq;
        std::vector<std::pair<std::string, Node> > map;
        bool is_seq = false, is_map = false;
        const Node &operator[](const std::string &k) const
        {
            for (size_t i = 0; i < map.size(); ++i)
                if (map[i].first == k) return map[i].second;
            throw Error(PBD_ERR_INVALID, "model file: missing key '" + k + "'");
        }
        bool has(const std::string &k) const
        {
            for (size_t i = 0; i < map.size(); ++i)
                if (map[i].first == k) return true;
            return false;
        }
        double num() const { return parse_num(scalar); }
        std::vector<double> nums() const
        {
            std::vector<double> v;
            if (is_seq) for (size_t i = 0; i < seq.size(); ++i) { if (seq[i].is_seq) { std::vector<double> s = seq[i].nums(); v.insert(v.end(), s.begin(), s.end()); } else v.push_back(seq[i].num()); }
            else if (!scalar.empty()) v.push_back(num());
            return v;
        }
    };
    static double parse_num(const std::string &t)
    {
        if (t == ".Inf" || t == ".inf" || t == "+.Inf") return std::numeric_limits<double>::infinity();
        if (t == "-.Inf" || t == "-.inf") return -std::numeric_limits<double>::infinity();
        if (t == ".Nan" || t == ".nan" || t == ".NaN") return std::numeric_limits<double>::quiet_NaN();
        return std::strtod(t.c_str(), NULL);
    }
    static std::string trim(const std::string &s)
    {
        size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    static int indent_of(const std::string &s) { return (int)s.find_first_not_of(' '); }
    static Node flow(const std::string &t, size_t &pos)
    {
        Node n; n.is_seq = true;
        ++pos;   // '['
        std::string tok;
        for (;;) {
            const char ch = t[pos];
            if (ch == '[') { n.seq.push_back(flow(t, pos)); tok.clear(); }
            else if (ch == ',' || ch == ']') {
                if (!trim(tok).empty()) { Node s; s.scalar = trim(tok); n.seq.push_back(s); }
                tok.clear();
                ++pos;
                if (ch == ']') return n;
            } else { tok += ch; ++pos; }
        }
    }
    static Node value(const std::string &rest)
    {
        if (!rest.empty() && rest[0] == '[') { size_t p = 0; return flow(rest, p); }
        Node s;
        s.scalar = rest;
        if (s.scalar.size() >= 2 && s.scalar[0] == '"') s.scalar = s.scalar.substr(1, s.scalar.size() - 2);
        return s;
    }
    static Node block(const std::vector<std::string> &L, size_t &i, int indent)
    {
        Node n;
        const bool seq = trim(L[i])[0] == '-';
        n.is_seq = seq; n.is_map = !seq;
        while (i < L.size()) {
            const int cur = indent_of(L[i]);
            if (cur < indent) break;
            const std::string body = trim(L[i]);
            if (seq) {
                std::string item = trim(body.substr(1));
                if (item.compare(0, 15, "!!opencv-matrix") == 0) { ++i; n.seq.push_back(block(L, i, cur + 2)); continue; }
                if (item.empty()) { ++i; n.seq.push_back(block(L, i, cur + 1)); continue; }
                n.seq.push_back(value(item));
                ++i;
            } else {
                const size_t c = body.find(':');
                const std::string key = body.substr(0, c), rest = trim(body.substr(c + 1));
                if (rest.compare(0, 15, "!!opencv-matrix") == 0) { ++i; n.map.push_back(std::make_pair(key, block(L, i, cur + 1))); continue; }
                if (rest.empty()) {
                    if (i + 1 < L.size() && indent_of(L[i + 1]) > cur) {
                        const int nxt = indent_of(L[i + 1]);
                        ++i;
                        n.map.push_back(std::make_pair(key, block(L, i, nxt)));
                        continue;
                    }
                    Node e; e.is_seq = true;
                    n.map.push_back(std::make_pair(key, e));
                } else {
                    n.map.push_back(std::make_pair(key, value(rest)));
                }
                ++i;
            }
        }
        return n;
    }
    static std::vector<int> ints(const Node &n)
    {
        std::vector<double> v = n.nums();
        return std::vector<int>(v.begin(), v.end());
    }

    // ---- the <opencv_storage> XML flavour (what the reference's configs name: conf/config_person.by_parts:30,
    // conf/config_face.by_parts:31).  Elements with children become maps (sequences when every child is <_>),
    // leaves become scalars or, with several whitespace-separated tokens, sequences; attributes are not needed
    // (an opencv-matrix is recognised by its rows / cols / data children).
    static void xml_skip(const std::string &t, size_t &p)
    {   // whitespace, comments, processing instructions
        for (;;) {
            while (p < t.size() && (t[p] == ' ' || t[p] == '\n' || t[p] == '\r' || t[p] == '\t')) ++p;
            if (t.compare(p, 4, "<!--") == 0) { const size_t e = t.find("-->", p); p = e == std::string::npos ? t.size() : e + 3; continue; }
            if (t.compare(p, 2, "<?") == 0) { const size_t e = t.find("?>", p); p = e == std::string::npos ? t.size() : e + 2; continue; }
            return;
        }
    }
    static Node xml_leaf(const std::string &text)
    {
        const std::string body = trim_ws(text);
        Node n;
        if (body.empty()) return n;
        if (body[0] == '"') { n.scalar = body.substr(1, body.rfind('"') - 1); return n; }
        std::vector<std::string> toks;
        std::istringstream is(body);
        for (std::string tk; is >> tk;) toks.push_back(tk);
        if (toks.size() == 1) { n.scalar = toks[0]; return n; }
        n.is_seq = true;
        for (size_t i = 0; i < toks.size(); ++i) { Node e; e.scalar = toks[i]; n.seq.push_back(e); }
        return n;
    }
    static std::string trim_ws(const std::string &s)
    {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    // parses the element starting at t[p] == '<'; returns its tag and value
    static Node xml_element(const std::string &t, size_t &p, std::string &tag)
    {
        if (p >= t.size() || t[p] != '<') throw Error(PBD_ERR_INVALID, "model file: malformed XML");
        const size_t gt = t.find('>', p);
        if (gt == std::string::npos) throw Error(PBD_ERR_INVALID, "model file: unterminated XML tag");
        std::string head = t.substr(p + 1, gt - p - 1);
        const bool self_closed = !head.empty() && head[head.size() - 1] == '/';
        if (self_closed) head.erase(head.size() - 1);
        tag = head.substr(0, head.find_first_of(" \t\r\n"));
        p = gt + 1;
        Node n;
        if (self_closed) return n;
        std::string text;
        std::vector<std::pair<std::string, Node> > kids;
        for (;;) {
            const size_t lt = t.find('<', p);
            if (lt == std::string::npos) throw Error(PBD_ERR_INVALID, "model file: unterminated XML element <" + tag + ">");
            text += t.substr(p, lt - p);
            p = lt;
            if (t.compare(p, 4, "<!--") == 0) { xml_skip(t, p); continue; }
            if (t.compare(p, 2, "<?") == 0) { xml_skip(t, p); continue; }
            if (t.compare(p, 2, "</") == 0) {
                const size_t close = t.find('>', p);
                if (close == std::string::npos) throw Error(PBD_ERR_INVALID, "model file: unterminated closing tag of <" + tag + ">");
                if (trim_ws(t.substr(p + 2, close - p - 2)) != tag)
                    throw Error(PBD_ERR_INVALID, "model file: <" + tag + "> closed by </" + trim_ws(t.substr(p + 2, close - p - 2)) + ">");
                p = close + 1;
                break;
            }
            std::string ktag;
            Node kid = xml_element(t, p, ktag);
            kids.push_back(std::make_pair(ktag, kid));
        }
        if (kids.empty()) return xml_leaf(text);
        bool all_items = true;
        for (size_t i = 0; i < kids.size(); ++i) all_items = all_items && kids[i].first == "_";
        if (all_items) { n.is_seq = true; for (size_t i = 0; i < kids.size(); ++i) n.seq.push_back(kids[i].second); }
        else { n.is_map = true; n.map = kids; }
        return n;
    }
    static Node xml_document(const std::string &t)
    {
        size_t p = 0;
        xml_skip(t, p);
        std::string tag;
        Node root = xml_element(t, p, tag);
        if (tag != "opencv_storage") throw Error(PBD_ERR_INVALID, "model file: root element <" + tag + "> is not <opencv_storage>");
        return root;
    }

public:
    bool deserialize(const std::string &filename)
    {   // src/FileStorageModel.cpp:96-159
        std::ifstream in(filename.c_str());
        if (!in) return false;
        std::stringstream whole;
        whole << in.rdbuf();
        const std::string text = whole.str();
        const size_t first = text.find_first_not_of(" \t\r\n");
        const bool is_xml = first != std::string::npos && text[first] == '<';
        Node doc;
        if (is_xml) {
            doc = xml_document(text);
        } else {
            std::vector<std::string> raw, L;
            std::istringstream lines(text);
            for (std::string ln; std::getline(lines, ln);) {
                if (!ln.empty() && ln[0] == '%') continue;
                if (trim(ln).empty() || trim(ln) == "---") continue;
                raw.push_back(ln);
            }
            int depth = 0;   // join flow sequences wrapped over several lines
            for (size_t i = 0; i < raw.size(); ++i) {
                if (depth == 0) L.push_back(raw[i]); else L.back() += " " + trim(raw[i]);
                for (size_t k = 0; k < raw[i].size(); ++k) depth += (raw[i][k] == '[') - (raw[i][k] == ']');
            }
            size_t i = 0;
            doc = block(L, i, 0);
        }
        name_ = doc.has("name") ? doc["name"].scalar : "";
        nscales_ = (int)doc["interval"].num();
        thresh_ = (float)doc["thresh"].num();
        binsize_ = (int)doc["sbin"].num();
        norient_ = (int)doc["norient"].num();
        flen_ = (int)doc["flen"].num();
        filtersw_.clear();
        const Node &fw = doc["filtersw"];
        for (size_t f = 0; f < fw.seq.size(); ++f) {
            const Node &m = fw.seq[f];
            MatT<double> w((int)m["rows"].num(), (int)m["cols"].num());
            const std::vector<double> d = m["data"].nums();
            if (d.size() != w.data.size()) throw Error(PBD_ERR_INVALID, "model file: filter size mismatch");
            w.data = d;
            filtersw_.push_back(w);
        }
        biasw_.clear();
        { std::vector<double> b = doc["biasw"].nums(); biasw_.assign(b.begin(), b.end()); }
        anchors_.clear();
        { std::vector<double> a = doc["anchors"].nums(); for (size_t k = 0; k + 1 < a.size(); k += 2) anchors_.push_back(Point((int)a[k], (int)a[k + 1])); }
        defw_.clear();
        const Node &defs = doc["defs"];
        for (size_t d = 0; d < defs.seq.size(); ++d) { std::vector<double> w = defs.seq[d].nums(); defw_.push_back(std::vector<float>(w.begin(), w.end())); }
        const Node &comps = doc["indexers"];
        const size_t nc = comps.map.size();
        parentid_.assign(nc, std::vector<int>());
        filterid_.assign(nc, std::vector<std::vector<int> >());
        biasid_ = filterid_; defid_ = filterid_;
        for (size_t c = 0; c < nc; ++c) {
            std::ostringstream cs; cs << "component-" << c;
            const Node &parts = comps[cs.str()];
            for (size_t p = 0; p < parts.map.size(); ++p) {
                std::ostringstream ps; ps << "part-" << p;
                const Node &part = parts[ps.str()];
                parentid_[c].push_back((int)part["parentid"].num());
                filterid_[c].push_back(ints(part["filterid"]));
                biasid_[c].push_back(ints(part["biasid"]));
                // what the writer wrote (scalar, sequence or empty), not the fork's isInt() shortcut
                // that collapses multi-mixture defids to [0] (src/FileStorageModel.cpp:148-152)
                defid_[c].push_back(part.has("defid") ? ints(part["defid"]) : std::vector<int>());
            }
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------- engines
// The traits pbd_bind.hpp's templates are instantiated with here (the OpenCV adapters supply the cv::Mat twin).
template <typename T>
struct HostTraits {
    typedef T Real;
    typedef MatT<T> Mat;
    typedef MatT<int32_t> IMat;
    typedef MatT<double> FilterMat;
    typedef pbdhost::Image Image;
    typedef pbdhost::Candidate Candidate;
    static void create(Mat &m, int rows, int cols) { m = Mat(rows, cols); }
    static T *ptr(Mat &m) { return m.ptr(); }
    static const T *cptr(const Mat &m) { return m.ptr(); }
    static int rows(const Mat &m) { return m.rows; }
    static int cols(const Mat &m) { return m.cols; }
    static void icreate(IMat &m, int rows, int cols) { m = IMat(rows, cols); }
    static int32_t *iptr(IMat &m) { return m.ptr(); }
    static Mat real_continuous(const Mat &m) { return m; }                 // MatT is always continuous T data
    static Mat rows_view(Mat &m, int r0, int r1)
    {   // MatT has no views: a copy
        Mat r(r1 - r0, m.cols);
        std::copy(m.ptr(r0), m.ptr(r0) + r.data.size(), r.data.begin());
        return r;
    }
    static const void *img_data(const Image &im) { return im.data; }
    static int img_rows(const Image &im) { return im.rows; }
    static int img_cols(const Image &im) { return im.cols; }
    static int img_channels(const Image &im) { return im.channels; }
    static size_t img_step(const Image &im) { return im.step; }
    static int img_depth(const Image &im) { return im.depth; }
    static void fail(int rc, const std::string &text) { throw Error(rc, text); }
    static void candidate(std::vector<Candidate> &out, const pbd_candidate_hdr &hd, const int32_t *r)
    {
        Candidate c;
        c.component_ = hd.component; c.frame = hd.frame; c.level = hd.level; c.root_x = hd.root_x; c.root_y = hd.root_y;
        for (int p = 0; p < hd.nparts; ++p) c.addPart(Rect(r[4 * p], r[4 * p + 1], r[4 * p + 2], r[4 * p + 3]), p == 0 ? hd.score : 0.0f);
        out.push_back(c);
    }
    static int filter_rows(const FilterMat &w) { return w.rows; }
    static void filter_values(const FilterMat &w, std::vector<double> &out) { out.insert(out.end(), w.data.begin(), w.data.end()); }
};

template <typename T>
class HOGFeatures {   // IFeatures
    pbd_handle *h_;
    std::vector<float> scales_;
public:
    explicit HOGFeatures(pbd_handle *h) : h_(h) {}
    size_t binsize() const { return (size_t)pbd_binsize(h_); }
    size_t nscales() const { return scales_.size(); }
    std::vector<float> scales() const { return scales_; }
    void pyramid(const Image &im, std::vector<MatT<T> > &pyrafeatures) { pbdbind::pyramid<HostTraits<T> >(h_, im, pyrafeatures, scales_); }
};

template <typename T>
class SpatialConvolutionEngine {   // IConvolutionEngine
    pbd_handle *h_;
    size_t nfilters_;
public:
    SpatialConvolutionEngine(pbd_handle *h, size_t nfilters) : h_(h), nfilters_(nfilters) {}
    void setFilters(const std::vector<MatT<T> > &filters)
    {
        pbdbind::set_filters<HostTraits<T> >(h_, filters);
        nfilters_ = filters.size();
    }
    // responses[level][filter] = H x W
    void pdf(const std::vector<MatT<T> > &features, std::vector<std::vector<MatT<T> > > &responses)
    {
        pbdbind::pdf<HostTraits<T> >(h_, nfilters_, features, responses);
    }
};

template <typename T>
class DynamicProgram {
    pbd_handle *h_;
    int nfilters_;
public:
    DynamicProgram(pbd_handle *h, int nfilters) : h_(h), nfilters_(nfilters) {}
    // scores[level][filter]; rootv/rooti[level][component]; the back-pointers stay on the device for argmin()
    void min(const std::vector<std::vector<MatT<T> > > &scores, std::vector<std::vector<MatT<T> > > &rootv,
             std::vector<std::vector<MatT<int32_t> > > &rooti, int ncomponents)
    {
        pbdbind::dp_min<HostTraits<T> >(h_, nfilters_, ncomponents, scores, rootv, rooti);
    }
    void argmin(const std::vector<float> &scales, std::vector<Candidate> &candidates, int capacity = 1 << 16)
    {
        pbdbind::dp_argmin<HostTraits<T> >(h_, scales, candidates, capacity);
    }
};

template <typename T>
class PartsBasedDetector {
    std::string name_;
    pbd_handle *h_;
    int device_;
    PartsBasedDetector(const PartsBasedDetector &);
    PartsBasedDetector &operator=(const PartsBasedDetector &);
public:
    explicit PartsBasedDetector(int device = 0) : h_(NULL), device_(device) {}
    ~PartsBasedDetector() { pbd_destroy(h_); }
    const std::string &name() const { return name_; }
    pbd_handle *handle() const { return h_; }
    void distributeModel(Model &model)
    {   // src/PartsBasedDetector.cpp:102-127
        pbd_destroy(h_);
        h_ = NULL;
        h_ = pbdbind::create<HostTraits<T> >(model, device_, PBD_CONV_EXACT, 1, 1 << 18);
        name_ = model.name();
    }
    void detect(const Image &im, std::vector<Candidate> &candidates) { detect(im, Image(), candidates); }
    void detect(const Image &im, const Image & /*depth: ignored by the reference too, src/PartsBasedDetector.cpp:91-93*/,
                std::vector<Candidate> &candidates)
    {
        if (!h_) throw Error(PBD_ERR_STATE, "detect() before distributeModel()");
        pbdbind::detect<HostTraits<T> >(h_, im, candidates, 1 << 16);
    }
};

// A stream of single frames over K handles (K HIP streams + workspaces) of one GPU, fed round-robin -- new surface: the
// reference's callers run detect(im) one frame at a time (cells/detect.cpp:213, ros/Node.cpp:144), and one frame's kernels do
// not fill an MI355X; frames on different handles overlap at kernel granularity (one 640x480 frame: 423 -> 525 frames/s with
// four handles).  Results come back in submission order and are those of detect() on one handle.
//     FrameStream<float> fs(model, 4);
//     for (;;) { while (fs.full()) { fs.next(cands); use(cands); }   fs.submit(frame); }
//     while (fs.pending()) { fs.next(cands); use(cands); }
// Frames are 8-bit (pbd_detect_batch_submit); a submitted frame's pixels are copied before submit() returns.
template <typename T>
class FrameStream {
    std::vector<pbd_handle *> h_;
    size_t submitted_, collected_;
    int capacity_;
    FrameStream(const FrameStream &);
    FrameStream &operator=(const FrameStream &);
public:
    FrameStream(Model &model, int nhandles = 4, int device = 0, int capacity = 1 << 16) : submitted_(0), collected_(0), capacity_(capacity)
    {
        if (nhandles < 1) throw Error(PBD_ERR_INVALID, "FrameStream needs at least one handle");
        try {
            for (int i = 0; i < nhandles; ++i) h_.push_back(pbdbind::create<HostTraits<T> >(model, device, PBD_CONV_EXACT, 1, capacity));
        } catch (...) {      // a later handle failed (out of memory): the earlier ones must not leak
            for (size_t i = 0; i < h_.size(); ++i) pbd_destroy(h_[i]);
            throw;
        }
    }
    ~FrameStream() { for (size_t i = 0; i < h_.size(); ++i) pbd_destroy(h_[i]); }
    size_t pending() const { return submitted_ - collected_; }
    bool full() const { return pending() >= h_.size(); }        // the handle the next frame would go to still holds a result
    void submit(const Image &im)
    {
        if (full()) throw Error(PBD_ERR_STATE, "FrameStream::submit(): every handle holds an uncollected frame; call next() first");
        if (im.depth != 0) throw Error(PBD_ERR_UNSUPPORTED, "FrameStream takes 8-bit frames");
        pbd_handle *h = h_[submitted_ % h_.size()];
        const void *img = im.data;
        pbdbind::check<HostTraits<T> >(h, pbd_detect_batch_submit(h, 1, &img, im.rows, im.cols, im.channels, im.step));
        ++submitted_;
    }
    void next(std::vector<Candidate> &candidates)
    {   // the oldest submitted frame's candidates
        if (!pending()) throw Error(PBD_ERR_STATE, "FrameStream::next(): nothing submitted");
        pbd_handle *h = h_[collected_ % h_.size()];
        std::vector<int32_t> buf((size_t)capacity_ * pbd_candidate_stride(h) + 1);
        int n = 0;
        pbdbind::check<HostTraits<T> >(h, pbd_detect_batch_wait(h, &buf[0], capacity_, &n));
        ++collected_;
        candidates.clear();
        pbdbind::unpack_candidates<HostTraits<T> >(h, buf, n, candidates);
    }
};

// binary PGM (P5) / PPM (P6, stored RGB -> returned BGR as cv::imread does)
inline bool readPNM(const std::string &path, std::vector<uint8_t> &pix, Image &im)
{
    std::ifstream in(path.c_str(), std::ios::binary);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    if (!(in >> magic >> w >> h >> maxv) || maxv != 255 || (magic != "P5" && magic != "P6")) return false;
    in.get();
    const int cn = magic == "P6" ? 3 : 1;
    pix.resize((size_t)w * h * cn);
    in.read(reinterpret_cast<char *>(pix.data()), (std::streamsize)pix.size());
    if (!in) return false;
    if (cn == 3) for (size_t i = 0; i < pix.size(); i += 3) std::swap(pix[i], pix[i + 2]);
    im.data = pix.data(); im.rows = h; im.cols = w; im.channels = cn; im.step = (size_t)w * cn; im.depth = 0;
    return true;
}

}  // namespace pbdhost
