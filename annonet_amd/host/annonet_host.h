// annonet_host.h — the host side of annonet's two mains above the drop-in boundary, written against
// include/NetPimpl.h so that it builds with or without dlib (ANNONET_HIP_NO_DLIB: this repository has no dlib).
//
//   reference (file:line)                                              here
//   AnnoClass, parse_anno_classes   annonet_parse_anno_classes.{h,cpp}  AnnoClass, parse_anno_classes (own small JSON reader: rapidjson is absent)
//   image_filenames_type, sample_type                 annonet.h:41-57   same names
//   find_image_files                               annonet.cpp:60-132   same name (std::filesystem walk; sorted, so runs are reproducible)
//   rgba_label_to_index_label, decode_rgba_label_image  annonet.cpp:23-58
//   resize_label_image (nearest neighbour)         annonet.cpp:134-141  same name; dlib::resize_image + interpolate_nearest_neighbor restated
//   read_sample                                    annonet.cpp:143-176  same name (PNG / PNM through image_io.h)
//   dlib::pipe                          annonet_infer_main.cpp:382-419  pipe<T>: bounded queue with enqueue / dequeue / disable
//   label_connected_blobs (8-neighbourhood, connected_if_equal, zero background)   annonet_infer_main.cpp:219-220
//   confusion matrices            annonet_infer_main.cpp:93-272,382-532  same names
// [UPSTREAM-UNVERIFIED] marks dlib routines restated from their published behaviour (dlib is not in the reference snapshot).
#ifndef ANNONET_HIP_HOST_H
#define ANNONET_HIP_HOST_H

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <filesystem>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>
#include <unordered_map>

#include "../../include/NetPimpl.h"
#include "image_io.h"

#ifdef ANNONET_HIP_NO_DLIB
namespace dlib {
struct rgb_alpha_pixel {
    unsigned char red = 0, green = 0, blue = 0, alpha = 0;
    rgb_alpha_pixel() = default;
    rgb_alpha_pixel(unsigned char r, unsigned char g, unsigned char b, unsigned char a) : red(r), green(g), blue(b), alpha(a) {}
    bool operator==(const rgb_alpha_pixel& o) const { return red == o.red && green == o.green && blue == o.blue && alpha == o.alpha; }
};
}  // namespace dlib
#endif

// ---------------------------------------------------------------------------------------- anno classes
struct AnnoClass {   // annonet_parse_anno_classes.h:22-30
    AnnoClass(uint16_t index, const dlib::rgb_alpha_pixel& rgba_label, const std::string& classlabel) : index(index), rgba_label(rgba_label), classlabel(classlabel) {}
    uint16_t index = 0;
    dlib::rgb_alpha_pixel rgba_label;
    std::string classlabel;
};
static const dlib::rgb_alpha_pixel rgba_ignore_label(0, 0, 0, 0);   // annonet_parse_anno_classes.h:32-34

namespace annonet_json {   // the subset of JSON anno_classes.json uses: objects, arrays, strings, numbers, true / false / null
struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double number = 0; bool boolean = false; std::string string;
    std::vector<Value> array;
    std::vector<std::pair<std::string, Value>> object;
    const Value* find(const std::string& key) const { for (const auto& kv : object) if (kv.first == key) return &kv.second; return nullptr; }
};
struct Parser {
    const std::string& s; size_t i = 0;
    explicit Parser(const std::string& text) : s(text) {}
    [[noreturn]] void fail() const { throw std::runtime_error("Error parsing json\n" + s); }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i; }
    Value value() {
        ws();
        if (i >= s.size()) fail();
        Value v;
        const char c = s[i];
        if (c == '{') {
            v.kind = Value::Object; ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            for (;;) {
                ws();
                Value k = value();
                if (k.kind != Value::String) fail();
                ws();
                if (i >= s.size() || s[i] != ':') fail();
                ++i;
                v.object.emplace_back(k.string, value());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return v; }
                fail();
            }
        }
        if (c == '[') {
            v.kind = Value::Array; ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            for (;;) {
                v.array.push_back(value());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return v; }
                fail();
            }
        }
        if (c == '"') {
            v.kind = Value::String; ++i;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) {
                    const char e = s[i + 1];
                    i += 2;
                    switch (e) {
                        case 'n': v.string += '\n'; break; case 't': v.string += '\t'; break; case 'r': v.string += '\r'; break;
                        case 'b': v.string += '\b'; break; case 'f': v.string += '\f'; break;
                        case 'u': {   // BMP code point -> UTF-8
                            if (i + 4 > s.size()) fail();
                            const unsigned cp = (unsigned)std::stoul(s.substr(i, 4), nullptr, 16);
                            i += 4;
                            if (cp < 0x80) v.string += (char)cp;
                            else if (cp < 0x800) { v.string += (char)(0xC0 | (cp >> 6)); v.string += (char)(0x80 | (cp & 0x3F)); }
                            else { v.string += (char)(0xE0 | (cp >> 12)); v.string += (char)(0x80 | ((cp >> 6) & 0x3F)); v.string += (char)(0x80 | (cp & 0x3F)); }
                            break;
                        }
                        default: v.string += e;
                    }
                } else v.string += s[i++];
            }
            if (i >= s.size()) fail();
            ++i;
            return v;
        }
        if (s.compare(i, 4, "true") == 0) { v.kind = Value::Bool; v.boolean = true; i += 4; return v; }
        if (s.compare(i, 5, "false") == 0) { v.kind = Value::Bool; i += 5; return v; }
        if (s.compare(i, 4, "null") == 0) { i += 4; return v; }
        size_t used = 0;
        try { v.number = std::stod(s.substr(i), &used); } catch (...) { fail(); }
        if (!used) fail();
        v.kind = Value::Number; i += used;
        return v;
    }
};
inline Value parse(const std::string& text) { Parser p(text); Value v = p.value(); p.ws(); if (p.i != text.size()) p.fail(); return v; }
}  // namespace annonet_json

inline std::vector<AnnoClass> parse_anno_classes(const std::string& json) {   // annonet_parse_anno_classes.cpp:22-83
    if (json.empty()) {
        return std::vector<AnnoClass>{AnnoClass(0, dlib::rgb_alpha_pixel(0, 255, 0, 64), "clean"), AnnoClass(1, dlib::rgb_alpha_pixel(255, 255, 0, 128), "minor defect"),
                                      AnnoClass(2, dlib::rgb_alpha_pixel(255, 0, 0, 128), "major defect")};
    }
    const annonet_json::Value doc = annonet_json::parse(json);
    if (doc.kind != annonet_json::Value::Object) throw std::runtime_error("Unexpected anno classes json content - the document should be an object");
    const annonet_json::Value* list = doc.find("anno_classes");
    if (!list || list->kind != annonet_json::Value::Array) throw std::runtime_error("Unexpected anno classes json content - there should be an anno_classes array");
    std::vector<AnnoClass> anno_classes;
    for (size_t i = 0; i < list->array.size(); ++i) {
        const annonet_json::Value& c = list->array[i];
        const annonet_json::Value* name = c.find("name");
        const annonet_json::Value* color = c.find("color");
        if (!name) throw std::runtime_error("Unexpected anno classes json content - no name found");
        if (!color) throw std::runtime_error("Unexpected anno classes json content - no color found");
        const annonet_json::Value *r = color->find("red"), *g = color->find("green"), *b = color->find("blue"), *a = color->find("alpha");
        if (!r || !g || !b || !a) throw std::runtime_error("Unexpected anno classes json content - color should have all components (red, green, blue, alpha)");
        const dlib::rgb_alpha_pixel rgba((unsigned char)(int)r->number, (unsigned char)(int)g->number, (unsigned char)(int)b->number, (unsigned char)(int)a->number);
        if (rgba == rgba_ignore_label) throw std::runtime_error("Unexpected anno classes json content - rgba (0, 0, 0, 0) is reserved for pixels to be ignored");
        anno_classes.push_back(AnnoClass((uint16_t)i, rgba, name->string));
    }
    return anno_classes;
}

// ---------------------------------------------------------------------------------------- samples
struct image_filenames_type { std::string image_filename, label_filename; };   // annonet.h:41-45

struct sample_type {   // annonet.h:49-57
    int original_width = 0, original_height = 0;
    image_filenames_type image_filenames;
    NetPimpl::input_type input_image;
    dlib::matrix<uint16_t> label_image;
    std::unordered_map<uint16_t, std::deque<dlib::point>> labeled_points_by_class;
    std::string error;
};

inline uint16_t rgba_label_to_index_label(const dlib::rgb_alpha_pixel& rgba_label, const std::vector<AnnoClass>& anno_classes) {   // annonet.cpp:23-39
    if (rgba_label == rgba_ignore_label) return dlib::loss_multiclass_log_per_pixel_::label_to_ignore;
    for (const AnnoClass& anno_class : anno_classes) if (anno_class.rgba_label == rgba_label) return anno_class.index;
    std::ostringstream error;
    error << "Unknown class: r = " << (int)rgba_label.red << ", g = " << (int)rgba_label.green << ", b = " << (int)rgba_label.blue << ", alpha = " << (int)rgba_label.alpha;
    throw std::runtime_error(error.str());
}

inline void decode_rgba_label_image(const dlib::matrix<dlib::rgb_alpha_pixel>& rgba_label_image, sample_type& ground_truth_sample, const std::vector<AnnoClass>& anno_classes) {   // annonet.cpp:41-58
    const long nr = rgba_label_image.nr(), nc = rgba_label_image.nc();
    ground_truth_sample.label_image.set_size(nr, nc);
    ground_truth_sample.labeled_points_by_class.clear();
    for (long r = 0; r < nr; ++r)
        for (long c = 0; c < nc; ++c) {
            const uint16_t label = rgba_label_to_index_label(rgba_label_image(r, c), anno_classes);
            if (label != dlib::loss_multiclass_log_per_pixel_::label_to_ignore) ground_truth_sample.labeled_points_by_class[label].push_back(dlib::point(c, r));
            ground_truth_sample.label_image(r, c) = label;
        }
}

inline std::vector<image_filenames_type> find_image_files(const std::string& anno_data_folder, bool require_ground_truth) {   // annonet.cpp:60-132
    namespace fs = std::filesystem;
    using annonet_io::ends_with;
    std::cout << std::endl << "Scanning...";
    std::vector<std::string> files;
    for (const auto& entry : fs::recursive_directory_iterator(anno_data_folder)) {
        if (!entry.is_regular_file()) continue;
        const std::string name = entry.path().string();
        if (ends_with(name, "_mask.png") || ends_with(name, "_result.png")) continue;
        if (ends_with(name, ".jpeg") || ends_with(name, ".jpg") || ends_with(name, ".JPG") || ends_with(name, ".png") || ends_with(name, ".PNG") ||
            ends_with(name, ".ppm") || ends_with(name, ".pgm"))   // (+ binary PNM: this build reads it without an image library)
            files.push_back(name);
    }
    std::sort(files.begin(), files.end());
    std::cout << " found " << files.size() << " candidates" << std::endl;
    std::vector<image_filenames_type> results;
    size_t added = 0, ignored = 0;
    for (size_t i = 0, total = files.size(); i < total; ++i) {
        image_filenames_type image_filenames;
        image_filenames.image_filename = files[i];
        const std::string label_filename = files[i] + "_mask.png";
        const bool label_file_exists = !!std::ifstream(label_filename, std::ios::binary);
        if (label_file_exists) image_filenames.label_filename = label_filename;
        if (label_file_exists || !require_ground_truth) { results.push_back(image_filenames); ++added; }
        else ++ignored;
        if (i == 0 || i == total - 1)
            std::cout << "\rScanned " << std::fixed << std::setprecision(2) << ((i + 1) * 100.0) / total << " % of " << total << " files: " << added << " added, " << ignored << " ignored";
    }
    std::cout << std::endl;
    return results;
}

// dlib::resize_image(in, out, interpolate_nearest_neighbor()) [UPSTREAM-UNVERIFIED]: the output pixel (r, c) samples the input at
// (c * (in_nc-1)/max(out_nc-1,1), r * (in_nr-1)/max(out_nr-1,1)), rounded to the nearest pixel (floor(v + 0.5)).
template <typename image_type>
void resize_label_image(image_type& label_image, int target_width, int target_height) {   // annonet.cpp:134-141
    image_type temp;
    temp.set_size(target_height, target_width);
    const long in_nr = label_image.nr(), in_nc = label_image.nc();
    const double x_scale = (in_nc - 1) / (double)std::max<long>(target_width - 1, 1), y_scale = (in_nr - 1) / (double)std::max<long>(target_height - 1, 1);
    for (long r = 0; r < target_height; ++r) {
        const long sy = (long)std::floor(r * y_scale + 0.5);
        for (long c = 0; c < target_width; ++c) {
            const long sx = (long)std::floor(c * x_scale + 0.5);
            if (sy >= 0 && sy < in_nr && sx >= 0 && sx < in_nc) temp(r, c) = label_image(sy, sx);
        }
    }
    std::swap(label_image, temp);
}

// dlib::resize_image(size_scale, img) with the default bilinear interpolation [UPSTREAM-UNVERIFIED]: new size = round(scale * old),
// same corner-aligned sampling grid as above, channels interpolated in float and rounded half up.  Scale 1 is the identity.
inline void resize_image_bilinear(double size_scale, NetPimpl::input_type& img) {
    if (size_scale == 1.0) return;
    const long in_nr = img.nr(), in_nc = img.nc();
    const long out_nr = (long)std::round(size_scale * in_nr), out_nc = (long)std::round(size_scale * in_nc);
    if (out_nr < 1 || out_nc < 1) throw std::runtime_error("image is too small for this downscaling factor");
    NetPimpl::input_type out;
    out.set_size(out_nr, out_nc);
    constexpr int C = NetPimpl::kInputChannels;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(&*img.begin());
    uint8_t* dst = reinterpret_cast<uint8_t*>(&*out.begin());
    const double x_scale = (in_nc - 1) / (double)std::max<long>(out_nc - 1, 1), y_scale = (in_nr - 1) / (double)std::max<long>(out_nr - 1, 1);
    for (long r = 0; r < out_nr; ++r) {
        const double y = r * y_scale;
        const long top = (long)std::floor(y), bottom = std::min(top + 1, in_nr - 1);
        const float fy = (float)(y - top);
        for (long c = 0; c < out_nc; ++c) {
            const double x = c * x_scale;
            const long left = (long)std::floor(x), right = std::min(left + 1, in_nc - 1);
            const float fx = (float)(x - left);
            for (int ch = 0; ch < C; ++ch) {
                const float tl = src[(top * in_nc + left) * C + ch], tr = src[(top * in_nc + right) * C + ch];
                const float bl = src[(bottom * in_nc + left) * C + ch], br = src[(bottom * in_nc + right) * C + ch];
                const float v = (1 - fy) * ((1 - fx) * tl + fx * tr) + fy * ((1 - fx) * bl + fx * br);
                dst[(r * out_nc + c) * C + ch] = (uint8_t)(v + 0.5f);
            }
        }
    }
    std::swap(img, out);
}

// dlib::load_image into the net's input type [UPSTREAM-UNVERIFIED conversions: gray -> r=g=b, alpha dropped; rgb -> gray = (r+g+b)/3]
inline void load_input_image(NetPimpl::input_type& image, const std::string& filename) {
    const annonet_io::Raster r = annonet_io::load_raster(filename);
    image.set_size(r.height, r.width);
    uint8_t* dst = reinterpret_cast<uint8_t*>(&*image.begin());
    const size_t n = (size_t)r.width * r.height;
    const int colour = r.channels >= 3 ? 3 : 1;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = r.data.data() + i * r.channels;
        if (NetPimpl::kInputChannels == 3) for (int c = 0; c < 3; ++c) dst[i * 3 + c] = colour == 3 ? p[c] : p[0];
        else dst[i] = colour == 3 ? (uint8_t)(((unsigned)p[0] + p[1] + p[2]) / 3) : p[0];
    }
}
inline void load_rgba_image(dlib::matrix<dlib::rgb_alpha_pixel>& image, const std::string& filename) {
    const annonet_io::Raster r = annonet_io::load_raster(filename);
    image.set_size(r.height, r.width);
    const size_t n = (size_t)r.width * r.height;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* p = r.data.data() + i * r.channels;
        dlib::rgb_alpha_pixel& q = *(image.begin() + i);
        if (r.channels >= 3) { q.red = p[0]; q.green = p[1]; q.blue = p[2]; q.alpha = r.channels == 4 ? p[3] : 255; }
        else { q.red = q.green = q.blue = p[0]; q.alpha = r.channels == 2 ? p[1] : 255; }
    }
}
inline void save_png(const dlib::matrix<dlib::rgb_alpha_pixel>& image, const std::string& filename) {   // annonet_infer_main.cpp:413
    annonet_io::Raster r;
    r.width = (int)image.nc(); r.height = (int)image.nr(); r.channels = 4;
    r.data.resize((size_t)r.width * r.height * 4);
    for (size_t i = 0; i < (size_t)r.width * r.height; ++i) {
        const dlib::rgb_alpha_pixel& q = *(image.begin() + i);
        r.data[i * 4] = q.red; r.data[i * 4 + 1] = q.green; r.data[i * 4 + 2] = q.blue; r.data[i * 4 + 3] = q.alpha;
    }
    annonet_io::save_raster_png(r, filename);
}

inline sample_type read_sample(const image_filenames_type& image_filenames, const std::vector<AnnoClass>& anno_classes, bool require_ground_truth, double downscaling_factor) {   // annonet.cpp:143-176
    sample_type sample;
    sample.image_filenames = image_filenames;
    try {
        dlib::matrix<dlib::rgb_alpha_pixel> rgba_label_image;
        load_input_image(sample.input_image, image_filenames.image_filename);
        sample.original_width = (int)sample.input_image.nc();
        sample.original_height = (int)sample.input_image.nr();
        resize_image_bilinear(1.0 / downscaling_factor, sample.input_image);
        if (!image_filenames.label_filename.empty()) {
            load_rgba_image(rgba_label_image, image_filenames.label_filename);
            if (rgba_label_image.nr() != sample.original_height || rgba_label_image.nc() != sample.original_width) sample.error = "Label image size mismatch";
            else {
                resize_label_image(rgba_label_image, (int)sample.input_image.nc(), (int)sample.input_image.nr());
                decode_rgba_label_image(rgba_label_image, sample, anno_classes);
            }
        } else if (require_ground_truth) sample.error = "No ground truth available";
    } catch (std::exception& e) { sample.error = e.what(); }
    return sample;
}

// ---------------------------------------------------------------------------------------- dlib::pipe
namespace anh_host {   // (::pipe is unistd's)
template <typename T>
class pipe {   // bounded multi-producer / multi-consumer queue: enqueue blocks when full, dequeue when empty; disable() releases everyone
  public:
    explicit pipe(size_t max_size) : max_(std::max<size_t>(max_size, 1)) {}
    bool enqueue(T item) {
        std::unique_lock<std::mutex> lock(m_);
        not_full_.wait(lock, [&] { return q_.size() < max_ || !enabled_; });
        if (!enabled_) return false;
        q_.push_back(std::move(item));
        not_empty_.notify_one();
        return true;
    }
    bool dequeue(T& item) {
        std::unique_lock<std::mutex> lock(m_);
        not_empty_.wait(lock, [&] { return !q_.empty() || !enabled_; });
        if (q_.empty()) return false;
        item = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void disable() { std::lock_guard<std::mutex> lock(m_); enabled_ = false; not_empty_.notify_all(); not_full_.notify_all(); }
    size_t size() { std::lock_guard<std::mutex> lock(m_); return q_.size(); }

  private:
    std::mutex m_;
    std::condition_variable not_empty_, not_full_;
    std::deque<T> q_;
    size_t max_;
    bool enabled_ = true;
};
}  // namespace anh_host

// ---------------------------------------------------------------------------------------- connected blobs
// dlib::label_connected_blobs(img, zero_pixels_are_background(), neighbors_8(), connected_if_equal(), out) [UPSTREAM-UNVERIFIED numbering]:
// background (value 0) pixels get 0; every 8-connected region of EQUAL non-zero value gets the next number, in raster order of
// its first pixel.  Returns the number of labels including the background one.
inline unsigned long label_connected_blobs(const dlib::matrix<uint16_t>& img, dlib::matrix<int>& blobs) {
    const long nr = img.nr(), nc = img.nc();
    blobs.set_size(nr, nc);
    std::fill(blobs.begin(), blobs.end(), 0);
    unsigned long next = 1;
    std::vector<std::pair<long, long>> stack;
    for (long r = 0; r < nr; ++r)
        for (long c = 0; c < nc; ++c) {
            if (img(r, c) == 0 || blobs(r, c) != 0) continue;
            const uint16_t v = img(r, c);
            blobs(r, c) = (int)next;
            stack.push_back({r, c});
            while (!stack.empty()) {
                const auto [y, x] = stack.back();
                stack.pop_back();
                for (long dy = -1; dy <= 1; ++dy)
                    for (long dx = -1; dx <= 1; ++dx) {
                        const long yy = y + dy, xx = x + dx;
                        if (yy < 0 || yy >= nr || xx < 0 || xx >= nc || blobs(yy, xx) != 0 || img(yy, xx) != v) continue;
                        blobs(yy, xx) = (int)next;
                        stack.push_back({yy, xx});
                    }
            }
            ++next;
        }
    return next;
}

// ---------------------------------------------------------------------------------------- confusion matrices
// Counts of (ground truth class, predicted class) pairs and the tool's printout of them (format of annonet_infer_main.cpp:101-194:
// a "predicted" caption, a class header with a "recall" column, one row per ground-truth class with the word "truth" on the
// middle row, a "precision" row, an "accuracy" line).  The layout is computed as strings first and padded afterwards.
class ConfusionMatrix {
  public:
    explicit ConfusionMatrix(size_t class_count = 0) : k_(class_count), cells_(class_count * class_count, 0) {}
    size_t classes() const { return k_; }
    void add(size_t truth, size_t predicted, size_t n = 1) { if (truth < k_ && predicted < k_) cells_[truth * k_ + predicted] += n; }   // (a 65535 label = an all-NaN pixel: not a class)
    size_t at(size_t truth, size_t predicted) const { return cells_[truth * k_ + predicted]; }
    size_t total() const { size_t t = 0; for (size_t v : cells_) t += v; return t; }

    void print(std::ostream& out, const std::vector<AnnoClass>& anno_classes) const {
        auto pad = [](const std::string& text, size_t width) { return text.size() >= width ? text : std::string(width - text.size(), ' ') + text; };
        size_t largest = 0, row_sum_all = 0, diagonal = 0;
        std::vector<size_t> column_sum(k_, 0), row_sum(k_, 0);
        for (size_t t = 0; t < k_; ++t)
            for (size_t p = 0; p < k_; ++p) {
                const size_t v = at(t, p);
                largest = std::max(largest, v); column_sum[p] += v; row_sum[t] += v; row_sum_all += v;
                if (t == p) diagonal += v;
            }
        const std::string truth_word = "truth", predicted_word = "predicted", recall_word = "recall", precision_word = "precision", full = "100 %";
        const size_t cell_w = std::max(full.size() + 1, std::to_string(largest).size() + 2);
        const size_t class_w = std::to_string(k_ - 1).size() + 3;
        const size_t lead_w = truth_word.size() + class_w, recall_w = recall_word.size() + 4;
        out << pad(predicted_word, lead_w + cell_w * k_ / 2 + predicted_word.size() / 2) << std::endl;
        std::string header(lead_w, ' ');
        for (const AnnoClass& c : anno_classes) header += pad(std::to_string(c.index), cell_w);
        out << header << pad(recall_word, recall_w) << std::endl;
        for (size_t t = 0; t < k_; ++t) {
            std::string line = t == (k_ - 1) / 2 ? truth_word : std::string(truth_word.size(), ' ');
            line += pad(std::to_string(t), class_w);
            for (size_t p = 0; p < k_; ++p) line += pad(std::to_string(at(t, p)), cell_w);
            std::ostringstream recall;   // setw applies to the number only; the " %" follows it (as the reference's stream does)
            recall << std::setw((int)recall_w) << std::fixed << std::setprecision(2) << at(t, t) * 100.0 / row_sum[t] << " %";
            out << line << recall.str() << std::endl;
        }
        const int precision_digits = (int)std::min<size_t>(2, cell_w - full.size() - 1);
        std::string precision_line = pad(precision_word, lead_w) + "  ";
        for (size_t p = 0; p < k_; ++p) {
            std::ostringstream cell;
            cell << std::right << std::setw((int)cell_w - 2) << std::fixed << std::setprecision(precision_digits);
            if (column_sum[p] > 0) cell << at(p, p) * 100.0 / column_sum[p] << " %";
            else cell << "-" << "  ";
            precision_line += cell.str();
        }
        out << precision_line << std::endl;
        std::ostringstream accuracy;
        accuracy << pad("accuracy", lead_w + k_ * cell_w) << std::setw((int)recall_w) << std::fixed << std::setprecision(2) << diagonal * 100.0 / row_sum_all << " %";
        out << accuracy.str() << std::endl;
    }

  private:
    size_t k_;
    std::vector<size_t> cells_;
};

// Region-level scoring (annonet_infer_main.cpp:202-272): every connected region of the ground truth AND every connected region of
// the result casts ONE vote: (the class most labelled pixels of the region carry in the ground truth, the class most of them
// carry in the result).  Where the ground truth of a region is predominantly a defect class, background predictions inside it
// are disregarded unless the result is background ONLY ("do not ignore any detection, even if small in area").  Regions without
// labelled pixels do not vote.  Ties go to the smaller class index (the reference leaves them to unordered_map order).
class RegionScorer {
  public:
    void score(ConfusionMatrix& matrix, const sample_type& truth, const dlib::matrix<uint16_t>& result) {
        if (truth.labeled_points_by_class.empty()) return;
        if (truth.label_image.nr() != result.nr() || truth.label_image.nc() != result.nc()) throw std::runtime_error("ground truth and result sizes differ");
        vote(matrix, truth, result, label_connected_blobs(truth.label_image, blobs_), blobs_);
        vote(matrix, truth, result, label_connected_blobs(result, blobs_), blobs_);
    }

  private:
    static constexpr uint16_t kNone = dlib::loss_multiclass_log_per_pixel_::label_to_ignore;
    struct Tally {   // per region: votes per class, dense for small class counts
        std::vector<size_t> by_class;
        void add(size_t cls) { if (cls >= by_class.size()) by_class.resize(cls + 1, 0); ++by_class[cls]; }
        uint16_t winner() const { uint16_t best = kNone; size_t most = 0; for (size_t c = 0; c < by_class.size(); ++c) if (by_class[c] > most) { most = by_class[c]; best = (uint16_t)c; } return best; }
        size_t distinct() const { size_t n = 0; for (size_t v : by_class) n += v > 0; return n; }
    };
    void vote(ConfusionMatrix& matrix, const sample_type& truth, const dlib::matrix<uint16_t>& result, unsigned long regions, const dlib::matrix<int>& region_of) {
        std::vector<Tally> in_truth(regions), in_result(regions);
        for (const auto& cls_points : truth.labeled_points_by_class)
            for (const dlib::point& p : cls_points.second) {
                const int region = region_of(p.y(), p.x());
                in_truth[region].add(cls_points.first);
                const uint16_t predicted = result(p.y(), p.x());
                if (predicted != kNone) in_result[region].add(predicted);
            }
        for (unsigned long r = 0; r < regions; ++r) {
            const uint16_t truth_class = in_truth[r].winner();
            if (truth_class == kNone) continue;
            Tally& predicted = in_result[r];
            const bool background_only = predicted.distinct() == 1 && !predicted.by_class.empty() && predicted.by_class[0] > 0;
            if (truth_class != 0 && !background_only && !predicted.by_class.empty()) predicted.by_class[0] = 0;
            matrix.add(truth_class, predicted.winner());
        }
    }
    dlib::matrix<int> blobs_;
};

#endif  // ANNONET_HIP_HOST_H
