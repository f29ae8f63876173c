// annonet_train_host.h — the data side of the reference's training tool above the drop-in boundary:
//
//   reference (annonet_train_main.cpp)                                    here
//   crop, add_random_noise, randomly_crop_image            :58-232        same names: the crop cut ON THE HOST, every augmentation
//   (the same draws as a crop SPEC for the device)                        draw_crop_spec: rectangle / flips / brightness / noise / colour
//                                                                         offsets drawn on the host, the crop itself cut in HBM
//   ignore_classes_to_ignore, ignore_large_nonzero_regions :414-502       same names (the latter through anh_ignore_large_nonzero_regions)
//   shared_lru_cache_using_std (lru-timday, absent)        :504-510       shared_lru_cache: thread-safe, loads outside the lock
//   pull_crops loader threads, dlib::pipe<crop>            :520-553       crop_loader_pool
// dlib::rand is replaced by host_rand (std::mt19937_64 behind dlib::rand's method names): the reference seeds with time(0) + thread
// number (:524), so no draw sequence is part of its behaviour.  [UPSTREAM-UNVERIFIED] marks restated dlib routines.
#ifndef ANNONET_HIP_TRAIN_HOST_H
#define ANNONET_HIP_TRAIN_HOST_H

#include <atomic>
#include <functional>
#include <limits>
#include <list>
#include <memory>
#include <random>
#include <set>

#include "annonet_host.h"

struct host_rand {   // the dlib::rand methods annonet uses
    std::mt19937_64 gen;
    explicit host_rand(uint64_t seed = 0) : gen(seed) {}
    uint32_t get_random_32bit_number() { return (uint32_t)(gen() >> 32); }
    uint64_t get_random_64bit_number() { return gen(); }
    double get_random_double() { return (double)(gen() >> 11) * (1.0 / 9007199254740992.0); }   // [0, 1)
    double get_double_in_range(double lo, double hi) { return lo + get_random_double() * (hi - lo); }
    long long get_integer_in_range(long long lo, long long hi) { return lo + (long long)(gen() % (uint64_t)(hi - lo + 1)); }   // inclusive, as add_random_noise uses it
    double get_random_gaussian() { std::normal_distribution<double> d(0.0, 1.0); return d(gen); }
};

// ---------------------------------------------------------------------------------------- crops
struct crop {   // annonet_train_main.cpp:58-69
    NetPimpl::input_type input_image;
    NetPimpl::training_label_type label_image;
    dlib::matrix<uint16_t> temporary_unweighted_label_image;
    std::string warning, error;
    // device path: the crop is NOT cut here; the spec travels to the GPU instead (dataset_index = the full image's number in the HBM-resident dataset)
    anh_crop_spec spec{};
    bool is_spec = false;
};

struct augmentation_options {   // the reference reads these from cxxopts in randomly_crop_image (:124,178,184-185,196-197,218,227)
    double further_downscaling_factor = 1.0, class_weight = 0.5, image_weight = 0.5;
    bool allow_flip_left_right = false, allow_flip_upside_down = false, allow_random_color_offset = false;
    double multiplicative_brightness_change_probability = 0.0, multiplicative_brightness_change_sigma = 0.1, noise_level_stddev = 0.0;
};

inline void add_random_noise(NetPimpl::input_type& image, double noise_level, host_rand& rnd) {   // :73-105
    const long long rounded_noise_level = static_cast<long long>(std::round(noise_level));
    if (rounded_noise_level == 0) return;
    uint8_t* p = reinterpret_cast<uint8_t*>(&*image.begin());
    const size_t n = (size_t)image.nr() * image.nc() * NetPimpl::kInputChannels;
    for (size_t i = 0; i < n; ++i) {
        const int noise = static_cast<int>(rnd.get_integer_in_range(-rounded_noise_level, rounded_noise_level));
        p[i] = (uint8_t)std::max(0, std::min(static_cast<int>(p[i]) + noise, 255));
    }
}

// dlib::apply_random_color_offset [UPSTREAM-UNVERIFIED]: three gaussian draws through the square root of an RGB covariance matrix,
// scaled by 0.1 and rounded -> per-channel integer offsets, applied with saturation
inline void draw_color_offset(host_rand& rnd, int offset[3]) {
    static const double tform[3][3] = {{-66.379, 25.094, 6.79698}, {-68.0492, -0.302309, -13.9539}, {-68.4907, -24.0199, 7.27653}};
    const double v[3] = {rnd.get_random_gaussian(), rnd.get_random_gaussian(), rnd.get_random_gaussian()};
    for (int i = 0; i < 3; ++i) offset[i] = (int)std::round(0.1 * (tform[i][0] * v[0] + tform[i][1] * v[1] + tform[i][2] * v[2]));
}
inline void apply_color_offset(NetPimpl::input_type& image, const int offset[3]) {
    if (NetPimpl::kInputChannels != 3) return;
    uint8_t* p = reinterpret_cast<uint8_t*>(&*image.begin());
    const size_t n = (size_t)image.nr() * image.nc();
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) p[i * 3 + c] = (uint8_t)std::max(0, std::min((int)p[i * 3 + c] + offset[c], 255));
}

inline dlib::rectangle random_rect_containing_point(host_rand& rnd, const dlib::point& point, long result_width, long result_height) {   // annonet_train.h:85-105
    anh_rect r;
    NetPimpl::check(anh_random_rect_containing_point(rnd.get_random_32bit_number(), rnd.get_random_32bit_number(), point.x(), point.y(), result_width, result_height, &r));
    return dlib::rectangle(r.left, r.top, r.right, r.bottom);
}

// The random decisions of randomly_crop_image (:110-232) in the reference's order, without touching pixels: which labelled point,
// the rectangle around it, flips, brightness factor, noise level (+ a seed for the device's counter-based draws), colour offsets.
inline anh_crop_spec draw_crop_spec(int dim, const sample_type& full_sample, int dataset_index, host_rand& rnd, const augmentation_options& o) {
    const size_t class_index = rnd.get_random_32bit_number() % full_sample.labeled_points_by_class.size();
    auto i = full_sample.labeled_points_by_class.begin();
    std::advance(i, (long)class_index);
    const size_t point_index = rnd.get_random_64bit_number() % i->second.size();
    const int dim_before_downscaling = (int)std::round(dim * o.further_downscaling_factor);
    const dlib::rectangle rect = random_rect_containing_point(rnd, i->second[point_index], dim_before_downscaling, dim_before_downscaling);
    anh_crop_spec s{};
    s.image = dataset_index; s.left = rect.left(); s.top = rect.top();
    s.further_downscaling_factor = o.further_downscaling_factor;
    s.flip_left_right = o.allow_flip_left_right && rnd.get_random_double() > 0.5;
    s.flip_upside_down = o.allow_flip_upside_down && rnd.get_random_double() > 0.5;
    s.brightness_change = 1.0;
    if (o.multiplicative_brightness_change_probability > 0.0 && rnd.get_double_in_range(0, 1) < o.multiplicative_brightness_change_probability)
        s.brightness_change = std::exp(rnd.get_random_gaussian() * o.multiplicative_brightness_change_sigma);
    if (o.noise_level_stddev > 0.0) {
        s.noise_level = (int)std::min(255LL, std::llround(std::fabs(rnd.get_random_gaussian() * o.noise_level_stddev)));
        s.noise_seed = rnd.get_random_64bit_number();
    }
    if (o.allow_random_color_offset && NetPimpl::kInputChannels == 3) draw_color_offset(rnd, s.color_offset);
    return s;
}

// The crop itself on the host, from a spec (= randomly_crop_image :129-231 with the draws already made).  Noise: the reference's
// sequential draws (the device path uses counter-based draws keyed by spec.noise_seed — equally distributed, not the same numbers).
inline void cut_crop_on_host(int dim, const sample_type& full_sample, const anh_crop_spec& s, crop& crop, host_rand& rnd, const augmentation_options& o) {
    constexpr int C = NetPimpl::kInputChannels;
    const int src = (int)std::round(dim * (s.further_downscaling_factor == 0.0 ? 1.0 : s.further_downscaling_factor));
    const long H = full_sample.input_image.nr(), W = full_sample.input_image.nc();
    NetPimpl::input_type chip;
    dlib::matrix<uint16_t> chip_labels;
    chip.set_size(src, src);
    chip_labels.set_size(src, src);
    const uint8_t* img = reinterpret_cast<const uint8_t*>(&*full_sample.input_image.begin());
    uint8_t* dst = reinterpret_cast<uint8_t*>(&*chip.begin());
    // extract_image_chip at scale 1 + outpaint (annonet.h:74-120) = clamp-to-edge crop; labels "ignore" outside the image (:150-158)
    for (long r = 0; r < src; ++r) {
        const long sy = s.top + r, cy = std::min(std::max(sy, 0L), H - 1);
        for (long c = 0; c < src; ++c) {
            const long sx = s.left + c, cx = std::min(std::max(sx, 0L), W - 1);
            for (int ch = 0; ch < C; ++ch) dst[(r * src + c) * C + ch] = img[(cy * W + cx) * C + ch];
            chip_labels(r, c) = (sy == cy && sx == cx) ? full_sample.label_image(cy, cx) : (uint16_t)dlib::loss_multiclass_log_per_pixel_::label_to_ignore;
        }
    }
    if (src != dim) {   // :160-171: resize_image bilinear / nearest neighbour (annonet_host.h restates both)
        NetPimpl::input_type resized;
        resized.set_size(dim, dim);
        uint8_t* out = reinterpret_cast<uint8_t*>(&*resized.begin());
        const double scale = (src - 1) / (double)std::max(dim - 1, 1);
        for (long r = 0; r < dim; ++r) {
            const double y = r * scale;
            const long y0 = (long)std::floor(y), y1 = std::min<long>(y0 + 1, src - 1);
            const float fy = (float)(y - y0);
            for (long c = 0; c < dim; ++c) {
                const double x = c * scale;
                const long x0 = (long)std::floor(x), x1 = std::min<long>(x0 + 1, src - 1);
                const float fx = (float)(x - x0);
                for (int ch = 0; ch < C; ++ch) {
                    const float tl = dst[(y0 * src + x0) * C + ch], tr = dst[(y0 * src + x1) * C + ch], bl = dst[(y1 * src + x0) * C + ch], br = dst[(y1 * src + x1) * C + ch];
                    const float top = (1.f - fx) * tl + fx * tr, bot = (1.f - fx) * bl + fx * br;
                    out[(r * dim + c) * C + ch] = (uint8_t)(int)((1.f - fy) * top + fy * bot + 0.5f);
                }
            }
        }
        resize_label_image(chip_labels, dim, dim);
        std::swap(chip, resized);
    }
    crop.temporary_unweighted_label_image = chip_labels;
    crop.label_image.set_size(dim, dim);
    NetPimpl::check(anh_set_weights(&*crop.temporary_unweighted_label_image.begin(), dim, dim, o.class_weight, o.image_weight, reinterpret_cast<anh_wlabel*>(&*crop.label_image.begin())));   // :178
    crop.input_image = chip;
    auto flip = [&](bool lr) {
        NetPimpl::input_type fi; NetPimpl::training_label_type fl;
        fi.set_size(dim, dim); fl.set_size(dim, dim);
        for (long r = 0; r < dim; ++r)
            for (long c = 0; c < dim; ++c) {
                const long sr = lr ? r : dim - 1 - r, sc = lr ? dim - 1 - c : c;
                fi(r, c) = crop.input_image(sr, sc); fl(r, c) = crop.label_image(sr, sc);
            }
        std::swap(crop.input_image, fi); std::swap(crop.label_image, fl);
    };
    if (s.flip_left_right) flip(true);    // :186-189
    if (s.flip_upside_down) flip(false);  // :190-193
    if (s.brightness_change != 1.0) {     // :196-216, tuc::round(tuc::clamp(...)) taken as round-half-up [UPSTREAM-UNVERIFIED: tuc is absent]
        uint8_t* p = reinterpret_cast<uint8_t*>(&*crop.input_image.begin());
        for (size_t i = 0; i < (size_t)dim * dim * C; ++i) p[i] = (uint8_t)std::floor(std::min(std::max(p[i] * s.brightness_change, 0.0), 255.0) + 0.5);
    }
    if (s.noise_level > 0) add_random_noise(crop.input_image, s.noise_level, rnd);
    if (s.color_offset[0] || s.color_offset[1] || s.color_offset[2]) apply_color_offset(crop.input_image, s.color_offset);
}

inline void randomly_crop_image(int dim, const sample_type& full_sample, crop& crop, host_rand& rnd, const augmentation_options& options) {   // :110-232
    if (full_sample.labeled_points_by_class.empty()) throw std::runtime_error("no labeled points");
    crop.spec = draw_crop_spec(dim, full_sample, 0, rnd, options);
    cut_crop_on_host(dim, full_sample, crop.spec, crop, rnd, options);
}

// ---------------------------------------------------------------------------------------- per-image preparation
inline void ignore_classes_to_ignore(sample_type& sample, const std::vector<uint16_t>& classes_to_ignore) {   // :414-424
    for (const auto class_to_ignore : classes_to_ignore) {
        const auto i = sample.labeled_points_by_class.find(class_to_ignore);
        if (i != sample.labeled_points_by_class.end()) {
            for (const dlib::point& point : i->second) sample.label_image(point.y(), point.x()) = dlib::loss_multiclass_log_per_pixel_::label_to_ignore;
            sample.labeled_points_by_class.erase(class_to_ignore);
        }
    }
}

// :426-502 through the library's restatement (anh_ignore_large_nonzero_regions, tested against scipy.ndimage); the point lists are rebuilt
inline void ignore_large_nonzero_regions(sample_type& sample, double by_area, double by_width, double by_height) {
    if (sample.labeled_points_by_class.empty()) return;
    if (sample.labeled_points_by_class.size() == 1 && sample.labeled_points_by_class.begin()->first == 0) return;
    int64_t ignored = 0;
    NetPimpl::check(anh_ignore_large_nonzero_regions(&*sample.label_image.begin(), (int)sample.label_image.nr(), (int)sample.label_image.nc(), by_area, by_width, by_height,
                                                     NetPimpl::TrainingNet::GetRequiredInputDimension(), &ignored));
    if (!ignored) return;
    sample.labeled_points_by_class.clear();
    for (long r = 0; r < sample.label_image.nr(); ++r)
        for (long c = 0; c < sample.label_image.nc(); ++c) {
            const uint16_t label = sample.label_image(r, c);
            if (label != dlib::loss_multiclass_log_per_pixel_::label_to_ignore) sample.labeled_points_by_class[label].push_back(dlib::point(c, r));
        }
}

// ---------------------------------------------------------------------------------------- LRU cache of full images
// shared_lru_cache_using_std<key, value, std::unordered_map>(loader, capacity) (lru-timday, absent): operator()(key) returns the
// cached value or loads it; least recently used entries go first.  Loads run outside the lock, so loader threads decode different
// files concurrently (two threads may load the same file once each; the second result is dropped).
template <typename key_type, typename value_type, typename hash_type = std::hash<key_type>>
class shared_lru_cache {
  public:
    shared_lru_cache(std::function<value_type(const key_type&)> loader, size_t capacity) : loader_(std::move(loader)), capacity_(std::max<size_t>(capacity, 1)) {}
    value_type operator()(const key_type& key) {
        {
            std::lock_guard<std::mutex> lock(m_);
            auto it = index_.find(key);
            if (it != index_.end()) { order_.splice(order_.begin(), order_, it->second); ++hits_; return it->second->second; }
        }
        value_type v = loader_(key);
        std::lock_guard<std::mutex> lock(m_);
        auto it = index_.find(key);
        if (it != index_.end()) { order_.splice(order_.begin(), order_, it->second); return it->second->second; }
        ++misses_;
        order_.emplace_front(key, std::move(v));
        index_[key] = order_.begin();
        while (order_.size() > capacity_) { index_.erase(order_.back().first); order_.pop_back(); ++evictions_; }
        return order_.front().second;
    }
    size_t hits() const { return hits_; } size_t misses() const { return misses_; } size_t evictions() const { return evictions_; }
    size_t size() { std::lock_guard<std::mutex> lock(m_); return order_.size(); }

  private:
    std::function<value_type(const key_type&)> loader_;
    size_t capacity_;
    std::mutex m_;
    std::list<std::pair<key_type, value_type>> order_;
    std::unordered_map<key_type, typename std::list<std::pair<key_type, value_type>>::iterator, hash_type> index_;
    std::atomic<size_t> hits_{0}, misses_{0}, evictions_{0};
};

struct image_filenames_hash {   // :42-49
    size_t operator()(const image_filenames_type& f) const { return std::hash<std::string>()(f.image_filename + ", " + f.label_filename); }
};
inline bool operator==(const image_filenames_type& a, const image_filenames_type& b) { return a.image_filename == b.image_filename && a.label_filename == b.label_filename; }

#endif  // ANNONET_HIP_TRAIN_HOST_H
