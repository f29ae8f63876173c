#include "hostlogic.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "common.h"
#include "spec.h"

namespace anh {

// ---------------------------------------------------------------------------------------------------
// Tiler — stand-in for tiling::get_tiles (tiling/tiling.{h,cpp} absent; contract from annonet_infer.cpp:42-164):
// full_rects cover the image with >= overlap shared pixels between neighbours, unique_rect is the part of a
// full_rect no other tile covers, and a single tile has unique == full.  Tiles come out row-major.
// ---------------------------------------------------------------------------------------------------
namespace {
struct Span { long first, last; };

std::vector<Span> cut_axis(long extent, long max_len, long overlap) {
    std::vector<Span> cuts;
    if (extent <= 0) return cuts;
    if (extent <= max_len) { cuts.push_back({0, extent - 1}); return cuts; }
    ANH_REQUIRE(overlap >= 0, "overlap must not be negative");
    ANH_REQUIRE(max_len > 2 * overlap, "max tile size must exceed twice the overlap");
    const long stride_max = max_len - overlap;
    const long pieces = (extent - overlap + stride_max - 1) / stride_max;
    const long len = (extent + (pieces - 1) * overlap + pieces - 1) / pieces;
    for (long i = 0; i < pieces; ++i) {
        const long first = i * (extent - len) / (pieces - 1);
        cuts.push_back({first, first + len - 1});
    }
    for (long i = 0; i + 2 < pieces; ++i)
        ANH_REQUIRE(cuts[i + 2].first > cuts[i].last + 1, "max tile size is too small for this overlap");
    return cuts;
}

Span exclusive_part(const std::vector<Span>& cuts, size_t i, long extent) {
    return {i == 0 ? 0 : cuts[i - 1].last + 1, i + 1 == cuts.size() ? extent - 1 : cuts[i + 1].first - 1};
}
}  // namespace

std::vector<anh_tile> make_tiles(int width, int height, const anh_tiling_params& p) {
    ANH_REQUIRE(width >= 0 && height >= 0, "negative image size");
    ANH_REQUIRE(p.max_tile_width >= 1 && p.max_tile_height >= 1, "max tile size must be positive");
    const std::vector<Span> cols = cut_axis(width, p.max_tile_width, p.overlap_x);
    const std::vector<Span> rows = cut_axis(height, p.max_tile_height, p.overlap_y);
    std::vector<anh_tile> tiles;
    tiles.reserve(cols.size() * rows.size());
    for (size_t r = 0; r < rows.size(); ++r) {
        const Span uy = exclusive_part(rows, r, height);
        for (size_t c = 0; c < cols.size(); ++c) {
            const Span ux = exclusive_part(cols, c, width);
            anh_tile t;
            t.full_rect = {cols[c].first, rows[r].first, cols[c].last, rows[r].last};
            t.unique_rect = {ux.first, uy.first, ux.last, uy.last};
            tiles.push_back(t);
        }
    }
    return tiles;
}

// annonet_infer.cpp:46-66: centre = tl + size/2 (integer division); side = GetRecommendedInputDimension(full side);
// left = cx - side/2.
TileWindow tile_window(const anh_tile& t, int levels) {
    const long fw = t.full_rect.right - t.full_rect.left + 1, fh = t.full_rect.bottom - t.full_rect.top + 1;
    const long cx = t.full_rect.left + fw / 2, cy = t.full_rect.top + fh / 2;
    TileWindow w;
    w.width = Spec::recommended_input_dim(levels, (int)fw);
    w.height = Spec::recommended_input_dim(levels, (int)fh);
    w.left = (int)(cx - w.width / 2);
    w.top = (int)(cy - w.height / 2);
    return w;
}

// ---------------------------------------------------------------------------------------------------
// set_weights (annonet_train.h:20-83).  Per-crop histogram -> w_l = (avg/count_l)^class_weight, renormalised so that
// the weights sum to total * (nr*nc/total)^image_weight.  "avg" divides by the histogram's *allocated* length
// (index*2+16 growth, annonet_train.h:29-34); it cancels in the normalisation but is kept for fidelity.
// ---------------------------------------------------------------------------------------------------
// the arithmetic of set_weights on a finished histogram (its LENGTH is part of the arithmetic: see above)
static std::vector<double> weights_from_histogram(const std::vector<size_t>& histogram, long long pixels, double class_weight, double image_weight) {
    size_t labelled = 0;
    for (size_t c : histogram) labelled += c;
    std::vector<double> weight_of(histogram.size(), 0.0);
    if (labelled > 0) {
        const double mean_count = labelled / (double)histogram.size();
        double raw_total = 0.0;
        for (size_t l = 0; l < histogram.size(); ++l) {
            if (histogram[l] == 0) continue;
            weight_of[l] = std::pow(mean_count / histogram[l], class_weight);
            raw_total += histogram[l] * weight_of[l];
        }
        const double wanted_total = labelled * std::pow(pixels / (double)labelled, image_weight);
        for (double& w : weight_of) w *= wanted_total / raw_total;
    }
    return weight_of;
}

void set_weights(const uint16_t* labels, int nr, int nc, double class_weight, double image_weight, anh_wlabel* out) {
    ANH_REQUIRE(nr >= 0 && nc >= 0, "negative label image size");
    const size_t n = (size_t)nr * nc;
    std::vector<size_t> histogram;
    for (size_t i = 0; i < n; ++i) {
        const uint16_t l = labels[i];
        if (l == ANH_LABEL_IGNORE) continue;
        if (l >= histogram.size()) histogram.resize((size_t)l * 2 + 16);
        ++histogram[l];
    }
    const std::vector<double> weight_of = weights_from_histogram(histogram, (long long)nr * nc, class_weight, image_weight);
    for (size_t i = 0; i < n; ++i) {
        out[i].label = labels[i];
        out[i].weight = labels[i] == ANH_LABEL_IGNORE ? 0.0f : (float)weight_of[labels[i]];
    }
}

// set_weights' per-label weight table from a histogram gathered elsewhere (the device crop path): `counts[l]` pixels carry
// label l < n_labels and its first row-major position is `first_position[l]` (unused where the count is 0).  The
// histogram's allocated length grows to 2 l + 16 whenever a label l >= the current length is met (annonet_train.h:29-34);
// only FIRST occurrences can trigger that, so replaying them in position order reproduces the length exactly.
void set_weights_table(const unsigned* counts, const unsigned* first_position, int n_labels, long long pixels, double class_weight,
                       double image_weight, float* table) {
    std::vector<std::pair<unsigned, int>> firsts;
    for (int l = 0; l < n_labels; ++l) if (counts[l]) firsts.emplace_back(first_position[l], l);
    std::sort(firsts.begin(), firsts.end());
    size_t length = 0;
    for (const auto& f : firsts) if ((size_t)f.second >= length) length = (size_t)f.second * 2 + 16;
    std::vector<size_t> histogram(length, 0);
    for (int l = 0; l < n_labels; ++l) if (counts[l]) histogram[l] = counts[l];
    const std::vector<double> weight_of = weights_from_histogram(histogram, pixels, class_weight, image_weight);
    for (int l = 0; l < n_labels; ++l) table[l] = (size_t)l < weight_of.size() ? (float)weight_of[l] : 0.f;
}

// random_rect_containing_point (annonet_train.h:85-105)
anh_rect random_rect_containing_point(uint32_t draw_x, uint32_t draw_y, long px, long py, long w, long h) {
    ANH_REQUIRE(w >= 1 && h >= 1, "rect size must be positive");
    const long cx_lo = px - (w - 1) / 2, cx_hi = px + w / 2;
    const long cy_lo = py - (h - 1) / 2, cy_hi = py + h / 2;
    const long cx = cx_lo + (long)(draw_x % (uint64_t)(cx_hi - cx_lo + 1));
    const long cy = cy_lo + (long)(draw_y % (uint64_t)(cy_hi - cy_lo + 1));
    anh_rect r;
    r.left = cx - w / 2; r.top = cy - h / 2;  // dlib::centered_rect
    r.right = r.left + w - 1; r.bottom = r.top + h - 1;
    ANH_REQUIRE(px >= r.left && px <= r.right && py >= r.top && py <= r.bottom, "rect does not contain the point");
    return r;
}

// outpaint (annonet.h:74-120): every pixel outside `inside` takes the value of the nearest pixel of `inside`.
void outpaint(uint8_t* image, int nr, int nc, int channels, anh_rect in) {
    in.left = std::max(in.left, 0L); in.top = std::max(in.top, 0L);
    in.right = std::min<long>(in.right, nc - 1); in.bottom = std::min<long>(in.bottom, nr - 1);
    if (in.left > in.right || in.top > in.bottom) return;
    for (long r = 0; r < nr; ++r) {
        const long sr = std::min(std::max(r, in.top), in.bottom);
        for (long c = 0; c < nc; ++c) {
            const long sc = std::min(std::max(c, in.left), in.right);
            if (sr == r && sc == c) continue;
            std::memmove(image + ((size_t)r * nc + c) * channels, image + ((size_t)sr * nc + sc) * channels, channels);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// annonet.dnn (annonet_train_main.cpp:557-565 writes, annonet_infer_main.cpp:340-351 reads):
//     serialize("annonet.dnn") << anno_classes_json << downscaling_factor << serialized_runtime_net
// i.e. dlib's stream serialization of (std::string, double, std::string), nothing else in the file.
// dlib (github.com/reunanen/dlib, un-vendored submodule, no pinned SHA in the snapshot) is ABSENT: the framing below
// restates dlib/serialize.h + dlib/float_details.h as published  [UPSTREAM-UNVERIFIED]:
//   integer   : one control byte = number of value bytes that follow (1..8), bit 7 set for a negative value, then the
//               magnitude, least-significant byte first, without leading zero bytes (zero = 01 00)
//   string    : its length as an unsigned integer, then the raw bytes
//   double    : (int64 mantissa, int16 exponent) as two integers with value = mantissa * 2^exponent; from frexp with
//               53 digits, trailing zero BYTES of the mantissa shifted into the exponent (1.0 = mantissa 16, exponent -4);
//               +inf / -inf / nan = mantissa 0 and exponent 32000 / 32001 / 32002
// The third string is RuntimeNet::Serialize's opaque blob (anh_runtime_serialize here; dlib's net format there).
// ---------------------------------------------------------------------------------------------------------------
namespace {
void put_int(std::string& out, long long v) {
    unsigned char buf[9];
    const bool neg = v < 0;
    unsigned long long m = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
    int n = 0;
    do { buf[1 + n++] = (unsigned char)(m & 0xFF); m >>= 8; } while (m != 0 && n < 8);
    buf[0] = (unsigned char)(n | (neg ? 0x80 : 0));
    out.append(reinterpret_cast<const char*>(buf), (size_t)n + 1);
}
long long get_int(const std::string& in, size_t& pos, int max_bytes, const char* what) {
    if (pos >= in.size()) fail(ANH_ERR_IO, std::string("annonet.dnn truncated while reading ") + what);
    const unsigned char ctl = (unsigned char)in[pos++];
    const int n = ctl & 0x0F;
    if ((ctl & 0x70) != 0 || n < 1 || n > max_bytes) fail(ANH_ERR_IO, std::string("annonet.dnn: bad integer header in ") + what);
    if (pos + (size_t)n > in.size()) fail(ANH_ERR_IO, std::string("annonet.dnn truncated while reading ") + what);
    unsigned long long m = 0;
    for (int i = n - 1; i >= 0; --i) m = (m << 8) | (unsigned char)in[pos + (size_t)i];
    pos += (size_t)n;
    return (ctl & 0x80) ? (long long)(0ull - m) : (long long)m;
}
void put_string(std::string& out, const std::string& s) { put_int(out, (long long)s.size()); out += s; }
std::string get_string(const std::string& in, size_t& pos, const char* what) {
    const long long n = get_int(in, pos, 8, what);
    if (n < 0 || (unsigned long long)n > in.size() - pos) fail(ANH_ERR_IO, std::string("annonet.dnn truncated while reading ") + what);
    std::string s = in.substr(pos, (size_t)n);
    pos += (size_t)n;
    return s;
}
void put_double(std::string& out, double v) {
    long long mantissa = 0; int exponent = 0;
    if (v == INFINITY) exponent = 32000;
    else if (v == -INFINITY) exponent = 32001;
    else if (v != v) exponent = 32002;
    else {
        int e;
        mantissa = (long long)(std::frexp(v, &e) * 9007199254740992.0);   // 2^53
        exponent = e - 53;
        for (int i = 0; i < 8 && (mantissa & 0xFF) == 0; ++i) { mantissa >>= 8; exponent += 8; }
    }
    put_int(out, mantissa); put_int(out, exponent);
}
double get_double(const std::string& in, size_t& pos, const char* what) {
    const long long mantissa = get_int(in, pos, 8, what);
    const long long exponent = get_int(in, pos, 2, what);
    if (exponent == 32000) return INFINITY;
    if (exponent == 32001) return -INFINITY;
    if (exponent == 32002) return NAN;
    return std::ldexp((double)mantissa, (int)exponent);
}
}  // namespace

std::string dnn_envelope_pack(const std::string& classes_json, double downscaling_factor, const std::string& net_blob) {
    std::string out;
    out.reserve(classes_json.size() + net_blob.size() + 40);
    put_string(out, classes_json); put_double(out, downscaling_factor); put_string(out, net_blob);
    return out;
}
void dnn_envelope_unpack(const std::string& file, std::string& classes_json, double& downscaling_factor, std::string& net_blob) {
    size_t pos = 0;
    classes_json = get_string(file, pos, "the class list");
    downscaling_factor = get_double(file, pos, "the downscaling factor");
    net_blob = get_string(file, pos, "the serialized net");
}

// ignore_large_nonzero_regions (annonet_train_main.cpp:434-502): 8-connected blobs of EQUAL label, background = label 0
// or the ignore label (annonet.h:26-37); a blob whose pixel count exceeds by_area * rf^2, or whose bounding box is wider
// than by_width * rf or taller than by_height * rf (rf = receptive-field side, GetRequiredInputDimension()), is relabelled
// to the ignore label.  The reference's early returns (no annotations / background only / thresholds beyond the image,
// :435-448) change nothing a blob test would not also leave alone, so they are not restated.  Returns the pixels ignored.
int64_t ignore_large_nonzero_regions(uint16_t* labels, int nr, int nc, double by_area, double by_width, double by_height, int receptive_field_side) {
    const double rf = (double)receptive_field_side;
    const double max_points = by_area * rf * rf, max_width = by_width * rf, max_height = by_height * rf;
    std::vector<uint8_t> seen((size_t)nr * nc, 0);
    std::vector<int> stack, blob;
    int64_t ignored = 0;
    for (int r0 = 0; r0 < nr; ++r0)
        for (int c0 = 0; c0 < nc; ++c0) {
            const size_t i0 = (size_t)r0 * nc + c0;
            const uint16_t label = labels[i0];
            if (seen[i0] || label == 0 || label == ANH_LABEL_IGNORE) continue;
            blob.clear(); stack.clear();
            stack.push_back((int)i0); seen[i0] = 1;
            int min_x = c0, max_x = c0, min_y = r0, max_y = r0;
            while (!stack.empty()) {
                const int i = stack.back(); stack.pop_back();
                blob.push_back(i);
                const int r = i / nc, c = i - r * nc;
                min_x = std::min(min_x, c); max_x = std::max(max_x, c); min_y = std::min(min_y, r); max_y = std::max(max_y, r);
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int rr = r + dr, cc = c + dc;
                        if ((dr == 0 && dc == 0) || rr < 0 || rr >= nr || cc < 0 || cc >= nc) continue;
                        const size_t j = (size_t)rr * nc + cc;
                        if (!seen[j] && labels[j] == label) { seen[j] = 1; stack.push_back((int)j); }
                    }
            }
            const bool too_large = (double)blob.size() > max_points || (double)(max_x - min_x + 1) > max_width || (double)(max_y - min_y + 1) > max_height;
            if (too_large) {
                for (int i : blob) labels[i] = ANH_LABEL_IGNORE;
                ignored += (int64_t)blob.size();
            }
        }
    return ignored;
}

// dlib count_steps_without_decrease [UPSTREAM-UNVERIFIED]: walk the history from newest to oldest, keep a least-squares
// line through what has been seen, and remember the longest suffix for which "the loss is going down" is not
// more likely than probability_of_decrease.
int64_t count_steps_without_decrease(const double* values, int64_t n, double probability_of_decrease) {
    double cnt = 0, sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0;
    int64_t longest = 0;
    for (int64_t back = 1; back <= n; ++back) {
        const double x = cnt, y = values[n - back];
        cnt += 1; sx += x; sy += y; sxx += x * x; sxy += x * y; syy += y * y;
        if (cnt <= 2) continue;
        const double vxx = sxx - sx * sx / cnt, vxy = sxy - sx * sy / cnt, vyy = syy - sy * sy / cnt;
        const double slope = vxy / vxx;
        const double resid = std::max(0.0, (vyy - slope * vxy) / (cnt - 2));
        const double stderr_slope = std::sqrt(resid / vxx);
        // x runs backwards in time, so a positive slope here is a decreasing loss
        const double p_decreasing = stderr_slope == 0 ? (slope > 0 ? 1.0 : 0.0)
                                                      : 1.0 - 0.5 * std::erfc(slope / stderr_slope / std::sqrt(2.0));
        if (p_decreasing < probability_of_decrease) longest = back;
    }
    return longest;
}

// dnn_trainer's schedule [UPSTREAM-UNVERIFIED]: every `threshold/200`-ish steps (budget of 200 per step) test the
// history; if the loss has not decreased over `threshold` steps, multiply the rate by `shrink` and forget the
// oldest `dump_amount` values.
void LrSchedule::record(double loss) {
    if (shrink != 1.0 && check_budget > threshold) {
        check_budget = 0;
        std::vector<double> h(history.begin(), history.end());
        steps_without_progress = (unsigned long)count_steps_without_decrease(h.data(), (int64_t)h.size(), 0.51);
        if (steps_without_progress >= threshold) {
            // second look without the largest 10% of the values (robust variant)
            std::vector<double> sorted = h;
            std::sort(sorted.begin(), sorted.end());
            const double cap = sorted.empty() ? 0 : sorted[(size_t)((sorted.size() - 1) * 0.9)];
            std::vector<double> trimmed;
            for (double v : h) if (v <= cap) trimmed.push_back(v);
            steps_without_progress = (unsigned long)count_steps_without_decrease(trimmed.data(), (int64_t)trimmed.size(), 0.51);
            if (steps_without_progress >= threshold * 9 / 10) {
                lr *= shrink;
                for (unsigned long i = 0; i < dump_amount && !history.empty(); ++i) history.pop_front();
                steps_without_progress = 0;
            }
        }
    }
    check_budget += 200;
    history.push_back(loss);
    while (history.size() > (size_t)threshold * 2 + 1) history.pop_front();
}

}  // namespace anh
