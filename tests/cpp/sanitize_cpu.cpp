// Sanitizer drive of the CPU-side code ("build the CPU restatement under ASan/UBSan in this container"; GPU sanitizers
// are not available on the pool): the library's host logic (hostlogic.cpp, spec.cpp) and the oracle (oracle/annonet_oracle.cpp) are
// compiled with -fsanitize=address,undefined and run through their edge cases: ragged tilings, rectangles outside the image,
// truncated / damaged envelopes, blobs at the image border, a tiny training step and a tiled inference.  Any report aborts.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../annonet_amd/csrc/hostlogic.h"
#include "../../annonet_amd/csrc/spec.h"

extern "C" {
struct orc_tile { long full[4]; long unique[4]; };
const char* orc_last_error();
void* orc_net_create(int levels, int in_ch, int classes, double scaler, int min_filters);
void orc_net_destroy(void* h);
int64_t orc_net_param_count(void* h);
float* orc_net_params(void* h);
int orc_required_input_dim(void* h);
int orc_recommended_input_dim(int levels, int n);
int orc_forward(void* h, const uint8_t* img, int n, int hh, int ww, float* out_nchw);
int orc_train_step(void* h, const uint8_t* images, const uint16_t* labels, const float* weights, int n, int hh, int ww, double loss_scale_n, int apply_update, double* loss);
int orc_set_weights(const uint16_t* labels, int nr, int nc, double class_weight, double image_weight, float* weights_out);
int orc_infer(void* h, const uint8_t* image, int H, int W, const double* gains, const double* detection_levels, long max_w, long max_h, long ov_x, long ov_y,
              const orc_tile* tiles_in, int64_t n_tiles_in, uint16_t* result, float* blended_out);
}

namespace anh { void set_last_error(const std::string&) {} }

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

template <typename F> static bool throws(F&& f) { try { f(); } catch (const std::exception&) { return true; } return false; }

int main() {
    using namespace anh;
    unsigned seed = 7;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    // ---- tiler: every size against small tiles, incl. sizes below the overlap ----
    for (int w : {1, 7, 35, 36, 96, 97, 200, 1023}) for (int h : {1, 40, 96, 333}) {
        anh_tiling_params p{96, 64, 19, 19};
        std::vector<anh_tile> t;
        if (throws([&] { t = make_tiles(w, h, p); })) continue;     // rejected geometries are fine; memory errors are not
        REQUIRE(!t.empty());
        for (const auto& x : t) REQUIRE(x.full_rect.left >= 0 && x.full_rect.right < w && x.full_rect.top >= 0 && x.full_rect.bottom < h);
    }
    // ---- set_weights incl. all-ignore and single-pixel images ----
    for (int n : {1, 5, 64}) {
        std::vector<uint16_t> lab((size_t)n * n);
        for (auto& v : lab) v = (rnd() % 5 == 0) ? 65535 : (uint16_t)(rnd() % 40);
        std::vector<anh_wlabel> out(lab.size());
        set_weights(lab.data(), n, n, 0.5, 0.5, out.data());
        std::fill(lab.begin(), lab.end(), (uint16_t)65535);
        set_weights(lab.data(), n, n, 1.0, 0.0, out.data());
    }
    // ---- outpaint with rectangles partly / entirely outside ----
    {
        std::vector<uint8_t> img(30 * 40 * 3, 9);
        outpaint(img.data(), 30, 40, 3, anh_rect{5, 5, 20, 10});
        outpaint(img.data(), 30, 40, 3, anh_rect{-10, -10, 100, 100});
        outpaint(img.data(), 30, 40, 3, anh_rect{50, 50, 60, 60});
        outpaint(img.data(), 30, 40, 1, anh_rect{39, 29, 39, 29});
    }
    // ---- annonet.dnn envelope: round trip and every truncation ----
    {
        const std::string blob(1000, '\x5a');
        const std::string file = dnn_envelope_pack("{\"anno_classes\": []}", 1.75, blob);
        std::string j, b; double f = 0;
        dnn_envelope_unpack(file, j, f, b);
        REQUIRE(f == 1.75 && b == blob);
        for (size_t cut = 0; cut < file.size(); cut += 7) REQUIRE(throws([&] { dnn_envelope_unpack(file.substr(0, cut), j, f, b); }));
        std::string damaged = file;
        damaged[0] = '\x7f';
        (void)throws([&] { dnn_envelope_unpack(damaged, j, f, b); });
        for (double v : {0.0, -0.0, 1e-310, 1e300, std::numeric_limits<double>::infinity(), std::nan("")}) { dnn_envelope_unpack(dnn_envelope_pack("", v, ""), j, f, b); REQUIRE(std::isnan(v) ? std::isnan(f) : f == v); }
    }
    // ---- large-region ignoring with blobs on the border ----
    {
        const int nr = 50, nc = 70;
        std::vector<uint16_t> lab((size_t)nr * nc, 0);
        for (int r = 0; r < nr; ++r) for (int c = 0; c < nc; ++c) if (r < 20 || c > 60) lab[(size_t)r * nc + c] = 1 + (c / 35);
        REQUIRE(ignore_large_nonzero_regions(lab.data(), nr, nc, 0.01, std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(), 35) > 0);
        (void)ignore_large_nonzero_regions(lab.data(), nr, nc, std::numeric_limits<double>::infinity(), 0.5, 0.5, 35);
    }
    // ---- learning-rate schedule ----
    {
        LrSchedule s;
        s.threshold = 20; s.dump_amount = 5; s.shrink = 0.5;
        for (int i = 0; i < 400; ++i) s.record(1.0 + 1e-3 * (double)(rnd() % 100));
        REQUIRE(s.lr < 0.1);
        std::vector<double> v(100, 1.0);
        (void)count_steps_without_decrease(v.data(), (int64_t)v.size(), 0.51);
        (void)count_steps_without_decrease(v.data(), 0, 0.51);
    }
    // ---- spec / dimension maths for every build variant ----
    for (int levels = 0; levels <= 3; ++levels) for (int ch : {1, 3}) {
        const Spec s = Spec::build(anh_net_config{levels, ch, 5, 0.3, 2, ANH_FP32});
        REQUIRE(s.n_params > 0 && s.required_input_dim() >= 1);
        for (int n : {1, 17, 227, 1000}) REQUIRE(Spec::recommended_input_dim(levels, n) >= n);
    }
    // ---- oracle: forward, one training step, tiled inference with gains and detection levels ----
    {
        void* net = orc_net_create(2, 3, 3, 0.25, 4);
        REQUIRE(net);
        float* p = orc_net_params(net);
        for (int64_t i = 0; i < orc_net_param_count(net); ++i) p[i] = (float)((int)(rnd() % 2001) - 1000) * 1e-4f;
        const int d = orc_recommended_input_dim(2, 36);
        std::vector<uint8_t> img((size_t)2 * d * d * 3);
        for (auto& v : img) v = (uint8_t)rnd();
        std::vector<uint16_t> lab((size_t)2 * d * d);
        for (auto& v : lab) v = (rnd() % 9 == 0) ? 65535 : (uint16_t)(rnd() % 3);
        std::vector<float> w(lab.size());
        REQUIRE(orc_set_weights(lab.data(), d, d, 0.5, 0.5, w.data()) == 0 && orc_set_weights(lab.data() + (size_t)d * d, d, d, 0.5, 0.5, w.data() + (size_t)d * d) == 0);
        std::vector<float> out((size_t)2 * 3 * d * d);
        REQUIRE(orc_forward(net, img.data(), 2, d, d, out.data()) == 0);
        double loss = 0;
        REQUIRE(orc_train_step(net, img.data(), lab.data(), w.data(), 2, d, d, 2.0, 1, &loss) == 0 && std::isfinite(loss));
        const int H = 97, W = 131;
        std::vector<uint8_t> big((size_t)H * W * 3);
        for (auto& v : big) v = (uint8_t)rnd();
        std::vector<uint16_t> res((size_t)H * W);
        std::vector<float> planes((size_t)3 * H * W);
        const double gains[3] = {0, 0.1, -0.1}, det[3] = {0, 0.05, 0.05};
        const int ov = orc_required_input_dim(net);
        REQUIRE(orc_infer(net, big.data(), H, W, gains, det, 96, 80, ov, ov, nullptr, 0, res.data(), planes.data()) == 0);
        REQUIRE(orc_infer(net, big.data(), H, W, nullptr, nullptr, 4096, 4096, ov, ov, nullptr, 0, res.data(), nullptr) == 0);
        orc_net_destroy(net);
    }
    std::printf("sanitize_cpu ok\n");
    return 0;
}
