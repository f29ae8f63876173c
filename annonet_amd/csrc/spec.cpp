#include "spec.h"

#include <algorithm>
#include <cmath>

#include "common.h"

namespace anh {

namespace {
const int kBaseWidth[4] = {32, 64, 128, 256};
}

// Encoder-decoder with additive skips:
//   stem   con 5x5 s1 p2            image -> C0                 bn relu
//   down_l con 3x3 s2 p0            C(l-1) -> C(l)              bn relu      l = 1..L
//   enc_l  con 3x3 s1 p1            C(l) -> C(l)                bn relu
//   up_l   cont 3x3 s2 p0           C(l) -> C(l-1)              bn relu      l = L..1
//   dec_l' con 3x3 s1 p1            (up_l + enc_(l-1)) -> C(l-1) bn relu
//   head   con 1x1 + bias           C0 -> K
// Widths follow SetNetWidth(scaler, min_filter_count) (annonet_train_main.cpp:402): C(l) = max(min, round(scaler*base(l))).
Spec Spec::build(const anh_net_config& cfg) {
    ANH_REQUIRE(cfg.levels >= 0 && cfg.levels <= 3, "level count must be 0..3");
    ANH_REQUIRE(cfg.in_channels == 1 || cfg.in_channels == 3, "input channels must be 1 or 3");
    ANH_REQUIRE(cfg.classes >= 1 && cfg.classes <= 64, "class count must be 1..64");
    ANH_REQUIRE(cfg.width_scaler > 0.0 && cfg.min_filters >= 1, "net width must be positive");
    ANH_REQUIRE(cfg.precision == ANH_FP32 || cfg.precision == ANH_BF16, "unknown precision");
    Spec s;
    s.cfg = cfg;
    auto push = [&](int type, int k, int stride, int pad, int cin, int cout, int in_a, int in_b, bool bn) {
        anh_layer_desc L{};
        L.type = type; L.k = k; L.stride = stride; L.pad = pad; L.cin = cin; L.cout = cout;
        L.in_a = in_a; L.in_b = in_b; L.has_bn = bn ? 1 : 0; L.has_bias = bn ? 0 : 1;
        L.b_off = L.g_off = L.beta_off = L.rs_off = -1;
        L.w_off = s.n_params;
        s.n_params += (int64_t)k * k * cin * cout;
        if (L.has_bias) { L.b_off = s.n_params; s.n_params += cout; }
        if (L.has_bn) {
            L.g_off = s.n_params; s.n_params += cout;
            L.beta_off = s.n_params; s.n_params += cout;
            L.rs_off = s.n_running; s.n_running += 2 * (int64_t)cout;
            ++s.n_bn;
        }
        s.layers.push_back(L);
        return (int)s.layers.size() - 1;
    };
    int width[4];
    for (int l = 0; l <= cfg.levels; ++l) width[l] = std::max(cfg.min_filters, (int)std::lround(cfg.width_scaler * kBaseWidth[l]));
    int skip[4];
    skip[0] = push(0, 5, 1, 2, cfg.in_channels, width[0], -1, -2, true);
    for (int l = 1; l <= cfg.levels; ++l) {
        const int down = push(0, 3, 2, 0, width[l - 1], width[l], skip[l - 1], -2, true);
        skip[l] = push(0, 3, 1, 1, width[l], width[l], down, -2, true);
    }
    int top = skip[cfg.levels];
    for (int l = cfg.levels; l >= 1; --l) {
        const int up = push(1, 3, 2, 0, width[l], width[l - 1], top, -2, true);
        top = push(0, 3, 1, 1, width[l - 1], width[l - 1], up, skip[l - 1], true);
    }
    push(0, 1, 1, 0, width[0], cfg.classes, top, -2, false);
    return s;
}

// Side of the input window one output pixel sees.  con: (k-1)*jump more, jump *= stride; cont with k=3, s=2
// reads ceil(k/s) = 2 input positions: (2-1)*jump more, jump /= stride.
int Spec::required_input_dim() const {
    int field = 1, jump = 1;
    for (const anh_layer_desc& L : layers) {
        if (L.type == 0) { field += (L.k - 1) * jump; jump *= L.stride; }
        else { field += ((L.k + L.stride - 1) / L.stride - 1) * jump; jump /= L.stride; }
    }
    return field;
}

// A side d survives `levels` rounds of (3x3, stride 2, no padding) exactly, and the transposed convs restore it,
// iff d = 2^levels * m + (2^levels - 1) with m >= 1.
int Spec::recommended_input_dim(int levels, int n) {
    const int q = 1 << levels;
    int m = n <= q - 1 ? 1 : (n - (q - 1) + q - 1) / q;
    if (m < 1) m = 1;
    return q * m + q - 1;
}

}  // namespace anh
