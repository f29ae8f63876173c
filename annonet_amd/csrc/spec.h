// spec.h — the net's layer list and dimension maths.
//
// Stands in for dlib-dnn-pimpl-wrapper/{NetStructure.h,NetDimensions.{h,cpp}}, which the reference
// includes (annonet_train.h:16, annonet_train_cuda.vcxproj:247,602-604) but does not ship.  DESIGN.md §2
// gives the structure in prose; the oracle carries its own independent statement of the same list and
// tests/test_host_logic.py checks the two agree.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/annonet_hip.h"

namespace anh {

struct Spec {
    anh_net_config cfg{};
    std::vector<anh_layer_desc> layers;
    int64_t n_params = 0;
    int64_t n_running = 0;
    int n_bn = 0;

    static Spec build(const anh_net_config& cfg);

    int64_t filter_count(int layer) const {
        const anh_layer_desc& L = layers[layer];
        return (int64_t)L.k * L.k * L.cin * L.cout;
    }
    static int out_dim(const anh_layer_desc& L, int in) {
        if (L.type == 0) return in + 2 * L.pad < L.k ? 0 : (in + 2 * L.pad - L.k) / L.stride + 1;
        return L.stride * (in - 1) + L.k - 2 * L.pad;
    }
    int required_input_dim() const;
    static int recommended_input_dim(int levels, int n);
    bool valid_input_dim(int n) const { return n >= 1 && recommended_input_dim(cfg.levels, n) == n; }
};

}  // namespace anh
