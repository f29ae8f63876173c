// annonet_infer_hip.h — annonet_infer() (annonet_infer.h:34-42) with the whole per-image loop on the GPU: the image is
// uploaded once, tiles are cut with a clamp-to-edge window inside the first conv kernel, logits are blended into
// resident class planes and the argmax runs on the device (annonet_infer.cpp:42-214); only the u16 label map comes
// back.  Same signature as the reference's function, so annonet_infer_main.cpp:468 compiles unchanged against it.
#ifndef ANNONET_INFER_HIP_H
#define ANNONET_INFER_HIP_H

#include <cstdlib>

#include "NetPimpl.h"
#include "tiling/dlib-wrapper.h"

// The reference leaves the blended class planes in temp.blended_output after every call (annonet_infer.cpp:80-85, annonet_infer.h:31).
// Here they live in HBM and travel back (K x H x W floats over PCIe) only when temp.keep_blended_output is set.  A drop-in host that
// READS temp.blended_output after the call keeps the reference's behaviour without touching its code: compile with
// -DANNONET_HIP_KEEP_BLENDED_OUTPUT=1, or run with ANH_KEEP_BLENDED_OUTPUT=1 in the environment — either makes the flag default to true.
#ifndef ANNONET_HIP_KEEP_BLENDED_OUTPUT
#define ANNONET_HIP_KEEP_BLENDED_OUTPUT 0
#endif
inline bool annonet_hip_keep_blended_default() {
    static const bool from_env = [] { const char* e = std::getenv("ANH_KEEP_BLENDED_OUTPUT"); return e && std::atoi(e) != 0; }();
    return ANNONET_HIP_KEEP_BLENDED_OUTPUT != 0 || from_env;
}

struct annonet_infer_temp {  // annonet_infer.h:26-32; the GPU path keeps its scratch inside the net handle
    NetPimpl::input_type input_tile;
    std::vector<dlib::point> detection_seeds;
    dlib::matrix<unsigned int> connected_blobs;
    std::vector<dlib::matrix<float>> blended_output;  // filled when keep_blended_output is set (see above for its default)
    bool keep_blended_output = annonet_hip_keep_blended_default();
};

inline void annonet_infer(NetPimpl::RuntimeNet& net, const NetPimpl::input_type& input_image, dlib::matrix<uint16_t>& result_image,
                          annonet_infer_temp& temp, const std::vector<double>& gains = std::vector<double>(),
                          const std::vector<double>& detection_levels = std::vector<double>(),
                          const tiling::parameters& tiling_parameters = tiling::parameters()) {
    anh_net_config cfg;
    NetPimpl::check(anh_runtime_config(net.handle(), &cfg));
    const int K = cfg.classes, H = (int)input_image.nr(), W = (int)input_image.nc();
    if (!gains.empty() && (int)gains.size() != K) throw std::runtime_error("annonet_infer: one gain per class expected");
    if (!detection_levels.empty() && (int)detection_levels.size() != K) throw std::runtime_error("annonet_infer: one detection level per class expected");
    result_image.set_size(H, W);
    std::vector<float> planes;
    if (temp.keep_blended_output) planes.resize((size_t)K * H * W);
    anh_tiling_params tp{tiling_parameters.max_tile_width, tiling_parameters.max_tile_height, tiling_parameters.overlap_x, tiling_parameters.overlap_y};
    NetPimpl::check(anh_infer(net.handle(), reinterpret_cast<const uint8_t*>(&*input_image.begin()), H, W, gains.empty() ? nullptr : gains.data(),
                              detection_levels.empty() ? nullptr : detection_levels.data(), &tp, &*result_image.begin(),
                              temp.keep_blended_output ? planes.data() : nullptr));
    if (temp.keep_blended_output) {
        temp.blended_output.resize(K);
        for (int k = 0; k < K; ++k) {
            temp.blended_output[k].set_size(H, W);
            std::copy(planes.begin() + (size_t)k * H * W, planes.begin() + (size_t)(k + 1) * H * W, temp.blended_output[k].begin());
        }
    }
}

#endif  // ANNONET_INFER_HIP_H
