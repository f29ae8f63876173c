// hostlogic.h — CPU-side pieces of the path that stay on the host: tiler, set_weights, crop rect, outpaint,
// and the trainer's learning-rate schedule.
#pragma once
#include <cstdint>
#include <deque>
#include <string>
#include <vector>

#include "../../include/annonet_hip.h"

namespace anh {

std::vector<anh_tile> make_tiles(int width, int height, const anh_tiling_params& p);
void set_weights(const uint16_t* labels, int nr, int nc, double class_weight, double image_weight, anh_wlabel* out);
void set_weights_table(const unsigned* counts, const unsigned* first_position, int n_labels, long long pixels, double class_weight,
                       double image_weight, float* table);
anh_rect random_rect_containing_point(uint32_t draw_x, uint32_t draw_y, long px, long py, long w, long h);
void outpaint(uint8_t* image, int nr, int nc, int channels, anh_rect inside);
int64_t count_steps_without_decrease(const double* values, int64_t n, double probability_of_decrease);
// annonet.dnn envelope: dlib serialize framing of (string classes_json, double downscaling_factor, string serialized_net)
std::string dnn_envelope_pack(const std::string& classes_json, double downscaling_factor, const std::string& net_blob);
void dnn_envelope_unpack(const std::string& file, std::string& classes_json, double& downscaling_factor, std::string& net_blob);
int64_t ignore_large_nonzero_regions(uint16_t* labels, int nr, int nc, double by_area, double by_width, double by_height, int receptive_field_side);

// Geometry of the net input cut around one tile (annonet_infer.cpp:46-66).
struct TileWindow { int left, top, width, height; };
TileWindow tile_window(const anh_tile& t, int levels);

// dlib dnn_trainer's "shrink the learning rate when the loss stops decreasing" rule [UPSTREAM-UNVERIFIED].
struct LrSchedule {
    double lr = 0.1, shrink = 0.1;
    unsigned long threshold = 2000, dump_amount = 400;
    std::deque<double> history;
    unsigned long check_budget = 0;
    unsigned long steps_without_progress = 0;
    void record(double loss);
};

}  // namespace anh
