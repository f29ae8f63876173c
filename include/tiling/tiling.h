// tiling/tiling.h — drop-in for the reference's tiling submodule (annonet_infer.h:23; used at annonet_infer.cpp:42,
// annonet_infer_main.cpp:423-427).  The submodule is absent from the reference snapshot; the contract implemented by
// anh_get_tiles is the one annonet_infer.cpp:42-164 relies on.
#ifndef ANNONET_HIP_TILING_H
#define ANNONET_HIP_TILING_H

#include <stdexcept>
#include <string>
#include <vector>

#include "../annonet_hip.h"

namespace tiling {

struct parameters {  // annonet_infer_main.cpp:423-427
    int max_tile_width = 1024;
    int max_tile_height = 1024;
    int overlap_x = 0;
    int overlap_y = 0;
};

struct rect { long left, top, right, bottom; };  // inclusive
struct tile { rect full_rect, unique_rect; };

inline std::vector<tile> get_tiles_raw(int width, int height, const parameters& p) {
    anh_tiling_params cp{p.max_tile_width, p.max_tile_height, p.overlap_x, p.overlap_y};
    anh_tile* out = nullptr;
    size_t n = 0;
    if (anh_get_tiles(width, height, &cp, &out, &n) != ANH_OK) throw std::runtime_error(std::string("tiling: ") + anh_last_error());
    std::vector<tile> tiles(n);
    for (size_t i = 0; i < n; ++i) {
        tiles[i].full_rect = {out[i].full_rect.left, out[i].full_rect.top, out[i].full_rect.right, out[i].full_rect.bottom};
        tiles[i].unique_rect = {out[i].unique_rect.left, out[i].unique_rect.top, out[i].unique_rect.right, out[i].unique_rect.bottom};
    }
    anh_free(out);
    return tiles;
}

}  // namespace tiling
#endif
