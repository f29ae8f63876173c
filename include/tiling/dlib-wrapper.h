// tiling/dlib-wrapper.h — tiling::dlib_tile / tiling::get_tiles with dlib::rectangle members, as used by
// annonet_infer.cpp:22,42,46-47,118,128,138,148-149.
#ifndef ANNONET_HIP_TILING_DLIB_WRAPPER_H
#define ANNONET_HIP_TILING_DLIB_WRAPPER_H

#include "tiling.h"

namespace tiling {

struct dlib_tile { dlib::rectangle full_rect, unique_rect; };

inline std::vector<dlib_tile> get_tiles(int width, int height, const parameters& p = parameters()) {
    std::vector<dlib_tile> out;
    for (const tile& t : get_tiles_raw(width, height, p)) {
        dlib_tile d;
        d.full_rect = dlib::rectangle(t.full_rect.left, t.full_rect.top, t.full_rect.right, t.full_rect.bottom);
        d.unique_rect = dlib::rectangle(t.unique_rect.left, t.unique_rect.top, t.unique_rect.right, t.unique_rect.bottom);
        out.push_back(d);
    }
    return out;
}

}  // namespace tiling
#endif
