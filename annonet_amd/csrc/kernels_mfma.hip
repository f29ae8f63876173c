// kernels_mfma.hip — bf16 MFMA implicit-GEMM kernels for the dense 3x3 convolutions (gfx950).
#include <hip/hip_runtime.h>

#include "common.h"
#include "kernels.h"

namespace anh {

bool mfma_conv_supported(const ConvArgs&) { return false; }
void launch_conv_mfma(const ConvArgs&, hipStream_t) { fail(ANH_ERR_INTERNAL, "conv_mfma: unsupported shape"); }
bool mfma_wgrad_supported(const WgradArgs&) { return false; }
void launch_wgrad_mfma(const WgradArgs&, hipStream_t) { fail(ANH_ERR_INTERNAL, "wgrad_mfma: unsupported shape"); }
int64_t wgrad_mfma_scratch_floats(const WgradArgs&) { return 0; }

}  // namespace anh
