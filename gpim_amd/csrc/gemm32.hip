// gemm32.hip -- float instantiations of the tile engine (precision = 'single'; see gemm_kernel.hpp / gemm.hip).
#include "gemm_kernel.hpp"

int launch_gemm_f32(gpimhip_ctx* h, bool a_km, bool b_km, int epi, const GemmArgs& g) {
    return launch_gemm_t<float>(h, a_km, b_km, epi, g);
}
