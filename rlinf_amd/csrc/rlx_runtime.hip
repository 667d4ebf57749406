// rlx_runtime.hip -- error plumbing and device queries shared by every entry point of librlx_hip.so.
#include <stdarg.h>
#include <string.h>

#include "rlx_common.h"

namespace rlx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int num_cu() {
    static thread_local int cached_dev = -1, cached_cu = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cached_cu;
    if (dev != cached_dev) {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) {
            cached_cu = cu;
            cached_dev = dev;
        }
    }
    return cached_cu;
}

}  // namespace rlx

extern "C" int rlx_version(void) { return RLX_VERSION; }

extern "C" int rlx_dev_variants(void) {
#ifdef RLX_DEV_VARIANTS
    return 1;
#else
    return 0;
#endif
}

extern "C" const char* rlx_last_error(void) { return rlx::g_err; }

extern "C" int rlx_abi_struct_sizes(size_t* sizes, int n) {
    const size_t all[] = {sizeof(rlx_gae_params), sizeof(rlx_ppo_loss_params), sizeof(rlx_gather_field), sizeof(rlx_adamw_group),
                          sizeof(rlx_adamw_params), sizeof(rlx_mlp_layout), sizeof(rlx_value_job), sizeof(rlx_rollout_step),
                          sizeof(rlx_ppo_step_args), sizeof(rlx_decoupled_loss_params), sizeof(rlx_token_rows),
                          sizeof(rlx_token_loss_params), sizeof(rlx_copy_segment)};
    const int count = (int)(sizeof(all) / sizeof(all[0]));
    for (int i = 0; sizes != nullptr && i < n && i < count; ++i) sizes[i] = all[i];
    return count;
}

extern "C" int rlx_device_info(int* num_cu_out, int* wave_size) {
    int dev = 0;
    RLX_HIP_CHECK(hipGetDevice(&dev));
    int cu = 0, ws = 0;
    RLX_HIP_CHECK(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev));
    RLX_HIP_CHECK(hipDeviceGetAttribute(&ws, hipDeviceAttributeWarpSize, dev));
    if (num_cu_out) *num_cu_out = cu;
    if (wave_size) *wave_size = ws;
    return RLX_OK;
}
