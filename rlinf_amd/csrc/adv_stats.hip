// adv_stats.hip -- advantage normalisation from sufficient statistics (pipeline mode's "global advantage stats"), gfx950.
//
//   masked_stats          rlinf/utils/distributed.py:942-954   (count, sum, sum of squares) of x[mask], in float64
//   normalize_from_stats  rlinf/utils/distributed.py:957-965   (x - mean) * rsqrt(max(var, 0) + 1e-5) in float64 -> float32
// as used by EnvWorker.send_rollout_trajectories_pipeline (rlinf/workers/env/env_worker.py:1548-1572): every env rank /
// pipeline stage reduces its own advantages to three numbers, the numbers are summed across ranks (torch.distributed,
// 24 bytes), and every batch is normalised with the global mean and (uncorrected) variance.  Two streaming passes,
// 5 B / element and 8 B / element; the statistics never leave the device.

#include <algorithm>

#include "rlx_common.h"

namespace rlx {
namespace {

constexpr int kStatBlocks = 1024;

__global__ __launch_bounds__(256) void masked_stats_partial_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                                   long long n, double* __restrict__ partials) {
    __shared__ double s_red[3 * 4];
    double acc[3] = {0.0, 0.0, 0.0};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool on = m ? m[i] != 0 : true;
        const double v = (double)x[i];
        if (on) acc[0] += 1.0, acc[1] += v, acc[2] += v * v;
    }
    block_sum<3>(acc, s_red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 3 + 0] = acc[0];
        partials[blockIdx.x * 3 + 1] = acc[1];
        partials[blockIdx.x * 3 + 2] = acc[2];
    }
}
__global__ __launch_bounds__(256) void masked_stats_final_kernel(const double* __restrict__ partials, int nparts,
                                                                 double* __restrict__ stats, int accumulate) {
    __shared__ double s_red[3 * 4];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        acc[0] += partials[i * 3], acc[1] += partials[i * 3 + 1], acc[2] += partials[i * 3 + 2];
    }
    block_sum<3>(acc, s_red);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; ++k) stats[k] = accumulate ? stats[k] + acc[k] : acc[k];
    }
}
__global__ __launch_bounds__(256) void normalize_from_stats_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                                   float* __restrict__ out, long long n) {
    const double count = stats[0] < 1.0 ? 1.0 : stats[0];  // clamp_min(1.0)
    const double mean = stats[1] / count;
    double var = stats[2] / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + 1e-5);  // torch.rsqrt in float64
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = (float)(((double)x[i] - mean) * rstd);
}

// masked_normalization's last line (rlinf/utils/distributed.py:917-937, high_precision): masked-out elements enter as 0,
// mean = sum / count, var = sumsq / count - mean^2 (biased), out = (x - mean) / (sqrt(var) + eps), all in f64 -> f32.
// count == 0 or a rounding-negative variance give NaN there, and here.
__global__ __launch_bounds__(256) void masked_normalize_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask,
                                                               const double* __restrict__ stats, double eps,
                                                               float* __restrict__ out, long long n) {
    const double mean = stats[1] / stats[0];
    const double var = stats[2] / stats[0] - mean * mean;
    const double denom = sqrt(var) + eps;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = (mask == nullptr || mask[i]) ? (double)x[i] : 0.0;
        out[i] = (float)((v - mean) / denom);
    }
}

int stat_grid(long long n) {
    return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, std::min<long long>(kStatBlocks, (long long)num_cu() * 4)));
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_masked_stats_workspace_bytes(int64_t n) {
    (void)n;
    return (size_t)kStatBlocks * 3 * sizeof(double);
}

extern "C" int rlx_masked_stats(const float* x, const uint8_t* mask, int64_t n, double* stats, int accumulate, void* workspace,
                                size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(n >= 0 && stats && workspace, "rlx_masked_stats: bad argument");
    RLX_REQUIRE(n == 0 || x, "rlx_masked_stats: NULL x");
    if (workspace_bytes < rlx_masked_stats_workspace_bytes(n)) {
        set_error("rlx_masked_stats: workspace too small");
        return RLX_ENOSPC;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* partials = static_cast<double*>(workspace);
    const int nblk = n > 0 ? stat_grid(n) : 1;
    hipLaunchKernelGGL(masked_stats_partial_kernel, dim3(nblk), dim3(256), 0, s, x, mask, (long long)n, partials);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(masked_stats_final_kernel, dim3(1), dim3(256), 0, s, partials, nblk, stats, accumulate);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_normalize_from_stats(const float* x, const double* stats, float* out, int64_t n, rlx_stream_t stream) {
    RLX_REQUIRE(n >= 0 && stats, "rlx_normalize_from_stats: bad argument");
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(x && out, "rlx_normalize_from_stats: NULL argument");
    hipLaunchKernelGGL(normalize_from_stats_kernel, dim3(stat_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, stats,
                       out, (long long)n);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_masked_normalize(const float* x, const uint8_t* mask, const double* stats, double eps, float* out, int64_t n,
                                    rlx_stream_t stream) {
    RLX_REQUIRE(n >= 0 && stats, "rlx_masked_normalize: bad argument");
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(x && out, "rlx_masked_normalize: NULL argument");
    hipLaunchKernelGGL(masked_normalize_kernel, dim3(stat_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, mask, stats,
                       eps, out, (long long)n);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
