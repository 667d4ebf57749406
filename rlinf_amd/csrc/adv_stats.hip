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

// a15 compute_rollout_metrics (rlinf/utils/metric_utils.py:422-506): masked sum / count / min / max of up to three arrays
// (rewards, advantages, returns) in ONE pass over them -- the reference gathers each with a boolean index and reduces it with
// three torch calls.  mask element e covers `per[k]` consecutive elements of array k (the loss mask is [.., 1] against
// [.., C] rewards under reward_type chunk_level).  out[k] = {sum, count, -min, max} in f64: sums reduce with SUM over ranks,
// (-min, max) with MAX, one collective each (scheduler/dist.py).
constexpr int kMetricArrays = 3;
struct MetricArgs {
    const float* x[kMetricArrays];
    long long n[kMetricArrays];
    int per[kMetricArrays];
    int count;
};
__global__ __launch_bounds__(256) void rollout_metrics_partial_kernel(MetricArgs a, const uint8_t* __restrict__ m,
                                                                      double* __restrict__ partials) {
    __shared__ double s_red[4 * 4];
    for (int k = 0; k < a.count; ++k) {
        double sum[2] = {0.0, 0.0};
        float lo = INFINITY, hi = -INFINITY;
        const long long stride = (long long)gridDim.x * blockDim.x;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n[k]; i += stride) {
            const bool on = m ? m[i / a.per[k]] != 0 : true;
            const float v = a.x[k][i];
            if (on) {
                sum[0] += (double)v;
                sum[1] += 1.0;
                lo = fminf(lo, v);
                hi = fmaxf(hi, v);
            }
        }
        block_sum<2>(sum, s_red);
        __syncthreads();
        // min / max: wave shuffles, then through LDS
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off, RLX_WAVE));
            hi = fmaxf(hi, __shfl_xor(hi, off, RLX_WAVE));
        }
        float* s_mm = reinterpret_cast<float*>(s_red + 8);
        if ((threadIdx.x & 63) == 0) s_mm[(threadIdx.x >> 6) * 2] = lo, s_mm[(threadIdx.x >> 6) * 2 + 1] = hi;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) lo = fminf(lo, s_mm[w * 2]), hi = fmaxf(hi, s_mm[w * 2 + 1]);
            double* p = partials + ((size_t)blockIdx.x * kMetricArrays + k) * 4;
            p[0] = sum[0], p[1] = sum[1], p[2] = (double)-lo, p[3] = (double)hi;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void rollout_metrics_final_kernel(const double* __restrict__ partials, int nparts, int count,
                                                                    double* __restrict__ out) {
    __shared__ double s_red[2 * 4];
    __shared__ double s_mm[2 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < count; ++k) {
        double sum[2] = {0.0, 0.0};
        double nlo = -INFINITY, hi = -INFINITY;
        for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
            const double* p = partials + ((size_t)i * kMetricArrays + k) * 4;
            sum[0] += p[0], sum[1] += p[1];
            nlo = fmax(nlo, p[2]), hi = fmax(hi, p[3]);
        }
        block_sum<2>(sum, s_red);
        // max is order-independent: a wave butterfly + four wave results (thread 0 used to walk 256 LDS pairs one after the other:
        // 17 us for a launch that reduces a few hundred numbers)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            nlo = fmax(nlo, __shfl_xor(nlo, off, 64));
            hi = fmax(hi, __shfl_xor(hi, off, 64));
        }
        if (lane == 0) s_mm[wave * 2] = nlo, s_mm[wave * 2 + 1] = hi;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 1; t < (int)(blockDim.x >> 6); ++t) nlo = fmax(nlo, s_mm[t * 2]), hi = fmax(hi, s_mm[t * 2 + 1]);
            out[k * 4 + 0] = sum[0], out[k * 4 + 1] = sum[1], out[k * 4 + 2] = nlo, out[k * 4 + 3] = hi;
        }
        __syncthreads();
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

extern "C" size_t rlx_rollout_metrics_workspace_bytes(void) { return (size_t)256 * kMetricArrays * 4 * sizeof(double); }

extern "C" int rlx_rollout_metrics(const float* const* arrays, const int64_t* sizes, int count, const uint8_t* mask,
                                   int64_t mask_elems, double* out, void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(arrays && sizes && out && workspace && count >= 1 && count <= kMetricArrays, "rlx_rollout_metrics: bad argument");
    if (workspace_bytes < rlx_rollout_metrics_workspace_bytes()) {
        set_error("rlx_rollout_metrics: workspace too small");
        return RLX_ENOSPC;
    }
    MetricArgs a{};
    a.count = count;
    long long nmax = 0;
    for (int k = 0; k < count; ++k) {
        RLX_REQUIRE(sizes[k] >= 0 && (sizes[k] == 0 || arrays[k] != nullptr), "rlx_rollout_metrics: array %d is NULL", k);
        a.x[k] = arrays[k];
        a.n[k] = sizes[k];
        a.per[k] = 1;
        if (mask != nullptr) {
            RLX_REQUIRE(mask_elems >= 1 && sizes[k] % mask_elems == 0, "rlx_rollout_metrics: array %d (%lld elements) is not a "
                        "multiple of the mask (%lld)", k, (long long)sizes[k], (long long)mask_elems);
            a.per[k] = (int)std::max<long long>(1, sizes[k] / mask_elems);
        }
        nmax = std::max<long long>(nmax, sizes[k]);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (int)std::max<long long>(1, std::min<long long>((nmax + 1023) / 1024, 256));
    double* partials = static_cast<double*>(workspace);
    hipLaunchKernelGGL(rollout_metrics_partial_kernel, dim3(nblk), dim3(256), 0, s, a, mask, partials);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(rollout_metrics_final_kernel, dim3(1), dim3(256), 0, s, partials, nblk, count, out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
