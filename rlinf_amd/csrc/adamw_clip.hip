// adamw_clip.hip -- global-norm gradient clipping + two-group AdamW on flat f32 buffers, gfx950.
//
// Replaces FSDPModelManager.optimizer_step (rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:429-463):
//   torch.nn.utils.clip_grad_norm_ (strategy/fsdp.py:363-369, the all-NO_SHARD branch)
//   + torch.optim.AdamW.step with the actor / value_head parameter groups (fsdp_model_manager.py:501-590),
//   including "skip the step when the norm is non-finite".
// Two launches per optimizer step: (1) sum the split-K gradient slabs, scale, write the reduced gradient
// and per-block sum-of-squares partials; (2) every block re-reduces the (<= 1024) partials, derives the
// clip coefficient and applies clip + AdamW to its slice.  HBM-bound: 28 B per parameter (+4 B per extra
// gradient slab); at 287 504 parameters it is launch-latency bound, which is why it is only two launches.

#include <algorithm>

#include "rlx_common.h"

namespace rlx {
namespace {

constexpr int kMaxParts = 1024;

// state (device, int32[2]): [0] = optimizer steps applied so far, [1] = "the previous call applied a step"
// (folded into [0] here, i.e. strictly after that call's update kernel and before this call's), so that a
// captured hipGraph can be replayed without host-side step bookkeeping.
__global__ __launch_bounds__(256) void grad_reduce_sqnorm(float* __restrict__ grads, long long n, int nslab, float scale,
                                                          double* __restrict__ partials, int* __restrict__ state) {
    __shared__ double s_red[4];
    if (state != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && state[1] != 0) {
        state[0] += 1;
        state[1] = 0;
    }
    double acc[1] = {0.0};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = grads[i];
        for (int k = 1; k < nslab; ++k) g += grads[(long long)k * n + i];
        g *= scale;
        if (nslab > 1 || scale != 1.f) grads[i] = g;
        acc[0] += (double)g * (double)g;
    }
    block_sum<1>(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(256) void clip_adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, rlx_adamw_params a,
                                                         const double* __restrict__ partials, int nparts,
                                                         float* __restrict__ stats, int* __restrict__ state) {
    __shared__ double s_red[4];
    __shared__ float s_coef;
    __shared__ int s_skip;
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc[0] += partials[i];
    block_sum<1>(acc, s_red);
    if (threadIdx.x == 0) {
        const float total_norm = (float)sqrt(acc[0]);
        float coef = 1.f;
        if (a.max_grad_norm > 0.f) coef = fminf(a.max_grad_norm / (total_norm + 1e-6f), 1.0f);  // clip_grad_norm_
        s_coef = coef;
        s_skip = !isfinite(total_norm);
        if (blockIdx.x == 0) {
            stats[0] = total_norm;
            stats[1] = s_skip ? 0.f : 1.f;
            if (state != nullptr) state[1] = s_skip ? 0 : 1;
        }
    }
    __syncthreads();
    const float coef = s_coef;
    const bool skip = s_skip != 0;
    // bias corrections in double, like torch's python scalars
    const int step = state != nullptr ? state[0] + 1 : a.step;  // state[0] is stable for the whole launch
    const double bc1 = 1.0 - pow((double)a.beta1, (double)step);
    const double bc2 = 1.0 - pow((double)a.beta2, (double)step);
    const float bc2_sqrt = (float)sqrt(bc2);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float lr = 0.f;
        bool in_group = false;
#pragma unroll
        for (int k = 0; k < RLX_ADAMW_MAX_GROUPS; ++k) {
            if (k < a.n_groups && i >= a.groups[k].begin && i < a.groups[k].end) {
                lr = a.groups[k].lr;
                in_group = true;
            }
        }
        const float gi = g[i] * coef;  // grads.mul_(clip_coef_clamped): always applied
        g[i] = gi;
        if (skip || !in_group) continue;
        const float step_size = (float)((double)lr / bc1);
        float pi = p[i] * (float)(1.0 - (double)lr * (double)a.weight_decay);       // param.mul_(1 - lr*wd)
        const float mi = m[i] + (gi - m[i]) * (float)(1.0 - (double)a.beta1);       // exp_avg.lerp_(grad, 1-b1)
        const float vi = v[i] * a.beta2 + (float)(1.0 - (double)a.beta2) * gi * gi; // mul_(b2).addcmul_(g,g,1-b2)
        const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
        pi = pi - step_size * (mi / denom);                                          // addcdiv_(m, denom, -step_size)
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ g, long long n, int nslab, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s = g[i];
        for (int k = 1; k < nslab; ++k) s += g[(long long)k * n + i];
        out[i] = s;
    }
}

int grid_for(long long n) {
    return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, std::min<long long>(kMaxParts, (long long)num_cu() * 4)));
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_adamw_workspace_bytes(int64_t n) {
    (void)n;
    return (size_t)kMaxParts * sizeof(double);
}

extern "C" int rlx_sum_slabs(const float* grads, int64_t n, int slabs, float* out, rlx_stream_t stream) {
    RLX_REQUIRE(n >= 0 && slabs >= 1, "rlx_sum_slabs: bad sizes");
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(grads && out, "rlx_sum_slabs: NULL argument");
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), grads, (long long)n,
                       slabs, out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_clip_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const rlx_adamw_params* p, float* stats, int32_t* step_state, void* workspace,
                                   size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(p != nullptr, "rlx_clip_adamw_step: NULL params struct");
    RLX_REQUIRE(n >= 0 && (p->step >= 1 || step_state != nullptr) && p->n_groups >= 0 && p->n_groups <= RLX_ADAMW_MAX_GROUPS && p->grad_partials >= 1,
                "rlx_clip_adamw_step: bad sizes (n=%lld step=%d groups=%d slabs=%d)", (long long)n, p->step, p->n_groups,
                p->grad_partials);
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(params && grads && exp_avg && exp_avg_sq && stats && workspace, "rlx_clip_adamw_step: NULL argument");
    if (workspace_bytes < rlx_adamw_workspace_bytes(n)) {
        set_error("rlx_clip_adamw_step: workspace too small");
        return RLX_ENOSPC;
    }
    for (int k = 0; k < p->n_groups; ++k)
        RLX_REQUIRE(p->groups[k].begin >= 0 && p->groups[k].end <= n && p->groups[k].begin <= p->groups[k].end,
                    "rlx_clip_adamw_step: group %d range out of bounds", k);
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* partials = static_cast<double*>(workspace);
    const int nblk = grid_for(n);
    hipLaunchKernelGGL(grad_reduce_sqnorm, dim3(nblk), dim3(256), 0, s, grads, (long long)n, p->grad_partials, p->grad_scale,
                       partials, step_state);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(clip_adamw_kernel, dim3(nblk), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, (long long)n, *p,
                       partials, nblk, stats, step_state);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
