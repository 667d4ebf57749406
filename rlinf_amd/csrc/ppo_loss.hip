// ppo_loss.hip -- fused PPO clipped-surrogate + clipped-Huber value loss, forward and backward, gfx950.
//
// Replaces (after rlinf/algorithms/utils.py:280-376 preprocess_loss_inputs):
//   compute_ppo_actor_loss   rlinf/algorithms/losses.py:170-312
//   compute_ppo_critic_loss  rlinf/algorithms/losses.py:315-380  (+ huber_loss utils.py:20-23,
//   masked_mean / masked_mean_ratio rlinf/utils/utils.py:323-356, explained-variance stats
//   rlinf/utils/metric_utils.py:232-258)
// i.e. ~40 elementwise torch kernels + ~15 reductions per micro-batch become ONE streaming pass
// (80 B read per sample at A = 8) + a 1-block finalize; backward is one expand pass (36 B written).
// All scalars (loss, 15 metrics, gradient scales) stay on the device: no .item() syncs.
//
// The per-element derivative is computed in the forward pass, exactly as autograd would for the
// reference graph, including its tie rules: torch.max / torch.min send half the gradient to each
// argument on ties, clamp passes gradient on its closed interval, where() routes to the taken branch.

#include <algorithm>

#include "ppo_loss_math.h"
#include "rlx_common.h"

namespace rlx {
namespace {

using namespace loss;

struct LossArgs {
    const float* lp;
    const float* old;
    const float* adv;
    const float* v;
    const float* pv;
    const float* ret;
    const uint8_t* m;
    const int64_t* msum;
    float* g_lp;
    float* g_v;
    double* partials;
    long long n;
    rlx_ppo_loss_params p;
};

template <bool VEC4>
__global__ __launch_bounds__(256) void ppo_loss_fwd_kernel(LossArgs a) {
    __shared__ double s_red[NS * 4];
    const rlx_ppo_loss_params& p = a.p;
    const int K = p.raw_per_adv, S = p.sub_per_adv, R = K / S;
    const bool ratio_mode = p.max_episode_steps > 0 && a.m != nullptr && a.msum != nullptr;
    const float half_delta = (float)(0.5 * (double)p.huber_delta);
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < a.n; e += stride) {
        const bool on = a.m ? a.m[e] != 0 : true;
        float w = 1.f;
        if (ratio_mode) w = ((float)a.msum[e] * 1.0f) / (float)p.max_episode_steps;
        const float adv = a.adv[e];
        acc[S_NM] += on ? 1.0 : 0.0;
        const float* lpp = a.lp + e * K;
        const float* olp = a.old + e * K;
        for (int s = 0; s < S; ++s) {
            float lp = 0.f, old = 0.f;
            if constexpr (VEC4) {
                for (int j = 0; j < R; j += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(lpp + s * R + j);
                    const float4 y = *reinterpret_cast<const float4*>(olp + s * R + j);
                    lp = fadd(fadd(fadd(fadd(lp, x.x), x.y), x.z), x.w);
                    old = fadd(fadd(fadd(fadd(old, y.x), y.y), y.z), y.w);
                }
            } else {
                for (int j = 0; j < R; ++j) {
                    lp = fadd(lp, lpp[s * R + j]);
                    old = fadd(old, olp[s * R + j]);
                }
            }
            const float g = actor_elem(p, lp, old, adv, on, w, ratio_mode, acc);
            a.g_lp[e * S + s] = g;
        }
        if (p.has_critic) {
            const float gv = critic_elem(p, a.v[e], a.pv[e], a.ret[e], on, w, ratio_mode, half_delta, acc);
            a.g_v[e] = gv;
        }
    }
    block_sum<NS>(acc, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) a.partials[(size_t)blockIdx.x * NS + k] = acc[k];
    }
}

__global__ __launch_bounds__(256) void ppo_loss_finalize(const double* partials, int nparts, long long n_adv,
                                                         rlx_ppo_loss_params p, int has_mask, int has_msum,
                                                         float* out) {
    __shared__ double s_red[NS * 4];
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] += partials[(size_t)i * NS + k];
    }
    block_sum<NS>(acc, s_red);
    if (threadIdx.x != 0) return;
    finalize_row(p, n_adv, has_mask != 0, has_msum != 0, acc, out);
}

__global__ __launch_bounds__(256) void ppo_loss_bwd_kernel(const float* g_lp, const float* g_v, const float* out,
                                                           const float* grad_out, float* d_lp, float* d_v,
                                                           long long n_adv, int K, int S) {
    const float go = grad_out ? grad_out[0] : 1.f;
    const float sa = go * out[RLX_PPO_ACTOR_GRAD_SCALE], sc = go * out[RLX_PPO_CRITIC_GRAD_SCALE];
    const int R = K / S;
    const long long total = n_adv * K;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long e = i / K;
        const int s = (int)(i - e * K) / R;
        d_lp[i] = sa * g_lp[e * S + s];
    }
    if (d_v) {
        for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_adv; e += stride) d_v[e] = sc * g_v[e];
    }
}

// The Gaussian policy's entropy does not depend on the state: per row it is sum_a (0.5 + 0.5 log 2pi + logstd_a), so
//   loss -= entropy_bonus * masked_mean(entropy)    (embodied_fsdp_actor_worker.py:679-690, reshape_entropy utils.py:384-408)
// is a constant push on the logstd gradient plus one scalar for the metrics.  elem_scale = 1, or 1/A when entropy_type is
// "token_level" and there is no loss mask (the mean then runs over the [mb, A] elements).
__global__ __launch_bounds__(64) void entropy_bonus_kernel(const float* __restrict__ logstd, int n_act,
                                                           float* __restrict__ grad_logstd, float* __restrict__ out_row,
                                                           float bonus, float grad_scale, int has_mask, float elem_scale,
                                                           const float* __restrict__ actor_scale) {
    const int a = threadIdx.x;
    // behind a decoupled rlx_ppo_step the logstd gradient is in sum form: pre-divide by the scale the slab sum applies later
    const float undo = actor_scale != nullptr ? 1.f / actor_scale[0] : 1.f;
    const float on = (!has_mask || out_row[18] > 0.f) ? 1.f : 0.f;  // masked_mean over an all-False mask is the (zero) sum
    float e = 0.f;
    if (a < n_act) {
        e = 0.5f + 0.91893853320467274178f + logf(expf(logstd[a]));  // Normal.entropy() on scale = exp(logstd)
        grad_logstd[a] -= bonus * grad_scale * elem_scale * on * undo;
    }
    e = wave_sum(e);
    if (a == 0) {
        const float ent_loss = e * elem_scale * on;
        out_row[19] = ent_loss;                                  // actor/entropy_loss
        out_row[RLX_PPO_LOSS] -= bonus * ent_loss;
    }
}

constexpr int kMaxLossBlocks = 1024;

int loss_grid(long long n) {
    const long long want = (n + 255) / 256;
    return (int)std::max<long long>(1, std::min<long long>(want, std::min<long long>(kMaxLossBlocks, (long long)num_cu() * 4)));
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_ppo_loss_workspace_bytes(int64_t n_adv) {
    (void)n_adv;
    return (size_t)kMaxLossBlocks * NS * sizeof(double);
}

extern "C" int rlx_ppo_loss_fwd(const float* logprobs, const float* old_logprobs, const float* advantages,
                                const float* values, const float* prev_values, const float* returns,
                                const uint8_t* loss_mask, const int64_t* loss_mask_sum, int64_t n_adv,
                                const rlx_ppo_loss_params* p, float* g_logp, float* g_value, float* out, void* workspace,
                                size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(p != nullptr && out != nullptr, "rlx_ppo_loss_fwd: NULL params/out");
    RLX_REQUIRE(n_adv >= 0, "rlx_ppo_loss_fwd: negative size");
    RLX_REQUIRE(p->raw_per_adv >= 1 && p->sub_per_adv >= 1 && p->raw_per_adv % p->sub_per_adv == 0,
                "rlx_ppo_loss_fwd: raw_per_adv=%d must be a positive multiple of sub_per_adv=%d", p->raw_per_adv,
                p->sub_per_adv);
    RLX_REQUIRE(n_adv == 0 || (logprobs && old_logprobs && advantages && g_logp), "rlx_ppo_loss_fwd: NULL actor argument");
    RLX_REQUIRE(!p->has_critic || n_adv == 0 || (values && prev_values && returns && g_value),
                "rlx_ppo_loss_fwd: has_critic set but a critic tensor is NULL");
    RLX_REQUIRE(!p->use_dual_clip || p->clip_ratio_c > 1.0f, "clip_ratio_c must be greater than 1.0");  // losses.py:262
    RLX_REQUIRE(workspace != nullptr, "rlx_ppo_loss_fwd: NULL workspace");
    if (workspace_bytes < rlx_ppo_loss_workspace_bytes(n_adv)) {
        set_error("rlx_ppo_loss_fwd: workspace too small");
        return RLX_ENOSPC;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    LossArgs a;
    a.lp = logprobs; a.old = old_logprobs; a.adv = advantages; a.v = values; a.pv = prev_values; a.ret = returns;
    a.m = loss_mask; a.msum = loss_mask_sum; a.g_lp = g_logp; a.g_v = g_value;
    a.partials = static_cast<double*>(workspace);
    a.n = n_adv; a.p = *p;
    const int nblk = loss_grid(n_adv);
    const int R = p->raw_per_adv / p->sub_per_adv;
    const bool vec4 = R % 4 == 0 && (reinterpret_cast<uintptr_t>(logprobs) | reinterpret_cast<uintptr_t>(old_logprobs)) % 16 == 0;
    if (vec4) hipLaunchKernelGGL(ppo_loss_fwd_kernel<true>, dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ppo_loss_fwd_kernel<false>, dim3(nblk), dim3(256), 0, s, a);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(ppo_loss_finalize, dim3(1), dim3(256), 0, s, a.partials, nblk, (long long)n_adv, *p,
                       loss_mask != nullptr ? 1 : 0, loss_mask_sum != nullptr ? 1 : 0, out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_ppo_loss_bwd(const float* g_logp, const float* g_value, const float* out, const float* grad_out,
                                float* d_logprobs, float* d_values, int64_t n_adv, int raw_per_adv, int sub_per_adv,
                                rlx_stream_t stream) {
    RLX_REQUIRE(n_adv >= 0 && raw_per_adv >= 1 && sub_per_adv >= 1 && raw_per_adv % sub_per_adv == 0,
                "rlx_ppo_loss_bwd: bad sizes");
    if (n_adv == 0) return RLX_OK;
    RLX_REQUIRE(g_logp && out && d_logprobs, "rlx_ppo_loss_bwd: NULL argument");
    RLX_REQUIRE(d_values == nullptr || g_value != nullptr, "rlx_ppo_loss_bwd: d_values requested without g_value");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ppo_loss_bwd_kernel, dim3(loss_grid(n_adv * raw_per_adv)), dim3(256), 0, s, g_logp, g_value, out,
                       grad_out, d_logprobs, d_values, (long long)n_adv, raw_per_adv, sub_per_adv);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

namespace {
int entropy_bonus_launch(const float* logstd, int n_act, float* grad_logstd, float* out_row, float entropy_bonus, float grad_scale,
                         int has_mask, float elem_scale, const float* actor_scale, rlx_stream_t stream) {
    RLX_REQUIRE(logstd && grad_logstd && out_row && n_act >= 1 && n_act <= 64, "rlx_gaussian_entropy_bonus: bad argument");
    hipLaunchKernelGGL(entropy_bonus_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), logstd, n_act,
                       grad_logstd, out_row, entropy_bonus, grad_scale, has_mask, elem_scale, actor_scale);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
}  // namespace

extern "C" int rlx_gaussian_entropy_bonus(const float* logstd, int n_act, float* grad_logstd, float* out_row,
                                          float entropy_bonus, float grad_scale, int has_mask, float elem_scale,
                                          rlx_stream_t stream) {
    return entropy_bonus_launch(logstd, n_act, grad_logstd, out_row, entropy_bonus, grad_scale, has_mask, elem_scale, nullptr, stream);
}

extern "C" int rlx_gaussian_entropy_bonus_deferred(const float* logstd, int n_act, float* grad_logstd, float* out_row,
                                                   float entropy_bonus, float grad_scale, int has_mask, float elem_scale,
                                                   const float* actor_scale, rlx_stream_t stream) {
    RLX_REQUIRE(actor_scale != nullptr, "rlx_gaussian_entropy_bonus_deferred: NULL actor_scale");
    return entropy_bonus_launch(logstd, n_act, grad_logstd, out_row, entropy_bonus, grad_scale, has_mask, elem_scale, actor_scale, stream);
}
