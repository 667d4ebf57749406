// decoupled_loss.hip -- decoupled (async) PPO actor loss + the PPO critic loss, forward; gfx950.
//
// Replaces, after preprocess_loss_inputs (rlinf/algorithms/utils.py:280-376, incl. its proximal_logprobs / versions
// branches :310-352):
//   compute_decoupled_ppo_actor_loss        rlinf/algorithms/losses.py:27-167
//   compute_decoupled_ppo_actor_critic_loss rlinf/algorithms/losses.py:383-393 (critic: losses.py:315-380)
// Same structure as ppo_loss.hip -- one streaming pass + a one-block finalisation, everything on the device -- and the
// same backward (rlx_ppo_loss_bwd expands g_logp / g_value with the scales in out[16], out[17]).
//
// The clipped surrogate is taken against the PROXIMAL policy (given, or the behaviour policy itself, or interpolated
// between behaviour and current by the version distance of each sample), and every term is importance-weighted by
// exp(prox - behaviour), optionally masked where that weight exceeds behave_weight_threshold.  The proximal log-probs
// are detached in the reference: no gradient flows through the interpolation.

#include <algorithm>

#include "ppo_loss_math.h"

namespace rlx {
namespace {

using namespace loss;

// reduction slots: ppo_loss_math.h (D_*), S_NM shared, the critic keeps S_VLOSS..; one more for the version sum
constexpr int NSD = NS + 1;  // + the version sum
constexpr int D_VER = NS;

struct DLossArgs {
    const float *lp, *old, *prox, *versions, *adv, *v, *pv, *ret;
    const uint8_t* m;
    const int64_t* msum;
    float *g_lp, *g_v;
    double* partials;
    long long n;
    rlx_decoupled_loss_params p;
};

__global__ __launch_bounds__(256) void decoupled_loss_fwd_kernel(DLossArgs a) {
    __shared__ double s_red[NSD * 4];
    const rlx_ppo_loss_params& p = a.p.ppo;
    const int K = p.raw_per_adv, S = p.sub_per_adv, R = K / S;
    const bool ratio_mode = p.max_episode_steps > 0 && a.m != nullptr && a.msum != nullptr;
    const float half_delta = (float)(0.5 * (double)p.huber_delta);
    const DecoupledMode dm{a.p.proximal_mode, a.p.use_behave_threshold, a.p.behave_weight_threshold, a.p.current_version};
    const int mode = dm.mode;
    double acc[NSD];
#pragma unroll
    for (int k = 0; k < NSD; ++k) acc[k] = 0.0;
    double critic_acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) critic_acc[k] = 0.0;

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < a.n; e += stride) {
        const bool on = a.m ? a.m[e] != 0 : true;
        float w = 1.f;
        if (ratio_mode) w = ((float)a.msum[e] * 1.0f) / (float)p.max_episode_steps;
        const float adv = a.adv[e];
        acc[S_NM] += on ? 1.0 : 0.0;
        for (int s = 0; s < S; ++s) {
            float lp = 0.f, old = 0.f, px = 0.f;
            for (int j = 0; j < R; ++j) {
                lp = fadd(lp, a.lp[e * K + s * R + j]);
                old = fadd(old, a.old[e * K + s * R + j]);
                if (mode == RLX_PROX_GIVEN) px = fadd(px, a.prox[e * K + s * R + j]);
            }
            const float vb = a.versions ? a.versions[e * K + s * R] : 0.f;  // the slice's first entry (utils.py:337,348)
            const float g = decoupled_actor_elem(p, dm, lp, old, px, vb, adv, on, w, ratio_mode, acc, D_VER);
            a.g_lp[e * S + s] = g;
        }
        if (p.has_critic) a.g_v[e] = critic_elem(p, a.v[e], a.pv[e], a.ret[e], on, w, ratio_mode, half_delta, critic_acc);
    }
#pragma unroll
    for (int k = S_VLOSS; k < NS; ++k) acc[k] = critic_acc[k];
    block_sum<NSD>(acc, s_red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NSD; ++k) a.partials[(size_t)blockIdx.x * NSD + k] = acc[k];
    }
}

__global__ __launch_bounds__(256) void decoupled_loss_finalize(const double* partials, int nparts, long long n_adv,
                                                               rlx_decoupled_loss_params dp, int has_mask, int has_msum,
                                                               float* out) {
    __shared__ double s_red[NSD * 4];
    double acc[NSD];
#pragma unroll
    for (int k = 0; k < NSD; ++k) acc[k] = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
#pragma unroll
        for (int k = 0; k < NSD; ++k) acc[k] += partials[(size_t)i * NSD + k];
    }
    block_sum<NSD>(acc, s_red);
    if (threadIdx.x != 0) return;
    double main[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) main[k] = acc[k];
    finalize_row_decoupled(dp.ppo, dp.use_behave_threshold != 0, n_adv, has_mask != 0, has_msum != 0, main, acc[D_VER], out);
}

constexpr int kMaxBlocks = 1024;
int dloss_grid(long long n) {
    const long long want = (n + 255) / 256;
    return (int)std::max<long long>(1, std::min<long long>(want, std::min<long long>(kMaxBlocks, (long long)num_cu() * 4)));
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_decoupled_loss_workspace_bytes(int64_t n_adv) {
    (void)n_adv;
    return (size_t)kMaxBlocks * NSD * sizeof(double);
}

extern "C" int rlx_decoupled_loss_fwd(const float* logprobs, const float* old_logprobs, const float* proximal_logprobs,
                                      const float* versions, const float* advantages, const float* values,
                                      const float* prev_values, const float* returns, const uint8_t* loss_mask,
                                      const int64_t* loss_mask_sum, int64_t n_adv, const rlx_decoupled_loss_params* dp,
                                      float* g_logp, float* g_value, float* out, void* workspace, size_t workspace_bytes,
                                      rlx_stream_t stream) {
    RLX_REQUIRE(dp != nullptr && out != nullptr, "rlx_decoupled_loss_fwd: NULL params/out");
    const rlx_ppo_loss_params* p = &dp->ppo;
    RLX_REQUIRE(n_adv >= 0, "rlx_decoupled_loss_fwd: negative size");
    RLX_REQUIRE(p->raw_per_adv >= 1 && p->sub_per_adv >= 1 && p->raw_per_adv % p->sub_per_adv == 0,
                "rlx_decoupled_loss_fwd: raw_per_adv=%d must be a positive multiple of sub_per_adv=%d", p->raw_per_adv,
                p->sub_per_adv);
    RLX_REQUIRE(n_adv == 0 || (logprobs && old_logprobs && advantages && g_logp), "rlx_decoupled_loss_fwd: NULL actor argument");
    RLX_REQUIRE(dp->proximal_mode >= RLX_PROX_GIVEN && dp->proximal_mode <= RLX_PROX_FROM_VERSIONS,
                "rlx_decoupled_loss_fwd: unknown proximal_mode %d", dp->proximal_mode);
    RLX_REQUIRE(dp->proximal_mode != RLX_PROX_GIVEN || n_adv == 0 || proximal_logprobs,
                "rlx_decoupled_loss_fwd: proximal_mode GIVEN without proximal_logprobs");
    RLX_REQUIRE(dp->proximal_mode != RLX_PROX_FROM_VERSIONS || n_adv == 0 || versions,
                "rlx_decoupled_loss_fwd: proximal_mode FROM_VERSIONS without versions");
    RLX_REQUIRE(!p->has_critic || n_adv == 0 || (values && prev_values && returns && g_value),
                "rlx_decoupled_loss_fwd: has_critic set but a critic tensor is NULL");
    RLX_REQUIRE(!p->use_dual_clip || p->clip_ratio_c > 1.0f, "clip_ratio_c must be greater than 1.0");  // losses.py:108
    RLX_REQUIRE(workspace != nullptr, "rlx_decoupled_loss_fwd: NULL workspace");
    if (workspace_bytes < rlx_decoupled_loss_workspace_bytes(n_adv)) {
        set_error("rlx_decoupled_loss_fwd: workspace too small");
        return RLX_ENOSPC;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    DLossArgs a;
    a.lp = logprobs, a.old = old_logprobs, a.prox = proximal_logprobs, a.versions = versions, a.adv = advantages;
    a.v = values, a.pv = prev_values, a.ret = returns, a.m = loss_mask, a.msum = loss_mask_sum;
    a.g_lp = g_logp, a.g_v = g_value, a.partials = static_cast<double*>(workspace), a.n = n_adv, a.p = *dp;
    const int nblk = dloss_grid(n_adv);
    hipLaunchKernelGGL(decoupled_loss_fwd_kernel, dim3(nblk), dim3(256), 0, s, a);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(decoupled_loss_finalize, dim3(1), dim3(256), 0, s, a.partials, nblk, (long long)n_adv, *dp,
                       loss_mask != nullptr ? 1 : 0, loss_mask_sum != nullptr ? 1 : 0, out);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
