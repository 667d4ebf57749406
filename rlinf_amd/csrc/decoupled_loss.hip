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

// reduction slots of the decoupled actor (the critic keeps S_VLOSS.. of ppo_loss_math.h, S_NM is shared)
enum { D_LOSS = S_LOSS, D_BM = S_ABS, D_PR = S_RATIO, D_CPR = S_RABS, D_CLIPF = S_CLIPPED, D_DUALF = S_DUAL, D_PKL = S_KL,
       D_BKL = S_CLIPFRAC };
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
    const int mode = a.p.proximal_mode;
    const float v_theta = a.p.current_version, v_prox = a.p.current_version - 1.0f;
    double acc[NSD];
#pragma unroll
    for (int k = 0; k < NSD; ++k) acc[k] = 0.0;
    double critic_acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) critic_acc[k] = 0.0;

    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < a.n; e += stride) {
        const bool on = a.m ? a.m[e] != 0 : true;
        const float mf = on ? 1.f : 0.f;
        float w = 1.f;
        if (ratio_mode) w = ((float)a.msum[e] * 1.0f) / (float)p.max_episode_steps;
        const float adv = a.adv[e], nadv = -adv;
        acc[S_NM] += on ? 1.0 : 0.0;
        for (int s = 0; s < S; ++s) {
            float lp = 0.f, old = 0.f, px = 0.f;
            for (int j = 0; j < R; ++j) {
                lp = fadd(lp, a.lp[e * K + s * R + j]);
                old = fadd(old, a.old[e * K + s * R + j]);
                if (mode == RLX_PROX_GIVEN) px = fadd(px, a.prox[e * K + s * R + j]);
            }
            const float vb = a.versions ? a.versions[e * K + s * R] : 0.f;  // the slice's first entry (utils.py:337,348)
            if (mode == RLX_PROX_FROM_VERSIONS) {
                const float diff = fsub(v_theta, vb), gap = fsub(v_prox, vb);
                float alpha = (diff > 0.f && vb >= 0.f) ? gap / diff : 0.f;
                alpha = fminf(fmaxf(alpha, 0.f), 1.f);
                px = fadd(old, fmul(alpha, fsub(lp, old)));
            } else if (mode == RLX_PROX_IS_OLD) {
                px = old;
            }
            const float lr = fsub(lp, px);
            const float ratio = on ? expf(lr) : 0.f;
            const float clipped = fminf(fmaxf(ratio, p.ratio_lo), p.ratio_hi);
            const float pl1 = fmul(nadv, ratio), pl2 = fmul(nadv, clipped);
            float pl = fmaxf(pl1, pl2);
            const float w1 = tie_weight_gt(pl1, pl2);
            const float in_rng = (ratio >= p.ratio_lo && ratio <= p.ratio_hi) ? 1.f : 0.f;
            float dpl = nadv * (w1 + (1.f - w1) * in_rng);
            bool dual = false;
            if (p.use_dual_clip) {
                const float sgn = adv > 0.f ? 1.f : (adv < 0.f ? -1.f : 0.f);
                const float pl3 = fmul(fmul(sgn, p.clip_ratio_c), adv);
                dual = pl3 < pl;
                dpl *= tie_weight_lt(pl, pl3);
                pl = fminf(pl, pl3);
            }
            const float bw = expf(fsub(px, old));
            const bool bm = on && (!a.p.use_behave_threshold || bw <= a.p.behave_weight_threshold);
            const float bmf = bm ? 1.f : 0.f;
            const float weighted = fmul(pl, bw);
            acc[D_LOSS] += (double)(ratio_mode ? fmul(weighted / w, bmf) : fmul(weighted, bmf));
            acc[D_BM] += bm ? 1.0 : 0.0;
            acc[D_PR] += (double)fmul(ratio, mf);
            acc[D_CPR] += (double)fmul(clipped, mf);
            acc[D_CLIPF] += (pl1 < pl2 && on) ? 1.0 : 0.0;
            acc[D_DUALF] += (dual && on) ? 1.0 : 0.0;
            acc[D_PKL] += on ? (double)lr : 0.0;
            acc[D_BKL] += bm ? (double)fsub(px, old) : 0.0;
            acc[D_VER] += on ? (double)vb : 0.0;
            float g = p.critic_warmup ? 0.f : dpl * ratio * bw * bmf;
            if (ratio_mode) g = g / w;
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
    const rlx_ppo_loss_params& p = dp.ppo;
    const double nm = acc[S_NM];
    const Denoms d = denominators(p, n_adv, nm, has_mask != 0, has_msum != 0);
    const bool ratio_mode = p.max_episode_steps > 0 && has_mask && has_msum;
    const double L = (double)n_adv * p.sub_per_adv, Lc = (double)n_adv;
    // loss_mask.count_nonzero() or 1 (the UNbroadcast mask); behav_mask.count_nonzero() or 1 (broadcast to the loss shape)
    const double n_valid = has_mask ? (nm > 0 ? nm : 1.0) : L;
    // with a threshold behav_mask is built at the loss shape; without one it IS loss_mask, whose sum the reference takes
    // before broadcasting (token_level: nm, not nm * action_dim)
    const double behav_cnt = dp.use_behave_threshold ? acc[D_BM] : (has_mask ? nm : L);
    const double n_behav = behav_cnt > 0 ? behav_cnt : 1.0;
    const double actor_den = ratio_mode ? L : n_behav;  // masked_mean over behav_mask; all-False -> the (zero) sum
    const float policy_loss = p.critic_warmup ? 0.f : (float)(acc[D_LOSS] / actor_den);
    const float value_loss = p.has_critic ? (float)(acc[S_VLOSS] / d.critic) : 0.f;
    out[RLX_DPPO_LOSS] = policy_loss + value_loss;
    out[RLX_DPPO_POLICY_LOSS] = policy_loss;
    out[RLX_DPPO_PROXIMAL_RATIO] = (float)(acc[D_PR] / n_valid);  // masked_mean with the unbroadcast mask (:147-150)
    out[RLX_DPPO_CLIPPED_PROXIMAL_RATIO] = (float)(acc[D_CPR] / n_valid);
    out[RLX_DPPO_DUAL_CLIP_FRACTION] = (float)(acc[D_DUALF] / n_valid);
    out[RLX_DPPO_BEHAV_CLIP_FRACTION] = (float)(1.0 - n_behav / n_valid);
    out[RLX_DPPO_PROXIMAL_APPROX_KL] = (float)(-acc[D_PKL] / n_valid);
    out[RLX_DPPO_BEHAV_APPROX_KL] = (float)(-acc[D_BKL] / n_behav);
    out[RLX_DPPO_CLIP_FRACTION] = (float)(acc[D_CLIPF] / n_valid);
    out[RLX_PPO_VALUE_LOSS] = value_loss;
    out[RLX_PPO_VALUE_CLIP_RATIO] = p.has_critic ? (float)(acc[S_VIND] / Lc) : 0.f;
    out[RLX_PPO_EV_COUNT] = (float)acc[S_EVN];
    out[RLX_PPO_EV_RETURNS_SUM] = (float)acc[S_EVR];
    out[RLX_PPO_EV_RETURNS_SQ_SUM] = (float)acc[S_EVRR];
    out[RLX_PPO_EV_ERRORS_SUM] = (float)acc[S_EVE];
    out[RLX_PPO_EV_ERRORS_SQ_SUM] = (float)acc[S_EVEE];
    out[RLX_PPO_ACTOR_GRAD_SCALE] = (float)(1.0 / actor_den);
    out[RLX_PPO_CRITIC_GRAD_SCALE] = (float)(1.0 / d.critic);
    out[18] = (float)nm;
    // versions[loss_mask].mean(): only defined by the reference when versions and loss_mask share a shape (sub == 1)
    out[RLX_DPPO_AVERAGE_VERSION] = (float)(acc[D_VER] / (has_mask ? (nm > 0 ? nm : 1.0) : L));
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
