// ppo_loss_math.h -- per-element arithmetic of the PPO actor / critic loss and its finalisation, shared by the
// stand-alone loss kernels (ppo_loss.hip) and the fused optimizer-step kernels (ppo_step.hip).
//   compute_ppo_actor_loss   rlinf/algorithms/losses.py:170-312
//   compute_ppo_critic_loss  rlinf/algorithms/losses.py:315-380  (+ huber_loss rlinf/algorithms/utils.py:20-23,
//   masked_mean / masked_mean_ratio rlinf/utils/utils.py:323-356, explained-variance stats metric_utils.py:232-258)
// The per-element derivative is computed together with the value, exactly as autograd would for the reference
// graph, including its tie rules: torch.max / torch.min send half the gradient to each argument on ties, clamp
// passes gradient on its closed interval, where() routes to the taken branch.
#pragma once

#include "rlx_common.h"

namespace rlx {
namespace loss {

constexpr int NS = 16;  // reduction slots
enum { S_NM = 0, S_LOSS, S_ABS, S_RATIO, S_RABS, S_CLIPPED, S_DUAL, S_KL, S_CLIPFRAC, S_VLOSS, S_VIND,
       S_EVN, S_EVR, S_EVRR, S_EVE, S_EVEE };

__device__ __forceinline__ float huber(float e, float delta, float half_delta) {
    const float ae = fabsf(e);
    return ae < delta ? fmul(0.5f, fmul(e, e)) : fmul(delta, fsub(ae, half_delta));
}
__device__ __forceinline__ float huber_grad(float e, float delta) {
    const float ae = fabsf(e);
    return ae < delta ? e : (e > 0.f ? delta : (e < 0.f ? -delta : 0.f));
}
__device__ __forceinline__ float tie_weight_gt(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float tie_weight_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// One loss element of the actor: lp / old = summed log-probs, adv = its advantage, `on` = loss-mask bit,
// w = loss_mask_sum / max_episode_steps (ratio_mode only).  Accumulates the metric sums and returns
// d(sum-form loss)/d(lp) (to be scaled by 1/denominator).
__device__ __forceinline__ float actor_elem(const rlx_ppo_loss_params& p, float lp, float old, float adv, bool on, float w,
                                            bool ratio_mode, double (&acc)[NS]) {
    const float mf = on ? 1.f : 0.f;
    const float nadv = -adv;
    float lr = fsub(lp, old);
    float cg = 1.f;  // gradient of the two log-ratio clamps
    if (p.use_clip_log_ratio_min) {
        if (!(lr >= p.clip_log_ratio_min)) cg = 0.f;
        lr = fmaxf(lr, p.clip_log_ratio_min);
    }
    if (p.use_clip_log_ratio_max) {
        if (!(lr <= p.clip_log_ratio_max)) cg = 0.f;
        lr = fminf(lr, p.clip_log_ratio_max);
    }
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
    const float contrib = ratio_mode ? fmul(pl / w, mf) : fmul(pl, mf);
    const float contrib_abs = ratio_mode ? fmul(fabsf(pl) / w, mf) : fmul(fabsf(pl), mf);
    acc[S_LOSS] += (double)contrib;
    acc[S_ABS] += (double)contrib_abs;
    acc[S_RATIO] += (double)fmul(ratio, mf);
    acc[S_RABS] += (double)fmul(fabsf(fsub(ratio, 1.f)), mf);
    acc[S_CLIPPED] += (double)fmul(clipped, mf);
    acc[S_DUAL] += (dual && on) ? (double)ratio : 0.0;
    acc[S_KL] += on ? (double)lr : 0.0;
    acc[S_CLIPFRAC] += (pl1 < pl2 && on) ? 1.0 : 0.0;
    float g = p.critic_warmup ? 0.f : dpl * ratio * cg * mf;  // ratio == d exp(lr)/d lr, 0 when masked
    if (ratio_mode) g = g / w;
    return g;
}

// One element of the critic: returns d(sum-form value loss)/d(v).
__device__ __forceinline__ float critic_elem(const rlx_ppo_loss_params& p, float v, float pv, float ret, bool on, float w,
                                             bool ratio_mode, float half_delta, double (&acc)[NS]) {
    const float mf = on ? 1.f : 0.f;
    const float diff = fsub(v, pv);
    const float cl = fminf(fmaxf(diff, -p.value_clip), p.value_clip);
    const float vclip = fadd(pv, cl);
    const float e1 = fsub(ret, v), e2 = fsub(ret, vclip);
    const float h1 = huber(e1, p.huber_delta, half_delta), h2 = huber(e2, p.huber_delta, half_delta);
    const float h = fmaxf(h1, h2);
    acc[S_VLOSS] += (double)(ratio_mode ? fmul(h / w, mf) : fmul(h, mf));
    acc[S_VIND] += fabsf(fsub(vclip, pv)) > p.value_clip ? 1.0 : 0.0;
    if (on) {
        acc[S_EVN] += 1.0;
        acc[S_EVR] += (double)ret;
        acc[S_EVRR] += (double)fmul(ret, ret);
        acc[S_EVE] += (double)e1;
        acc[S_EVEE] += (double)fmul(e1, e1);
    }
    const float wh1 = tie_weight_gt(h1, h2);
    const float pass = (diff >= -p.value_clip && diff <= p.value_clip) ? 1.f : 0.f;
    float gv = -(wh1 * huber_grad(e1, p.huber_delta) + (1.f - wh1) * huber_grad(e2, p.huber_delta) * pass) * mf;
    if (ratio_mode) gv = gv / w;
    return gv;
}

// ---- decoupled (asynchronous) PPO actor, rlinf/algorithms/losses.py:27-167 -----------------------------------------------
// The clipped surrogate is taken against the PROXIMAL policy (given, or the behaviour policy itself, or interpolated between
// behaviour and current by the version distance of the sample: losses.py:72-89) and every term is importance-weighted by
// exp(prox - behaviour), optionally masked where that weight exceeds behave_weight_threshold.  The proximal log-probs are
// detached in the reference: no gradient flows through the interpolation.
// Reduction slots of the decoupled actor: the classic actor's slots 1-8 re-used, plus the version sum (slot `ver_slot`: 16 in
// the stand-alone kernel's 17-slot rows; the fused step parks it in slot S_VLOSS of the ACTOR network's partial row, which
// the critic's sums -- they live in the value network's row -- leave free).
enum { D_LOSS = S_LOSS, D_BM = S_ABS, D_PR = S_RATIO, D_CPR = S_RABS, D_CLIPF = S_CLIPPED, D_DUALF = S_DUAL, D_PKL = S_KL,
       D_BKL = S_CLIPFRAC };
struct DecoupledMode {
    int mode;              // rlx_proximal_mode
    int use_threshold;
    float threshold;
    float v_theta;         // current_version
};
// lp / old / px_given = summed log-probs of the element's slice (px_given only for RLX_PROX_GIVEN), vb = the slice's first
// behaviour version (0 without versions).  Returns d(sum-form loss)/d(lp).
template <int N>
__device__ __forceinline__ float decoupled_actor_elem(const rlx_ppo_loss_params& p, const DecoupledMode& dm, float lp, float old,
                                                      float px_given, float vb, float adv, bool on, float w, bool ratio_mode,
                                                      double (&acc)[N], int ver_slot) {
    const float mf = on ? 1.f : 0.f;
    const float nadv = -adv;
    float px = px_given;
    if (dm.mode == RLX_PROX_FROM_VERSIONS) {
        const float v_prox = dm.v_theta - 1.0f;
        const float diff = fsub(dm.v_theta, vb), gap = fsub(v_prox, vb);
        float alpha = (diff > 0.f && vb >= 0.f) ? gap / diff : 0.f;
        alpha = fminf(fmaxf(alpha, 0.f), 1.f);
        px = fadd(old, fmul(alpha, fsub(lp, old)));
    } else if (dm.mode == RLX_PROX_IS_OLD) {
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
    const bool bm = on && (!dm.use_threshold || bw <= dm.threshold);
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
    acc[ver_slot] += on ? (double)vb : 0.0;
    float g = p.critic_warmup ? 0.f : dpl * ratio * bw * bmf;
    if (ratio_mode) g = g / w;
    return g;
}

struct Denoms {
    double actor, critic, metric, count;
};
// masked_mean: sum/sum(mask); all-False mask -> plain sum (which is 0); no mask -> mean
__device__ __forceinline__ Denoms denominators(const rlx_ppo_loss_params& p, long long n_adv, double nm, bool has_mask,
                                               bool has_msum) {
    const double L = (double)n_adv * p.sub_per_adv;  // loss elements
    const double Lc = (double)n_adv;
    const bool ratio_mode = p.max_episode_steps > 0 && has_mask && has_msum;
    Denoms d;
    d.actor = ratio_mode ? L : (has_mask ? (nm > 0 ? nm : 1.0) : L);
    d.critic = ratio_mode ? Lc : (has_mask ? (nm > 0 ? nm : 1.0) : Lc);
    d.metric = has_mask ? (nm > 0 ? nm * (p.metric_unbroadcast ? 1 : p.sub_per_adv) : 1.0) : L;
    d.count = has_mask ? (nm > 0 ? nm : 1.0) : L;  // loss_mask.count_nonzero() or 1
    return d;
}

// acc = the NS sums over the whole micro-batch -> the rlx_ppo_out row
__device__ __forceinline__ void finalize_row(const rlx_ppo_loss_params& p, long long n_adv, bool has_mask, bool has_msum,
                                             const double (&acc)[NS], float* out) {
    const double nm = acc[S_NM];
    const Denoms d = denominators(p, n_adv, nm, has_mask, has_msum);
    const double Lc = (double)n_adv;
    const float policy_loss = p.critic_warmup ? 0.f : (float)(acc[S_LOSS] / d.actor);
    const float value_loss = p.has_critic ? (float)(acc[S_VLOSS] / d.critic) : 0.f;
    out[RLX_PPO_LOSS] = policy_loss + value_loss;
    out[RLX_PPO_POLICY_LOSS] = policy_loss;
    out[RLX_PPO_POLICY_LOSS_ABS] = (float)(acc[S_ABS] / d.actor);
    out[RLX_PPO_RATIO] = (float)(acc[S_RATIO] / d.metric);
    out[RLX_PPO_RATIO_ABS] = (float)(acc[S_RABS] / d.metric);
    out[RLX_PPO_CLIPPED_RATIO] = (float)(acc[S_CLIPPED] / d.metric);
    out[RLX_PPO_DUAL_CLIPPED_RATIO] = (float)(acc[S_DUAL] / d.metric);
    out[RLX_PPO_APPROX_KL] = (float)(-acc[S_KL] / d.count);
    out[RLX_PPO_CLIP_FRACTION] = (float)(acc[S_CLIPFRAC] / d.count);
    out[RLX_PPO_VALUE_LOSS] = value_loss;
    out[RLX_PPO_VALUE_CLIP_RATIO] = p.has_critic ? (float)(acc[S_VIND] / Lc) : 0.f;
    out[RLX_PPO_EV_COUNT] = (float)acc[S_EVN];
    out[RLX_PPO_EV_RETURNS_SUM] = (float)acc[S_EVR];
    out[RLX_PPO_EV_RETURNS_SQ_SUM] = (float)acc[S_EVRR];
    out[RLX_PPO_EV_ERRORS_SUM] = (float)acc[S_EVE];
    out[RLX_PPO_EV_ERRORS_SQ_SUM] = (float)acc[S_EVEE];
    out[RLX_PPO_ACTOR_GRAD_SCALE] = (float)(1.0 / d.actor);
    out[RLX_PPO_CRITIC_GRAD_SCALE] = (float)(1.0 / d.critic);
    out[18] = (float)nm;
    out[19] = 0.f;
}

// the decoupled row (rlx_dppo_out): acc = the classic 16 sums with the decoupled actor's in slots 1-8, ver_sum = the version sum
__device__ __forceinline__ void finalize_row_decoupled(const rlx_ppo_loss_params& p, bool use_threshold, long long n_adv, bool has_mask,
                                                       bool has_msum, const double (&acc)[NS], double ver_sum, float* out) {
    const double nm = acc[S_NM];
    const Denoms d = denominators(p, n_adv, nm, has_mask, has_msum);
    const bool ratio_mode = p.max_episode_steps > 0 && has_mask && has_msum;
    const double L = (double)n_adv * p.sub_per_adv, Lc = (double)n_adv;
    // loss_mask.count_nonzero() or 1 (the UNbroadcast mask); behav_mask.count_nonzero() or 1 (broadcast to the loss shape)
    const double n_valid = has_mask ? (nm > 0 ? nm : 1.0) : L;
    // with a threshold behav_mask is built at the loss shape; without one it IS loss_mask, whose sum the reference takes
    // before broadcasting (token_level: nm, not nm * action_dim)
    const double behav_cnt = use_threshold ? acc[D_BM] : (has_mask ? nm : L);
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
    out[RLX_DPPO_AVERAGE_VERSION] = (float)(ver_sum / (has_mask ? (nm > 0 ? nm : 1.0) : L));
}

}  // namespace loss
}  // namespace rlx
