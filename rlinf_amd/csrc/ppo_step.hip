// ppo_step.hip -- the two hot launches of the actor-learner loop, gfx950:
//
//   rlx_mlp_rollout_step   ONE launch per rollout step: obs-preprocess -> actor + value MLPs -> Gaussian sample,
//                          log-prob, value, written straight into the trajectory-buffer rows; value-only jobs
//                          (bootstrap value of the previous step's terminal observations folded into that step's
//                          reward row; the closing value row) ride in the same grid.
//                          replaces MLPPolicy.predict_action_batch (mlp_policy.py:295-320),
//                          MultiStepRolloutWorker.get_bootstrap_values (huggingface_worker.py:612-627) and
//                          EnvWorker.compute_bootstrap_rewards (env_worker.py:718-758).
//   rlx_ppo_step           one optimizer step's forward + loss + backward in TWO launches:
//     (1) ppo_step_fused   per 32-row tile and network: training forward (mlp_policy.py:202-236), PPO actor / critic
//                          loss element math + metric sums (losses.py:170-380), loss backward, backward-data chain.
//                          Everything a row needs is row-local once the mean's denominator is known (the kernel
//                          counts the loss mask itself), so activations never leave the CU between layers: the tile
//                          lives in one LDS slab that is overwritten in place, the tanh derivative factors stay in
//                          registers (96 VGPRs) from the forward sweep to the backward sweep.
//     (2) ppo_step_dw      weight gradients dW_l = dZ_l^T H_{l-1} as split-K (over batch rows) 128x128 MFMA tiles into
//                          per-slab gradient buffers, bias gradients as column sums, head gradients folded in from
//                          the per-tile partials of (1), and the metric row finalised by the last block.
//
// Dense layers run on v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate = an fmaf chain; 32-cycle issue):
// a wave owns RT x CT tiles of 16x16 and feeds 4 MFMAs per operand from ONE ds_read_b128 (lane l holds k-block
// l>>4; the k permutation is the same for A and B).  Weights stream L2 -> registers -> LDS in double-buffered
// 16-k chunks, one barrier per chunk; two workgroups per CU keep each SIMD's matrix pipe busy while its other wave
// sits in an epilogue or a barrier.

#include <stdlib.h>

#include <algorithm>

#include "ppo_step_common.h"

namespace rlx {
namespace {

using namespace loss;
using namespace step;

__global__ __launch_bounds__(256) void pack_tiles_kernel(const float* __restrict__ params, rlx_mlp_layout lay,
                                                         float* __restrict__ tiles) {
    const size_t total = 2 * Tiles::per_net();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / Tiles::per_net());
        size_t r = i - y * Tiles::per_net();
        int m = 0, K = Tiles::K1P;
        if (r >= (size_t)HID * Tiles::K1P) {
            r -= (size_t)HID * Tiles::K1P;
            m = 1 + (int)(r / ((size_t)HID * HID));
            r %= (size_t)HID * HID;
            K = HID;
        }
        // r = ((nb * NIT + it) * 2 + h) * 256 + l * 4 + j
        const int j = (int)(r & 3), l = (int)((r >> 2) & 63), h = (int)((r >> 8) & 1);
        const int q = (int)(r >> 9), nit = K / 32, it = q % nit, nb = q / nit;
        const int n = nb * 16 + (l & 15), k = it * 32 + 16 * h + 4 * (l >> 4) + j;
        float v;
        if (m == 0) v = k < lay.obs_dim ? params[lay.off_w[y][0] + (size_t)n * lay.obs_dim + k] : 0.f;
        else if (m <= 2) v = params[lay.off_w[y][m] + (size_t)n * HID + k];           // W_m[n][k]
        else v = params[lay.off_w[y][m - 2] + (size_t)k * HID + n];                    // (W_{m-2})^T[n][k] = W[k][n]
        tiles[i] = v;
    }
}

template <int RT, int CT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[RT][CT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int RT, int CT>
__device__ __forceinline__ void mfma_step(const f32x4 (&a)[RT], const f32x4 (&b)[CT], f32x4 (&acc)[RT][CT]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][q], b[ct][q], acc[rt][ct], 0, 0, 0);
}


// acc = X[0:BM, 0:32*nit] . W^T for one weight matrix given as fragment tiles (struct Tiles).
// Barrier-free K loop: every wave owns 16*CT output columns, so the weight tiles it needs are its own -- they go
// L2 -> registers directly (1 KiB coalesced per load instruction), through a ring of PD register stages that keeps PD
// 32-k iterations of weights in flight.  Only the A operand (the activation slab, read-only during the GEMM) comes
// from LDS.  The waves of a workgroup never wait for each other inside the loop; one barrier at the end lets the
// epilogue overwrite the slab.
// (History, measured on MI355X: weight chunks staged through LDS with a barrier per chunk left the matrix pipe 36 % busy;
//  row-major weights read as 16 rows x 64 B per instruction cost 75 of 122 us in load issue; a "k < K ? load : 0"
//  tail select made hipcc serialise 64 L2 round trips.)
template <int RT, int NW, int PD, int ABL>
struct RowGemm {
    typedef Geo<RT, NW> G;
    static constexpr int CT = G::CT, KI = 32, MAXIT = HID / KI;
    f32x4 bq[PD][2][CT];
    const float* wbase;
    int nit;

    // (Rotating the k-tile order per workgroup, to spread identical requests over the L2 channels, was measured: no gain,
    //  and it makes results depend on the block index -- dropped.)
    __device__ __forceinline__ void gload(int it, f32x4 (&b)[2][CT]) {
        const int itx = it;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if constexpr (ABL & ABL_NO_WLOAD) b[h][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
                else b[h][ct] = *reinterpret_cast<const f32x4*>(wbase + ((size_t)(ct * nit + itx) * 2 + h) * 256);
            }
    }
    // Issue the first PD iterations of weight loads for matrix P.  Call it BEFORE the epilogue that produces the GEMM's
    // input: the weights do not depend on it, so their L2 / Infinity-Cache latency hides behind the tanh sweep.
    __device__ __forceinline__ void prefetch(const float* __restrict__ P, int nit_) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        nit = nit_;
        // tile (nb, it, h) sits at ((nb * nit + it) * 2 + h) * 256 floats; this lane's float4 at + lane * 4
        wbase = P + ((size_t)(wave * CT) * nit * 2) * 256 + lane * 4;
#pragma unroll
        for (int d = 0; d < PD; ++d)
            if (d < nit) gload(d, bq[d]);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch loads up here: hipcc otherwise sinks them next to their uses
    }
    __device__ __forceinline__ void run(const float* X, f32x4 (&acc)[RT][CT]) {
        const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
        zero_acc(acc);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if (it >= nit) break;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 a[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    a[rt] = *reinterpret_cast<const f32x4*>(X + (rt * 16 + r16) * XS + it * KI + 16 * h + 4 * kq);
                if constexpr (ABL & ABL_NO_MFMA) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] += a[rt] * bq[it % PD][h][ct];
                } else {
                    mfma_step(a, bq[it % PD][h], acc);
                }
            }
            if (it + PD < nit) {
                gload(it + PD, bq[it % PD]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();
    }
};

// coalesced copy of the slab's first 256 columns to a row-major [M][256] global array (1 KiB per row)
template <int RT, int NW, int ABL = 0>
__device__ __forceinline__ void flush_rows(const float* X, float* __restrict__ dst, long long m0, long long M) {
    typedef Geo<RT, NW> G;
    if constexpr (ABL & ABL_NO_FLUSH) return;
    for (int f = threadIdx.x; f < G::BM * 64; f += G::NT) {
        const int row = f >> 6, c4 = (f & 63) * 4;
        if (m0 + row < M) *reinterpret_cast<f32x4*>(dst + (size_t)(m0 + row) * HID + c4) = *reinterpret_cast<const f32x4*>(X + row * XS + c4);
    }
}

// obs-preprocess: the state rows themselves (mlp_policy.py:122-124); zero-pad the k tail and the rows past M.
// states_copy: optional second destination (forward_inputs.states row of the trajectory buffer).
template <int RT, int NW>
__device__ __forceinline__ void load_states(const float* __restrict__ states, float* __restrict__ states_copy, int D,
                                            long long m0, long long M, float* X) {
    typedef Geo<RT, NW> G;
    const int kp = round_up(D, KPAD);
    for (int i = threadIdx.x; i < G::BM * kp; i += G::NT) {
        const int r = i / kp, c = i % kp;
        const bool ok = c < D && m0 + r < M;
        // clamped, unconditional load (see gemm_rows); rows past M re-read the last row, columns past D the last column
        const size_t src = (size_t)min(m0 + r, M - 1) * D + min(c, D - 1);
        const float x = states[src];
        if (ok && states_copy) states_copy[src] = x;
        X[r * XS + c] = ok ? x : 0.f;
    }
}

// forward hidden-layer epilogue: h = tanh(acc + bias) -> slab (in place); KEEP: remember 1 - h^2 per element
template <int RT, int NW, bool KEEP, int ABL = 0>
__device__ __forceinline__ void epilogue_tanh(const f32x4 (&acc)[RT][Geo<RT, NW>::CT], const float* __restrict__ bias, float* X,
                                              f32x4 (*kept)[Geo<RT, NW>::CT]) {
    typedef Geo<RT, NW> G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) {
        const int col = wave * 16 * G::CT + ct * 16 + r16;
        const float b = bias[col];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = acc[rt][ct][r] + b;
                const float h = (ABL & ABL_NO_TANH) ? fminf(fmaxf(z, -1.f), 1.f) : fast_tanh(z);
                X[(rt * 16 + 4 * kq + r) * XS + col] = h;
                if constexpr (KEEP) kept[rt][ct][r] = 1.f - h * h;
            }
    }
    lds_barrier();
}

// head output (row, o): fmaf chain over the 256 hidden units in k order, then the bias
__device__ __forceinline__ float head_dot(const float* xr, const float* wr, float bias, bool has_bias) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < HID; j += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(xr + j);
        const f32x4 w = *reinterpret_cast<const f32x4*>(wr + j);
        s = fmaf(x[0], w[0], s);
        s = fmaf(x[1], w[1], s);
        s = fmaf(x[2], w[2], s);
        s = fmaf(x[3], w[3], s);
    }
    if (has_bias) s += bias;
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// rollout step: RT = 1 (16-row tiles), NW = 8 (512 threads): latency matters, B is ~1024 rows per step
// ---------------------------------------------------------------------------------------------------------------
template <int PD>
__global__ __launch_bounds__(512) void rollout_step_kernel(RolloutArgs a) {
    constexpr int RT = 1, NW = 8;
    typedef Geo<RT, NW> G;
    extern __shared__ __align__(16) float smem[];
    float* X = smem;
    float* W4s = smem + G::BM * XS;
    float* b4s = W4s + MAX_OUT * W4S;
    const rlx_mlp_layout& lay = a.lay;
    const int D = lay.obs_dim, tid = threadIdx.x;

    int b = blockIdx.x, y, job;  // job 0 = policy (y = net), 1 / 2 = value-only jobs
    long long m0, M;
    const float* states;
    float* states_copy = nullptr;
    if (b < 2 * a.tiles_policy) {
        job = 0; y = b & 1; m0 = (long long)(b >> 1) * G::BM; M = a.M; states = a.states;
        if (y == 1) states_copy = a.states_copy;
    } else {
        b -= 2 * a.tiles_policy;
        job = b < a.tiles_vj0 ? 1 : 2;
        if (job == 2) b -= a.tiles_vj0;
        y = 0; m0 = (long long)b * G::BM; M = a.vj[job - 1].m; states = a.vj[job - 1].states;
    }
    Stamps ts{a.stamps, 0};
    ts.mark();
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    RowGemm<RT, NW, PD, 0> gemm;
    gemm.prefetch(a.tiles + Tiles::mat(y, 0), Tiles::K1P / 32);
    load_states<RT, NW>(states, states_copy, D, m0, M, X);
    stage_head(a.params + lay.off_w[y][3], lay.off_b[y][3] >= 0 ? a.params + lay.off_b[y][3] : nullptr, n_out, W4s, b4s, G::NT);
    lds_barrier();
    ts.mark();
    f32x4 acc[RT][G::CT];
    gemm.run(X, acc);
    gemm.prefetch(a.tiles + Tiles::mat(y, 1), HID / 32);
    ts.mark();
    epilogue_tanh<RT, NW, false>(acc, a.params + lay.off_b[y][0], X, nullptr);
    ts.mark();
    gemm.run(X, acc);
    gemm.prefetch(a.tiles + Tiles::mat(y, 2), HID / 32);
    ts.mark();
    epilogue_tanh<RT, NW, false>(acc, a.params + lay.off_b[y][1], X, nullptr);
    ts.mark();
    gemm.run(X, acc);
    ts.mark();
    epilogue_tanh<RT, NW, false>(acc, a.params + lay.off_b[y][2], X, nullptr);
    ts.mark();

    for (int idx = tid; idx < G::BM * n_out; idx += G::NT) {
        const int row = idx / n_out, o = idx % n_out;
        const float s = head_dot(X + row * XS, W4s + o * W4S, b4s[o], lay.off_b[y][3] >= 0);
        if (m0 + row >= M) continue;
        const size_t g = (size_t)(m0 + row) * n_out + o;
        if (job == 0 && y == 0) {
            a.value[g] = s;
        } else if (job == 0) {
            const float mean = s;
            const float logstd = a.params[lay.off_logstd + o];
            const float stdv = expf(logstd);
            const float act = a.eps ? fadd(fmul(a.eps[g], stdv), mean) : mean;  // torch.normal: eps*std + mean; eval: mean
            const float d = fsub(act, mean);
            const float var = fmul(stdv, stdv);
            const float log_scale = logf(stdv);
            // Normal.log_prob: -((x - loc)**2) / (2*var) - log(scale) - log(sqrt(2*pi))
            a.logprob[g] = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
            a.action[g] = act;
        } else {
            value_job_output(a.vj[job - 1], g, (size_t)(m0 + row), o, s);
        }
    }
    ts.mark();
}

// ---------------------------------------------------------------------------------------------------------------
// fused optimizer-step kernel (1): forward + loss + backward-data for one tile of one network
// ---------------------------------------------------------------------------------------------------------------
// OP: head outputs padded to 8 or 16 (compile time): the loops over head outputs run unpredicated on zero padding --
// a runtime "o < n_out" guard around an LDS read serialises every read behind its own lgkmcnt(0).
template <int RT, int NW, int PD, int ABL, int OP, bool DEC = false>  // DEC: the decoupled actor loss (StepArgs.dec)
__global__ __launch_bounds__(64 * NW) void ppo_step_fused_kernel(StepArgs a) {
    typedef Geo<RT, NW> G;
    constexpr int BM = G::BM, CT = G::CT;
    extern __shared__ __align__(16) float smem[];
    float* X = smem;
    float* W4s = smem + BM * XS;                      // [MAX_OUT][W4S]
    float* b4s = W4s + MAX_OUT * W4S;                 // [MAX_OUT]
    float* sHead = b4s + MAX_OUT;                     // [BM][MAX_OUT] head outputs, then d(loss)/d(head output)
    float* sLp = sHead + BM * MAX_OUT;                // [BM][MAX_OUT] per-dimension log-probs, then per-dim d/d logstd
    float* sG = sLp + BM * MAX_OUT;                   // [BM][MAX_OUT] d(loss)/d(summed log-prob)
    float* sD = sG + BM * MAX_OUT;                    // [BM][MAX_OUT] action - mean
    double* sRed = reinterpret_cast<double*>(smem + BM * XS + G::AUX_FLOATS);  // NS * NW doubles of reduction scratch ...
    double* sNm = sRed + 256;                         // ... and the mask count (no static __shared__: it would shift the
                                                      // dynamic region off its 16-byte alignment)
    const rlx_mlp_layout& lay = a.lay;
    const rlx_ppo_loss_params& p = a.p;
    const int y = blockIdx.y, tile = blockIdx.x, D = lay.obs_dim, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const long long m0 = (long long)tile * BM, M = a.M;
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const int K = p.raw_per_adv, S = p.sub_per_adv, R = K / S;
    const int npr = lay.act_dim / K;                  // advantage elements per row
    const long long n_adv = M * npr;
    const bool has_mask = a.loss_mask != nullptr, has_msum = a.loss_mask_sum != nullptr;
    const bool ratio_mode = p.max_episode_steps > 0 && has_mask && has_msum;

    // ---- mask count of the whole micro-batch (the mean's denominator): 1 byte per element, L2-resident ------------
    if (has_mask) {
        double cnt[1] = {0.0};
        for (long long e = tid; e < n_adv; e += G::NT) cnt[0] += a.loss_mask[e] != 0 ? 1.0 : 0.0;
        block_sum<1>(cnt, sRed);
        if (tid == 0) sNm[0] = cnt[0];
    }

    // ---- forward -----------------------------------------------------------------------------------------------------
    Stamps ts{a.stamps, 0};
    ts.mark();
    RowGemm<RT, NW, PD, ABL> gemm;
    gemm.prefetch(a.tiles + Tiles::mat(y, 0), Tiles::K1P / 32);
    load_states<RT, NW>(a.states, nullptr, D, m0, M, X);
    stage_head(a.params + lay.off_w[y][3], lay.off_b[y][3] >= 0 ? a.params + lay.off_b[y][3] : nullptr, n_out, W4s, b4s, G::NT);
    for (int i = tid; i < BM * MAX_OUT; i += G::NT) sHead[i] = 0.f;                       // zero padding for o >= n_out
    for (int i = n_out * W4S + tid; i < OP * W4S; i += G::NT) W4s[i] = 0.f;
    lds_barrier();
    ts.mark();
    f32x4 acc[RT][CT];
    f32x4 kept[3][RT][CT];  // 1 - h_l^2 in the accumulator layout: the backward epilogues need exactly these lanes
    float* hy = a.h + (size_t)(y * 2) * M * HID;
    float* dzy = a.dz + (size_t)(y * 3) * M * HID;
    gemm.run(X, acc);
    gemm.prefetch(a.tiles + Tiles::mat(y, 1), HID / 32);
    ts.mark();
    epilogue_tanh<RT, NW, true, ABL>(acc, a.params + lay.off_b[y][0], X, kept[0]);
    flush_rows<RT, NW, ABL>(X, hy, m0, M);
    ts.mark();
    gemm.run(X, acc);
    gemm.prefetch(a.tiles + Tiles::mat(y, 2), HID / 32);
    ts.mark();
    epilogue_tanh<RT, NW, true, ABL>(acc, a.params + lay.off_b[y][1], X, kept[1]);
    flush_rows<RT, NW, ABL>(X, hy + (size_t)M * HID, m0, M);
    ts.mark();
    gemm.run(X, acc);
    ts.mark();
    epilogue_tanh<RT, NW, true, ABL>(acc, a.params + lay.off_b[y][2], X, kept[2]);
    ts.mark();

    // ---- head + loss element math ---------------------------------------------------------------------------------------
    const double nm = has_mask ? sNm[0] : 0.0;
    const Denoms den = denominators(p, n_adv, nm, has_mask, has_msum);
    const float half_delta = (float)(0.5 * (double)p.huber_delta);
    double lacc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) lacc[k] = 0.0;
    DecoupledMode dmode{};
    if constexpr (DEC) dmode = decoupled_mode_now(a.dec);

    for (int idx = tid; idx < BM * n_out; idx += G::NT) {
        const int row = idx / n_out, o = idx % n_out;
        const float s = head_dot(X + row * XS, W4s + o * W4S, b4s[o], lay.off_b[y][3] >= 0);
        sHead[row * MAX_OUT + o] = s;
        if (y == 1) {
            const size_t g = (size_t)min(m0 + row, M - 1) * n_out + o;  // clamped (rows past M are discarded below)
            const float stdv = expf(a.params[lay.off_logstd + o]);
            const float d = fsub(a.action[g], s);
            const float var = fmul(stdv, stdv);
            const float log_scale = logf(stdv);
            sLp[row * MAX_OUT + o] = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
            sD[row * MAX_OUT + o] = d;
        }
    }
    lds_barrier();
    if (y == 1) {
        // one lane of wave 0 per advantage element of the tile: summed log-probs -> ratio / clip / dual clip -> gradient
        for (int idx = tid; idx < BM * npr && wave == 0; idx += 64) {  // wave 0 only: its lanes hold every metric sum
            const int row = idx / npr, c = idx % npr;
            if (m0 + row >= M) continue;
            const long long e = (m0 + row) * npr + c;
            const bool on = has_mask ? a.loss_mask[e] != 0 : true;
            float w = 1.f;
            if (ratio_mode) w = ((float)a.loss_mask_sum[e] * 1.0f) / (float)p.max_episode_steps;
            const float adv = a.advantages[e];
            lacc[S_NM] += on ? 1.0 : 0.0;
            const float* olp = a.old_logprobs + (size_t)(m0 + row) * lay.act_dim + c * K;
            for (int s = 0; s < S; ++s) {
                float lp = 0.f, old = 0.f;
                for (int j = 0; j < R; ++j) {
                    lp = fadd(lp, sLp[row * MAX_OUT + c * K + s * R + j]);
                    old = fadd(old, olp[s * R + j]);
                }
                if constexpr (DEC) {  // sum form (the denominator is out[RLX_PPO_ACTOR_GRAD_SCALE] of the finished row); the slice's
                                      // raw entries are [e * K + s * R, + R) of the [M, act_dim] arrays
                    const size_t r0 = (size_t)e * K + s * R;
                    float px = 0.f;
                    if (dmode.mode == RLX_PROX_GIVEN)
                        for (int j = 0; j < R; ++j) px = fadd(px, a.dec.proximal[r0 + j]);
                    const float vb = a.dec.versions != nullptr ? a.dec.versions[r0] : 0.f;
                    sG[row * MAX_OUT + c * S + s] =
                        a.grad_out * decoupled_actor_elem(p, dmode, lp, old, px, vb, adv, on, w, ratio_mode, lacc, S_VLOSS);
                } else {
                    const float g = actor_elem(p, lp, old, adv, on, w, ratio_mode, lacc);
                    sG[row * MAX_OUT + c * S + s] = (a.grad_out * (float)(1.0 / den.actor)) * g;
                }
            }
        }
        lds_barrier();
        for (int idx = tid; idx < BM * n_out; idx += G::NT) {
            const int row = idx / n_out, o = idx % n_out;
            float dmu = 0.f, dls = 0.f;
            if (m0 + row < M) {
                const float dlp = sG[row * MAX_OUT + (o / K) * S + (o % K) / R];
                const float stdv = expf(a.params[lay.off_logstd + o]);
                const float var = stdv * stdv, d = sD[row * MAX_OUT + o];
                dmu = dlp * d / var;                     // d logprob / d mean
                dls = dlp * (d * d / var - 1.f);         // d logprob / d logstd
            }
            sHead[row * MAX_OUT + o] = dmu;
            sLp[row * MAX_OUT + o] = dls;
        }
    } else {
        for (int idx = tid; idx < BM * n_out && wave == 0; idx += 64) {  // wave 0 only (metric sums)
            const int row = idx / n_out, o = idx % n_out;
            float gv = 0.f;
            if (m0 + row < M && p.has_critic) {
                const long long e = (m0 + row) * n_out + o;
                const bool on = has_mask ? a.loss_mask[e] != 0 : true;
                float w = 1.f;
                if (ratio_mode) w = ((float)a.loss_mask_sum[e] * 1.0f) / (float)p.max_episode_steps;
                gv = (a.grad_out * (float)(1.0 / den.critic)) *
                     critic_elem(p, sHead[row * MAX_OUT + o], a.prev_values[e], a.returns[e], on, w, ratio_mode, half_delta, lacc);
            }
            sHead[row * MAX_OUT + o] = gv;  // overwritten by the thread that read it
        }
    }
    // metric sums of this tile: every contribution sits in wave 0 -> a wave-level reduction, no barrier, no LDS scratch
    if (wave == 0) {
        double* lp = a.loss_part + ((size_t)tile * 2 + y) * NS;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const double v = wave_sum(lacc[k]);
            if (lane == 0) lp[k] = v;
        }
    }
    lds_barrier();
    ts.mark();

    // ---- head parameter gradients, per 32-row half tile (h3 is still in the slab): thread = (half, hidden column j) ----
    for (int u = tid; u < (BM / 32) * HID; u += G::NT) {
        const int sub = u / HID, j = u % HID, r0 = sub * 32;
        float* part = a.head_part + ((size_t)(tile * (BM / 32) + sub) * 2 + y) * a.head_stride;
        float s[OP];
#pragma unroll
        for (int o = 0; o < OP; ++o) s[o] = 0.f;
#pragma unroll 4
        for (int row = r0; row < r0 + 32; ++row) {
            const float hv = X[row * XS + j];
#pragma unroll
            for (int q = 0; q < OP / 4; ++q) {
                const f32x4 sh = *reinterpret_cast<const f32x4*>(sHead + row * MAX_OUT + 4 * q);  // same address in every lane
#pragma unroll
                for (int i = 0; i < 4; ++i) s[4 * q + i] = fmaf(sh[i], hv, s[4 * q + i]);
            }
        }
#pragma unroll
        for (int o = 0; o < OP; ++o)
            if (o < n_out) part[o * HID + j] = s[o];
        if (j < n_out) {
            float sb = 0.f, sl = 0.f;
            for (int row = r0; row < r0 + 32; ++row) {
                sb += sHead[row * MAX_OUT + j];
                sl += sLp[row * MAX_OUT + j];
            }
            part[n_out * HID + j] = sb;
            part[n_out * HID + n_out + j] = y == 1 ? sl : 0.f;
        }
    }
    ts.mark();

    // ---- dZ3 = (dOut . W4) * (1 - h3^2), in the accumulator layout, into the slab ---------------------------------------
    gemm.prefetch(a.tiles + Tiles::mat(y, 4), HID / 32);  // W3^T tiles for the first backward GEMM
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = wave * 16 * CT + ct * 16 + r16;
        float w4[OP];
#pragma unroll
        for (int o = 0; o < OP; ++o) w4[o] = W4s[o * W4S + col];  // rows >= n_out are zero
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + 4 * kq + r;
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < OP / 4; ++q) {
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(sHead + row * MAX_OUT + 4 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) s = fmaf(sh[i], w4[4 * q + i], s);
                }
                acc[rt][ct][r] = s * kept[2][rt][ct][r];
            }
    }
    lds_barrier();  // every read of h3 / sHead is done: the slab may be overwritten
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[(rt * 16 + 4 * kq + r) * XS + wave * 16 * CT + ct * 16 + r16] = acc[rt][ct][r];
    lds_barrier();
    flush_rows<RT, NW, ABL>(X, dzy + 2 * (size_t)M * HID, m0, M);
    ts.mark();

    // ---- backward-data chain -------------------------------------------------------------------------------------------------
#pragma unroll
    for (int l = 2; l >= 1; --l) {
        gemm.run(X, acc);  // dH_l = dZ_{l+1} . W_{l+1}  (W^T tiles, prefetched)
        if (l == 2) gemm.prefetch(a.tiles + Tiles::mat(y, 3), HID / 32);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    X[(rt * 16 + 4 * kq + r) * XS + wave * 16 * CT + ct * 16 + r16] = acc[rt][ct][r] * kept[l - 1][rt][ct][r];
        lds_barrier();
        flush_rows<RT, NW, ABL>(X, dzy + (size_t)(l - 1) * M * HID, m0, M);
        ts.mark();
    }
    ts.mark();
}

// ---------------------------------------------------------------------------------------------------------------
// fused optimizer-step kernel (2): weight gradients (split-K slabs), head-gradient fold, metric finalisation
//   1-D grid: [GEMM items (padded to a multiple of 8, XCD-grouped)] [slabs x 2 head-fold blocks] [1 finalise block]
//   GEMM item = (slab, matrix, 128x128 output tile); 4 waves as 2x2, each 2x2 tiles of v_mfma_f32_32x32x2_f32.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int frag_row32(int r, int khalf) { return (r & 3) + 8 * (r >> 2) + 4 * khalf; }

// ---- f32 products on the bf16 matrix pipe ("3 x bf16 split") ------------------------------------------------------------
// An f32 number is EXACTLY hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (3 x 8 significand bits,
// the differences are exact in f32: tests/test_bf16_split_exact.py), and a product of two bf16 numbers is exact in f32.  So
// a . b = sum over the nine partial products; the six kept here -- hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid -- leave out terms
// below 2^-24 of |a||b| (mid.lo, lo.mid, lo.lo), i.e. below the rounding of the f32 accumulation itself.  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each for K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each for K = 2): 2.7 x less
// matrix-pipe time for the same f32-level accuracy -- the f32 MFMA rate of this part is 1/16 of its bf16 rate.  The operand
// split runs on the VALU (packed converts), next to the matrix pipe.
typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
struct DwSplit3 {
    dw_bf16x8 hi, mid, lo;
};
__device__ __forceinline__ DwSplit3 dw_split3(const float (&w)[8]) {
    DwSplit3 q;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)w[j];
        const float r1 = fsub(w[j], (float)h);
        const __bf16 m = (__bf16)r1;
        const float r2 = fsub(r1, (float)m);
        q.hi[j] = h;
        q.mid[j] = m;
        q.lo[j] = (__bf16)r2;
    }
    return q;
}
__device__ __forceinline__ f32x16 dw_mfma6(const DwSplit3& a, const DwSplit3& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);   // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

// SPLIT: the products run as 3 x bf16 splits on v_mfma_f32_32x32x16_bf16 (default); false: exact f32 products on
// v_mfma_f32_32x32x2_f32 (RLX_F32_EXACT_MFMA=1).
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void ppo_step_dw_kernel(DwArgs a) {
    __shared__ __align__(16) float As[2][32][128];
    __shared__ __align__(16) float Bs[2][32][128];
    double* s_red = reinterpret_cast<double*>(&As[0][0][0]);
    const rlx_mlp_layout& lay = a.lay;
    const long long M = a.M;
    const int tid = threadIdx.x;
    const int gemm_blocks = round_up(a.gemm_items, 8);
    int b = blockIdx.x;

    if (b >= gemm_blocks) {
        b -= gemm_blocks;
        if (b < a.slabs * 2) {
            // ---- head gradients: slab s takes the 32-row partials t == s (mod slabs); thread = hidden column j -------------
            head_reduce_block(a, b >> 1, b & 1, tid);
        } else {
            // ---- metric row: sum the per-tile partials of both networks ---------------------------------------
            metric_block(a, s_red, tid, 256);
        }
        return;
    }
    // XCD grouping: block b runs on XCD b % 8; give each XCD a contiguous run of items, so the tiles that share A / B
    // operand rows (same slab, same matrix) hit the same L2
    const int item = (b & 7) * (gemm_blocks >> 3) + (b >> 3);
    if (item >= a.gemm_items) return;
    const int s = item / 20, w = item % 20;  // w: 0-15 hidden matrices (4 tiles each), 16-19 first layers (2 tiles each)
    int y, l, i0, j0;
    if (w < 16) {
        const int mat = w >> 2, tile = w & 3;
        y = mat >> 1; l = 1 + (mat & 1); i0 = (tile >> 1) * 128; j0 = (tile & 1) * 128;
    } else {
        y = (w - 16) >> 1; l = 0; i0 = ((w - 16) & 1) * 128; j0 = 0;
    }
    const int Kin = l == 0 ? lay.obs_dim : HID;
    const float* A = a.dz + (size_t)(y * 3 + l) * M * HID;                          // [M][256]
    const float* Bm = l == 0 ? a.states : a.h + (size_t)(y * 2 + l - 1) * M * HID;  // [M][Kin]
    const bool vecB = (Kin % 4) == 0;
    const int lane = tid & 63, wave = tid >> 6, lrow = lane & 31, khalf = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const long long r_begin = (long long)s * a.rows_per_slab;
    const long long r_end = min(M, r_begin + a.rows_per_slab);
    const int nchunks = r_end > r_begin ? (int)((r_end - r_begin + 31) / 32) : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    float bsum[2] = {0.f, 0.f};

    const int ldr = tid >> 5, ldc4 = (tid & 31) * 4;  // loader: row ldr + 8*i, 4 columns at ldc4
    float4 sa[4], sb[4];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long row = r_begin + (long long)c * 32 + ldr + 8 * i;
            const bool ok = row < r_end;
            sa[i] = ok ? *reinterpret_cast<const float4*>(A + (size_t)row * HID + i0 + ldc4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const int col = j0 + ldc4;
                if (vecB) {
                    if (col < Kin) v = *reinterpret_cast<const float4*>(Bm + (size_t)row * Kin + col);
                } else {
                    const float* q = Bm + (size_t)row * Kin + col;
                    if (col + 0 < Kin) v.x = q[0];
                    if (col + 1 < Kin) v.y = q[1];
                    if (col + 2 < Kin) v.z = q[2];
                    if (col + 3 < Kin) v.w = q[3];
                }
            }
            sb[i] = v;
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[buf][ldr + 8 * i][ldc4]) = sa[i];
            *reinterpret_cast<float4*>(&Bs[buf][ldr + 8 * i][ldc4]) = sb[i];
        }
    };
    if (nchunks > 0) {
        gload(0);
        swrite(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunks) gload(c + 1);
        if constexpr (SPLIT) {
            // two 16-k blocks per 32-row chunk; lane (column lrow, k group khalf) holds k = 16 kb + 8 khalf + j, j = 0 .. 7 of its
            // column for A and for B alike (K is only the summation index: the slots just have to pair up)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                float af[2][8], bf[2][8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * kb + 8 * khalf + j;
                    af[0][j] = As[cur][k][wi * 64 + lrow];
                    af[1][j] = As[cur][k][wi * 64 + 32 + lrow];
                    bf[0][j] = Bs[cur][k][wj * 64 + lrow];
                    bf[1][j] = Bs[cur][k][wj * 64 + 32 + lrow];
                }
                const DwSplit3 A0 = dw_split3(af[0]), A1 = dw_split3(af[1]), B0 = dw_split3(bf[0]), B1 = dw_split3(bf[1]);
                acc[0][0] = dw_mfma6(A0, B0, acc[0][0]);
                acc[0][1] = dw_mfma6(A0, B1, acc[0][1]);
                acc[1][0] = dw_mfma6(A1, B0, acc[1][0]);
                acc[1][1] = dw_mfma6(A1, B1, acc[1][1]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bsum[0] += af[0][j];
                    bsum[1] += af[1][j];
                }
            }
        } else {
#pragma unroll 4
            for (int kp = 0; kp < 16; ++kp) {
                const int k = 2 * kp + khalf;
                const float a0 = As[cur][k][wi * 64 + lrow], a1 = As[cur][k][wi * 64 + 32 + lrow];
                const float b0 = Bs[cur][k][wj * 64 + lrow], b1 = Bs[cur][k][wj * 64 + 32 + lrow];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                bsum[0] += a0;
                bsum[1] += a1;
            }
        }
        if (c + 1 < nchunks) swrite(cur ^ 1);
        __syncthreads();
    }
    float* slab = a.grads + (size_t)s * lay.n_params;
    float* dW = slab + lay.off_w[y][l];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = j0 + wj * 64 + u * 32 + lrow;
            if (col < Kin) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wi * 64 + t * 32 + frag_row32(r, khalf);
                    dW[(size_t)row * Kin + col] = acc[t][u][r];
                }
            }
        }
    if (j0 == 0 && wj == 0) {  // bias gradient = column sums of dZ, folded in as the A fragments go by
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float tot = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            if (khalf == 0) slab[lay.off_b[y][l] + i0 + wi * 64 + t * 32 + lrow] = tot;
        }
    }
}

int pack_tiles(const float* params, const rlx_mlp_layout& lay, float* tiles, hipStream_t st) {
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(num_cu() * 4), dim3(256), 0, st, params, lay, tiles);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // namespace
namespace step {
long long* g_timing_buffer = nullptr;
}
}  // namespace rlx

using namespace rlx;

// development hook (not declared in include/rlx.h): device buffer of >= 32 int64 for the phase stamps, or NULL
extern "C" void rlx_dev_set_timing_buffer(void* device_ptr) { g_timing_buffer = static_cast<long long*>(device_ptr); }

extern "C" int rlx_mlp_rollout_step(const rlx_rollout_step* r, rlx_stream_t stream) {
    RLX_REQUIRE(r != nullptr, "rlx_mlp_rollout_step: NULL argument struct");
    if (int rc = check_layout(r->layout, "rlx_mlp_rollout_step")) return rc;
    RLX_REQUIRE(r->params != nullptr && r->tiles != nullptr, "rlx_mlp_rollout_step: NULL params / tiles");
    RLX_REQUIRE(r->m >= 0 && r->n_value_jobs >= 0 && r->n_value_jobs <= 2, "rlx_mlp_rollout_step: bad sizes");
    RLX_REQUIRE(r->m == 0 || (r->states && r->action && r->logprob && r->value), "rlx_mlp_rollout_step: NULL policy tensor");
    RolloutArgs a{};
    a.stamps = g_timing_buffer;
    a.params = r->params; a.tiles = r->tiles; a.lay = *r->layout; a.states = r->states; a.eps = r->eps; a.M = r->m;
    a.action = r->action; a.logprob = r->logprob; a.value = r->value; a.states_copy = r->states_copy;
    const int bm = r->bf16 ? rollout_bm_bf16() : 16;  // rows per workgroup
    a.tiles_policy = ceil_div(r->m, bm);
    int tv[2] = {0, 0};
    for (int k = 0; k < r->n_value_jobs; ++k) {
        const rlx_value_job& j = r->value_jobs[k];
        RLX_REQUIRE(j.m >= 0 && (j.m == 0 || j.states != nullptr), "rlx_mlp_rollout_step: value job %d has no states", k);
        const bool env_store = j.env_rewards != nullptr;
        RLX_REQUIRE(j.rewards == nullptr || ((env_store || j.flags != nullptr) && j.chunk >= 1),
                    "rlx_mlp_rollout_step: value job %d folds rewards without flags", k);
        RLX_REQUIRE(!env_store || (j.rewards && j.env_terminations && j.env_truncations && j.done_row && j.termination_row && j.truncation_row),
                    "rlx_mlp_rollout_step: value job %d stores env rows but a pointer is NULL", k);
        a.vj[k] = ValueJob{j.states, j.m, j.values, j.rewards, j.flags, j.chunk, j.gamma, j.env_rewards, j.env_terminations,
                           j.env_truncations, j.done_row, j.termination_row, j.truncation_row, j.flag_is_truncation};
        tv[k] = ceil_div(j.m, bm);
    }
    a.tiles_vj0 = tv[0]; a.tiles_vj1 = tv[1];
    const int blocks = 2 * a.tiles_policy + tv[0] + tv[1];
    if (blocks == 0) return RLX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (r->bf16) return launch_rollout_bf16(a, blocks, st);
    if (f32_split()) return launch_rollout_f32x(a, blocks, st);
    const size_t lds = Geo<1, 8>::LDS_BYTES;
    const int v = dev_variant("RLX_ROLLOUT_PD", 2);
#define RLX_LAUNCH_ROLLOUT(PDV)                                                                 \
    do {                                                                                        \
        if (int rc = set_lds(rollout_step_kernel<PDV>, lds)) return rc;                         \
        hipLaunchKernelGGL(rollout_step_kernel<PDV>, dim3(blocks), dim3(512), lds, st, a);      \
    } while (0)
#ifdef RLX_DEV_VARIANTS
    if (v == 1) RLX_LAUNCH_ROLLOUT(1);
    else if (v == 4) RLX_LAUNCH_ROLLOUT(4);
    else if (v == 8) RLX_LAUNCH_ROLLOUT(8);
    else RLX_LAUNCH_ROLLOUT(2);
#else
    (void)v;
    RLX_LAUNCH_ROLLOUT(2);
#endif
#undef RLX_LAUNCH_ROLLOUT
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" size_t rlx_mlp_tiles_bytes(const rlx_mlp_layout* lay) {
    (void)lay;
    return std::max(2 * Tiles::per_net() * sizeof(float), f32x_tiles_bytes());  // any of the image formats fits
}

extern "C" size_t rlx_mlp_tiles_bytes_for(const rlx_mlp_layout* lay, int32_t bf16) {
    (void)lay;
    if (bf16) return 2 * Tiles::per_net() * sizeof(__bf16);
    return f32_split() ? f32x_tiles_bytes() : 2 * Tiles::per_net() * sizeof(float);
}

extern "C" int rlx_mlp_pack_tiles(const float* params, const rlx_mlp_layout* lay, float* tiles, rlx_stream_t stream) {
    if (int rc = check_layout(lay, "rlx_mlp_pack_tiles")) return rc;
    RLX_REQUIRE(params && tiles, "rlx_mlp_pack_tiles: NULL argument");
    if (f32_split()) return pack_tiles_f32x(params, *lay, tiles, static_cast<hipStream_t>(stream));
    return pack_tiles(params, *lay, tiles, static_cast<hipStream_t>(stream));
}

extern "C" int rlx_mlp_pack_tiles_bf16(const float* params, const rlx_mlp_layout* lay, void* tiles, rlx_stream_t stream) {
    if (int rc = check_layout(lay, "rlx_mlp_pack_tiles_bf16")) return rc;
    RLX_REQUIRE(params && tiles, "rlx_mlp_pack_tiles_bf16: NULL argument");
    return pack_tiles_bf16(params, *lay, tiles, static_cast<hipStream_t>(stream));
}

extern "C" int rlx_ppo_step_slabs(const rlx_mlp_layout* lay, int64_t m) {
    if (!lay || m <= 0) return 1;
    return plan_step(lay, m).slabs;
}

extern "C" int rlx_ppo_step_slabs_for(const rlx_mlp_layout* lay, int64_t m, int32_t bf16) {
    if (!lay || m <= 0) return 1;
    return plan_step(lay, m, bf16 != 0).slabs;
}

extern "C" size_t rlx_ppo_step_workspace_bytes(const rlx_mlp_layout* lay, int64_t m) {
    if (!lay || m <= 0) return 256;
    return std::max({plan_step(lay, m, false).bytes, plan_step(lay, m, true).bytes, plan_step(lay, m, true, true).bytes});  // any launch fits
}

extern "C" int rlx_ppo_step(const rlx_ppo_step_args* s, rlx_stream_t stream) {
    RLX_REQUIRE(s != nullptr && s->loss != nullptr, "rlx_ppo_step: NULL argument struct");
    if (int rc = check_layout(s->layout, "rlx_ppo_step")) return rc;
    const rlx_mlp_layout& lay = *s->layout;
    const rlx_ppo_loss_params& p = *s->loss;
    RLX_REQUIRE(s->m >= 1, "rlx_ppo_step: empty micro-batch");
    RLX_REQUIRE(p.raw_per_adv >= 1 && p.sub_per_adv >= 1 && p.raw_per_adv % p.sub_per_adv == 0 && lay.act_dim % p.raw_per_adv == 0,
                "rlx_ppo_step: raw_per_adv=%d / sub_per_adv=%d do not tile act_dim=%d", p.raw_per_adv, p.sub_per_adv, lay.act_dim);
    RLX_REQUIRE(!p.has_critic || lay.act_dim / p.raw_per_adv == lay.val_dim,
                "rlx_ppo_step: %d advantage elements per row but %d value outputs", lay.act_dim / p.raw_per_adv, lay.val_dim);
    RLX_REQUIRE(!p.use_dual_clip || p.clip_ratio_c > 1.0f, "clip_ratio_c must be greater than 1.0");  // losses.py:262
    RLX_REQUIRE(s->params && s->states && s->action && s->old_logprobs && s->advantages && s->grads && s->out && s->workspace,
                "rlx_ppo_step: NULL argument");
    RLX_REQUIRE(!p.has_critic || (s->prev_values && s->returns), "rlx_ppo_step: has_critic set but a critic tensor is NULL");
    const bool bf16 = s->bf16 != 0;
    const rlx_decoupled_loss_params* dp = s->decoupled;
    if (dp != nullptr) {
        RLX_REQUIRE(dp->proximal_mode >= RLX_PROX_GIVEN && dp->proximal_mode <= RLX_PROX_FROM_VERSIONS,
                    "rlx_ppo_step: unknown proximal_mode %d", dp->proximal_mode);
        RLX_REQUIRE(dp->proximal_mode != RLX_PROX_GIVEN || s->proximal_logprobs, "rlx_ppo_step: proximal_mode GIVEN without proximal_logprobs");
        RLX_REQUIRE(dp->proximal_mode != RLX_PROX_FROM_VERSIONS || s->versions, "rlx_ppo_step: proximal_mode FROM_VERSIONS without versions");
    }
    const bool rows = bf16 && dp == nullptr && fused_rows_bf16() && fused_rows_eligible(lay, p);
    const StepPlan pl = plan_step(&lay, s->m, bf16, rows);
    RLX_REQUIRE(s->slabs == pl.slabs, "rlx_ppo_step: grads holds %d slabs, rlx_ppo_step_slabs() says %d", s->slabs, pl.slabs);
    if (s->workspace_bytes < pl.bytes) {
        set_error("rlx_ppo_step: workspace %zu < %zu bytes", s->workspace_bytes, pl.bytes);
        return RLX_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(s->workspace);
    StepArgs a{};
    a.stamps = g_timing_buffer;
    a.merged_loss_pass = dev_variant("RLX_FUSED_MERGED", 1);
    a.xcd_rows = dev_variant("RLX_FUSED_XCD_ROWS", 0);  // measured: no effect (see the kernel), identity kept
    a.params = s->params; a.lay = lay; a.states = s->states; a.action = s->action; a.old_logprobs = s->old_logprobs;
    a.advantages = s->advantages; a.prev_values = s->prev_values; a.returns = s->returns; a.loss_mask = s->loss_mask;
    a.loss_mask_sum = s->loss_mask_sum; a.M = s->m; a.p = p; a.grad_out = s->grad_out;
    a.h = reinterpret_cast<float*>(ws + pl.off_h); a.dz = reinterpret_cast<float*>(ws + pl.off_dz);
    a.head_part = reinterpret_cast<float*>(ws + pl.off_head); a.loss_part = reinterpret_cast<double*>(ws + pl.off_loss);
    a.head_stride = pl.head_stride;
    if (dp != nullptr) {
        a.dec.on = 1;
        a.dec.mode = loss::DecoupledMode{dp->proximal_mode, dp->use_behave_threshold, dp->behave_weight_threshold, dp->current_version};
        a.dec.v_theta_dev = s->current_version_dev; a.dec.proximal = s->proximal_logprobs; a.dec.versions = s->versions;
    }
    DwArgs d{};
    d.stamps = g_timing_buffer ? g_timing_buffer + 32 : nullptr;
    d.lay = lay; d.states = s->states; d.h = a.h; d.dz = a.dz; d.head_part = a.head_part; d.loss_part = a.loss_part;
    d.M = s->m; d.rows_per_slab = pl.rows_per_slab; d.slabs = pl.slabs; d.tiles = pl.loss_slots; d.head_parts = pl.head_parts;
    d.head_stride = pl.head_stride;
    d.gemm_items = pl.slabs * 20; d.grads = s->grads; d.p = p; d.has_mask = s->loss_mask != nullptr;
#ifdef RLX_DEV_VARIANTS  // timing tool only (the sums come out multiplied): not reachable in the product build
    d.repeat = std::max(1, dev_variant("RLX_DW_REPEAT", 1));
#else
    d.repeat = 1;
#endif
    d.has_msum = s->loss_mask_sum != nullptr; d.out = s->out;
    d.decoupled = dp != nullptr; d.dec_use_threshold = dp != nullptr && dp->use_behave_threshold != 0;
    const int dw_blocks = round_up(d.gemm_items, 8) + pl.slabs * 2 + 1;
    if (bf16) {
        if (s->tiles != nullptr) {
            a.tiles = s->tiles;
        } else {
            void* tiles = ws + pl.off_tiles;
            a.tiles = static_cast<const float*>(tiles);
            if (int rc = pack_tiles_bf16(s->params, lay, tiles, st)) return rc;
        }
        return launch_step_bf16(a, d, ws + pl.off_st, pl.tiles, dw_blocks, lay.act_dim <= 8 && lay.val_dim <= 8, rows, st);
    }
    if (f32_split()) {  // the default f32 path: three bf16 planes per operand on the bf16 matrix pipe (ppo_step_f32x.hip)
        if (s->tiles != nullptr) {
            a.tiles = s->tiles;
        } else {
            void* tiles = ws + pl.off_tiles;
            a.tiles = static_cast<const float*>(tiles);
            if (int rc = pack_tiles_f32x(s->params, lay, tiles, st)) return rc;
        }
        return launch_step_f32x(a, d, ws + pl.off_st, pl.tiles, dw_blocks, st);
    }
    if (s->tiles != nullptr) {
        a.tiles = s->tiles;
    } else {
        float* tiles = reinterpret_cast<float*>(ws + pl.off_tiles);
        a.tiles = tiles;
        if (int rc = pack_tiles(s->params, lay, tiles, st)) return rc;
    }
    const size_t lds = Geo<4, 8>::LDS_BYTES;
    const int v = dev_variant("RLX_STEP_VARIANT", 0);  // development: PD * 100 + ablation bits
    const bool op8 = lay.act_dim <= 8 && lay.val_dim <= 8;
#define RLX_LAUNCH_FUSED(PDV, ABLV, OPV)                                                                                  \
    do {                                                                                                                  \
        if (int rc = set_lds(ppo_step_fused_kernel<4, 8, PDV, ABLV, OPV>, lds)) return rc;                                \
        hipLaunchKernelGGL((ppo_step_fused_kernel<4, 8, PDV, ABLV, OPV>), dim3(pl.tiles, 2), dim3(512), lds, st, a);      \
    } while (0)
    // ablation builds (RLX_STEP_VARIANT, development): compiled only with -DRLX_DEV_VARIANTS, see tools/bench_step.py
    if (a.dec.on) {
        if (op8) {
            if (int rc = set_lds(ppo_step_fused_kernel<4, 8, 2, 0, 8, true>, lds)) return rc;
            hipLaunchKernelGGL((ppo_step_fused_kernel<4, 8, 2, 0, 8, true>), dim3(pl.tiles, 2), dim3(512), lds, st, a);
        } else {
            if (int rc = set_lds(ppo_step_fused_kernel<4, 8, 2, 0, 16, true>, lds)) return rc;
            hipLaunchKernelGGL((ppo_step_fused_kernel<4, 8, 2, 0, 16, true>), dim3(pl.tiles, 2), dim3(512), lds, st, a);
        }
    } else if (!op8) {
        RLX_LAUNCH_FUSED(2, 0, 16);
    } else {
#ifdef RLX_DEV_VARIANTS
        switch (v) {
            case 201: RLX_LAUNCH_FUSED(2, 1, 8); break;
            case 202: RLX_LAUNCH_FUSED(2, 2, 8); break;
            case 204: RLX_LAUNCH_FUSED(2, 4, 8); break;
            case 208: RLX_LAUNCH_FUSED(2, 8, 8); break;
            case 400: RLX_LAUNCH_FUSED(4, 0, 8); break;
            case 300: RLX_LAUNCH_FUSED(3, 0, 8); break;
            default: RLX_LAUNCH_FUSED(2, 0, 8); break;
        }
#else
        (void)v;
        RLX_LAUNCH_FUSED(2, 0, 8);
#endif
    }
#undef RLX_LAUNCH_FUSED
    RLX_LAUNCH_CHECK();
    const int blocks = dw_blocks;
    // (reached with RLX_F32_EXACT_MFMA=1 only: exact f32 products on v_mfma_f32_32x32x2_f32)
    hipLaunchKernelGGL(ppo_step_dw_kernel<false>, dim3(blocks), dim3(256), 0, st, d);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
