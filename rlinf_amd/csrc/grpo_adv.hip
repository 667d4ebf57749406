// grpo_adv.hip -- embodied GRPO advantages, gfx950.
//
// Replaces calculate_scores (rlinf/algorithms/utils.py:134-152): per-env reverse scan
//     s = s * ~done[t+1] + r[t]     (return of the env's FIRST episode segment)
// and compute_grpo_advantages (rlinf/algorithms/advantages.py:89-121): consecutive envs form groups,
//     a = (s - mean_g) / (std_g(unbiased) + 1e-6),   adv[t,b] = a[b] * loss_mask[t,b].
//
// Kernel 1 (grpo_scores): the scan is the affine map g -> r + alive*g with alive in {0,1}; lanes run
// along envs, the time axis is split across the block's waves, each wave reduces its segment to a map
// (A, G) and wave 0 composes them through LDS.  4+1 B read per env-step.
// Kernel 2 (grpo_broadcast): every block normalises the scores of its own env columns (group members
// are L2-resident, group_size loads each) and streams mask rows -> advantage rows (1 B read, 4 B write).

#include "rlx_common.h"

namespace rlx {
namespace {

__device__ __forceinline__ size_t tc_index(int f, size_t b, size_t B, int C) {
    return ((size_t)(f / C) * B + b) * C + (f % C);
}

__global__ __launch_bounds__(512) void grpo_scores(const float* __restrict__ r, const uint8_t* __restrict__ d,
                                                   float* __restrict__ scores, int T, int B, int C) {
    extern __shared__ float smf[];
    const int lane = threadIdx.x & 63;
    const int nseg = blockDim.x >> 6;
    const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t b = (size_t)blockIdx.x * 64 + lane;
    const bool active = b < (size_t)B;
    const int seg_len = (T + nseg - 1) / nseg;
    const int t_lo = seg * seg_len, t_hi = min(T, t_lo + seg_len);
    float g = 0.f, A = 1.f;
    if (active) {
        if (C == 1) {
#pragma unroll 8
            for (int t = t_hi - 1; t >= t_lo; --t) {
                const float alive = d[(size_t)(t + 1) * B + b] ? 0.f : 1.f;
                g = fadd(fmul(g, alive), r[(size_t)t * B + b]);
                A *= alive;
            }
        } else {
            for (int t = t_hi - 1; t >= t_lo; --t) {
                const float alive = d[tc_index(t + 1 + (C - 1), b, B, C)] ? 0.f : 1.f;
                g = fadd(fmul(g, alive), r[tc_index(t, b, B, C)]);
                A *= alive;
            }
        }
    }
    if (nseg == 1) {
        if (active) scores[b] = g;
        return;
    }
    smf[seg * 64 + lane] = A;
    smf[(nseg + seg) * 64 + lane] = g;
    __syncthreads();
    if (seg == 0 && active) {
        float acc = 0.f;
        for (int s = nseg - 1; s >= 0; --s) acc = fadd(fmul(acc, smf[s * 64 + lane]), smf[(nseg + s) * 64 + lane]);
        scores[b] = acc;
    }
}

// grid.x = env groups of 64*4 (C==1) lanes*vec, grid.y = time slabs.
__global__ __launch_bounds__(256) void grpo_broadcast(const float* __restrict__ scores, const uint8_t* __restrict__ m,
                                                      float* __restrict__ adv, int n_chunk, int B, int C, int G,
                                                      float eps, int rows_per_block) {
    // one thread per (b, c) column of the [n_chunk][B*C] matrix; rows are time chunks
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t ncol = (size_t)B * C;
    if (col >= ncol) return;
    const size_t b = col / C;
    const size_t g0 = (b / G) * G;
    float sum = 0.f;
    for (int j = 0; j < G; ++j) sum += scores[g0 + j];
    const float mean = sum / (float)G;
    float ss = 0.f;
    for (int j = 0; j < G; ++j) {
        const float dlt = scores[g0 + j] - mean;
        ss += dlt * dlt;
    }
    const float sd = sqrtf(ss / (float)(G - 1));  // unbiased; G == 1 -> NaN exactly like torch.std
    const float a = fsub(scores[b], mean) / fadd(sd, eps);
    const int k_lo = blockIdx.y * rows_per_block, k_hi = min(n_chunk, k_lo + rows_per_block);
#pragma unroll 4
    for (int k = k_lo; k < k_hi; ++k) {
        const size_t i = (size_t)k * ncol + col;
        adv[i] = fmul(a, m[i] ? 1.f : 0.f);
    }
}


// reward filter (EmbodiedFSDPActor._process_received_rollout_batch, embodied_fsdp_actor_worker.py:235-281): a prompt group
// whose mean (masked) episode reward falls outside [lower, upper] is dropped from the loss.  One block per group:
// reduce, decide, then write the group's mask columns.
__global__ __launch_bounds__(256) void reward_filter_kernel(const float* __restrict__ r, const uint8_t* __restrict__ m,
                                                            uint8_t* __restrict__ out, int n_chunk, int B, int C, int G,
                                                            int out_c, float lower, float upper) {
    __shared__ double scratch[4];
    __shared__ int s_keep;
    const int grp = blockIdx.x;
    const int width = G * C;  // contiguous floats of this group in every time row
    const size_t row_stride = (size_t)B * C;
    const size_t base = (size_t)grp * width;
    double acc[1] = {0.0};
    for (long long i = threadIdx.x; i < (long long)n_chunk * width; i += blockDim.x) {
        const size_t idx = (size_t)(i / width) * row_stride + base + (size_t)(i % width);
        const float v = r[idx];
        acc[0] += (double)(m ? fmul(v, m[idx] ? 1.f : 0.f) : v);
    }
    block_sum<1>(acc, scratch);
    if (threadIdx.x == 0) {
        const float mean = (float)(acc[0] / (double)G);
        s_keep = (mean >= lower && mean <= upper) ? 1 : 0;
    }
    __syncthreads();
    const uint8_t keep = (uint8_t)s_keep;
    const int owidth = G * out_c;
    const size_t orow = (size_t)B * out_c, obase = (size_t)grp * owidth;
    for (long long i = threadIdx.x; i < (long long)n_chunk * owidth; i += blockDim.x) {
        const size_t t = (size_t)(i / owidth), j = (size_t)(i % owidth);
        out[t * orow + obase + j] = m ? (uint8_t)(keep & (m[t * row_stride + base + j] ? 1 : 0)) : keep;
    }
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" int rlx_reward_filter_mask(const float* rewards, const uint8_t* loss_mask, uint8_t* out_mask, int n_chunk,
                                      int batch, int chunk, int group_size, float lower, float upper,
                                      rlx_stream_t stream) {
    RLX_REQUIRE(n_chunk >= 0 && batch >= 0 && chunk >= 1, "rlx_reward_filter_mask: bad sizes");
    RLX_REQUIRE(group_size >= 1 && batch % group_size == 0, "batch %d not divisible by group_size %d", batch, group_size);
    if (batch == 0 || n_chunk == 0) return RLX_OK;
    RLX_REQUIRE(rewards && out_mask, "rlx_reward_filter_mask: NULL argument");
    hipLaunchKernelGGL(reward_filter_kernel, dim3(batch / group_size), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rewards, loss_mask, out_mask, n_chunk, batch, chunk, group_size, loss_mask ? chunk : 1, lower, upper);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

static int launch_scores(const float* rewards, const uint8_t* dones, float* scores, int n_chunk, int batch, int chunk,
                         hipStream_t s) {
    const int T = n_chunk * chunk;
    const int groups = ceil_div(batch, 64);
    int nseg = 1;
    while (nseg < 8 && groups * nseg < 4 * num_cu() && T / (nseg * 2) >= 8) nseg *= 2;
    hipLaunchKernelGGL(grpo_scores, dim3(groups), dim3(64 * nseg), nseg > 1 ? (size_t)2 * nseg * 64 * sizeof(float) : 0, s,
                       rewards, dones, scores, T, batch, chunk);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

static int launch_broadcast(const float* scores, const uint8_t* loss_mask, float* advantages, int n_chunk, int batch,
                            int chunk, int group_size, float eps, hipStream_t s) {
    const size_t ncol = (size_t)batch * chunk;
    const int gx = ceil_div((long long)ncol, 256);
    int gy = 1;
    while (gx * gy < 4 * num_cu() && n_chunk / (gy * 2) >= 4) gy *= 2;
    const int rows = ceil_div(n_chunk, gy);
    hipLaunchKernelGGL(grpo_broadcast, dim3(gx, ceil_div(n_chunk, rows)), dim3(256), 0, s, scores, loss_mask,
                       advantages, n_chunk, batch, chunk, group_size, eps, rows);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_grpo_from_scores(const float* scores, const uint8_t* loss_mask, float* advantages, int n_chunk,
                                    int batch, int chunk, int group_size, float eps, rlx_stream_t stream) {
    RLX_REQUIRE(n_chunk >= 0 && batch >= 0 && chunk >= 1, "rlx_grpo_from_scores: bad sizes");
    RLX_REQUIRE(group_size >= 1 && batch % group_size == 0, "rlx_grpo_from_scores: batch %d %% group_size %d != 0", batch,
                group_size);
    if (batch == 0 || n_chunk == 0) return RLX_OK;
    RLX_REQUIRE(scores && loss_mask && advantages, "rlx_grpo_from_scores: NULL argument");
    return launch_broadcast(scores, loss_mask, advantages, n_chunk, batch, chunk, group_size, eps,
                            static_cast<hipStream_t>(stream));
}

extern "C" int rlx_episode_scores(const float* rewards, const uint8_t* dones, float* scores, int n_chunk, int batch,
                                  int chunk, rlx_stream_t stream) {
    RLX_REQUIRE(n_chunk >= 0 && batch >= 0 && chunk >= 1, "rlx_episode_scores: bad sizes");
    if (batch == 0) return RLX_OK;
    RLX_REQUIRE(scores && (n_chunk == 0 || (rewards && dones)), "rlx_episode_scores: NULL argument");
    return launch_scores(rewards, dones, scores, n_chunk, batch, chunk, static_cast<hipStream_t>(stream));
}

extern "C" int rlx_grpo_group_adv(const float* rewards, const uint8_t* dones, const uint8_t* loss_mask, float* scores,
                                  float* advantages, int n_chunk, int batch, int chunk, int group_size, float eps,
                                  rlx_stream_t stream) {
    RLX_REQUIRE(n_chunk >= 0 && batch >= 0 && chunk >= 1, "rlx_grpo_group_adv: bad sizes");
    RLX_REQUIRE(group_size >= 1 && batch % group_size == 0, "rlx_grpo_group_adv: batch %d %% group_size %d != 0", batch,
                group_size);
    if (batch == 0) return RLX_OK;
    RLX_REQUIRE(rewards && dones && scores && (n_chunk == 0 || (loss_mask && advantages)),
                "rlx_grpo_group_adv: NULL argument (a loss mask is required, advantages.py:118)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = launch_scores(rewards, dones, scores, n_chunk, batch, chunk, s);
    if (rc != RLX_OK || n_chunk == 0) return rc;
    return launch_broadcast(scores, loss_mask, advantages, n_chunk, batch, chunk, group_size, eps, s);
}
