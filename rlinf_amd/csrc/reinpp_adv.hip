// reinpp_adv.hip -- Reinforce++ advantages on reasoning batches in the sequence-major [bsz, seq] layout, gfx950.
//
// Replaces preprocess_reasoning_advantages_inputs (rlinf/algorithms/utils.py:218-219,245-251: transposes to [seq, bsz]),
// compute_reinpp_advantages (rlinf/algorithms/advantages.py:300-364: reward scatter, KL penalty, flip-cumsum-flip,
// masked mean / variance, normalise) and postprocess_reasoning_advantages_outputs (utils.py:265-277: transpose + copy):
//
//   returns   one workgroup per sequence: locate the reward position, suffix-sum r[t] = [t == eos] reward - beta * kl[t]
//             from the right in 1024-element tiles (coalesced 16-byte loads, wave-shuffle scans, f64 carries), write the
//             return-to-go, accumulate (count, sum, sum of squares) of the masked returns -> per-sequence partials
//             (rows of 16-byte aligned length: reinpp_returns_reg_kernel further down -- the same sums from registers)
//   reduce    <= 64 workgroups: per-sequence partials -> group sums
//   normalize every workgroup folds the group sums into mean, rsqrt(max(var, 1e-8)) itself; adv = (ret - mean) * rstd, float4
//
// HBM-bound: 4 (logprob) + 4 (ref) + 1 (mask) read + 4 written, then 4 + 4 for the normalisation = 21 B per token.
//
// The reward position follows the reference as written: "eos" of sequence b is seq-1 minus the index of the FIRST True in
// the mask of sequence bsz-1-b (advantages.py:337-341 flips the batch axis of the [seq, bsz] mask, not the time axis).
// Response masks start with True, for which this is seq-1: the scalar reward sits on the last position of the row.

#include <algorithm>
#include <stdlib.h>

#include "rlx_common.h"

#ifndef RLX_REINPP_NT
#define RLX_REINPP_NT 1  // streaming loads of logprob / ref_logprob, streaming store of the normalised output (+6 %, measured)
#endif

namespace rlx {
namespace {

constexpr int RT = 256;

__device__ __forceinline__ float kl_value(int kind, float lp, float ref) {  // kl_penalty(logprob, ref_logprob), utils.py:26-64
    const float diff = fsub(lp, ref);
    switch (kind) {
        case RLX_KL_K1: return diff;
        case RLX_KL_ABS: return fabsf(diff);
        case RLX_KL_K2: return fmul(0.5f, fmul(diff, diff));
        default: {
            const float kl = fminf(fmaxf(fsub(ref, lp), -20.f), 20.f);
            return fminf(fmaxf(fsub(fsub(expf(kl), kl), 1.f), -10.f), 10.f);
        }
    }
}

__device__ __forceinline__ double wave_incl_scan_from_right(double v, int lane) {
    // inclusive sum over lanes >= this lane (suffix), wave64
#pragma unroll
    for (int off = 1; off < RLX_WAVE; off <<= 1) {
        const double o = __shfl_down(v, off, RLX_WAVE);
        if (lane + off < RLX_WAVE) v += o;
    }
    return v;
}

template <int RTV>  // lanes per sequence; a tile is 4 RTV tokens
__global__ __launch_bounds__(RTV) void reinpp_returns_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ mask,
                                                            const float* __restrict__ logprob, const float* __restrict__ ref,
                                                            int kl_kind, float kl_beta, float* __restrict__ ret,
                                                            double* __restrict__ partials, long long B, long long S, int aligned) {
    constexpr int RT = RTV, TILE = RTV * 4;
    __shared__ long long s_first;
    __shared__ double s_wave[RT / RLX_WAVE];
    __shared__ double s_red[3][RT / RLX_WAVE];
    const long long b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (RLX_WAVE - 1), wave = tid / RLX_WAVE;
    const bool has_kl = kl_beta > 0.f;
    const float* lp = logprob + b * S;
    const float* rf = ref + b * S;
    const uint8_t* m = mask + b * S;
    float* out = ret + b * S;
    const bool vec_ok = aligned && (S % 4 == 0);  // every row then starts 16-byte aligned
    const long long n_tiles = (S + TILE - 1) / TILE;
    // Short rows (one or two tiles): the tile's operands travel in registers one tile AHEAD of the arithmetic -- the loads of
    // the rightmost tile are issued before the mirrored-mask scan below, so a workgroup pays one memory round trip instead of
    // two dependent ones (32768 x 1024: 0.47 -> 0.51 of the HBM peak).  Longer rows load inside the loop: holding the next
    // tile across the scan cost 5 % there (4096 x 8192: 0.69 -> 0.65), the block's other waves already cover the latency.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    struct Tile {
        f32x4 a, c;
        uint32_t mw;
    };
    auto fetch = [&](long long tile) {
        Tile t;
        const long long q = tile * TILE + (long long)tid * 4;
        const long long qc = (vec_ok && q + 4 <= S) ? q : 0;  // clamped, unconditional loads (partial tiles take the scalar path)
        if (has_kl && vec_ok) {
#if RLX_REINPP_NT
            t.a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(lp + qc));
            t.c = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rf + qc));
#else
            t.a = *reinterpret_cast<const f32x4*>(lp + qc);
            t.c = *reinterpret_cast<const f32x4*>(rf + qc);
#endif
        } else {
            t.a = f32x4{0.f, 0.f, 0.f, 0.f};
            t.c = t.a;
        }
        t.mw = vec_ok ? *reinterpret_cast<const uint32_t*>(m + qc) : 0u;
        return t;
    };
    const bool ahead = n_tiles <= 2;
    Tile nxt = fetch(ahead ? n_tiles - 1 : 0);
    const float reward = rewards[b];                       // requested with the tile: not one more round trip further down
    const uint8_t mirror_first = mask[(B - 1 - b) * S];    // response masks start with True: then eos = S - 1, no search
    // ---- reward position (see the header): first True of the mirrored sequence's mask; none -> argmax's 0
    if (tid == 0) s_first = mirror_first ? 0 : S;
    __syncthreads();
    if (!mirror_first) {  // (block-uniform)
        const uint8_t* mm = mask + (B - 1 - b) * S;
        long long first = S;
        if (aligned && S % 4 == 0) {  // four mask bytes per lane and load (bool bytes are 0 / 1: the lowest set bit names the byte)
            const uint32_t* m4 = reinterpret_cast<const uint32_t*>(mm);
            for (long long w = tid; w < S / 4 && first == S; w += RT) {
                const uint32_t x = m4[w];
                if (x) first = w * 4 + (__ffs((int)x) - 1) / 8;
            }
        } else {
            for (long long t = tid; t < S && first == S; t += RT)
                if (mm[t]) first = t;
        }
        // lanes stop at their own first hit; the block minimum is the row's first True
        for (int off = 32; off > 0; off >>= 1) {
            const long long o = __shfl_xor(first, off, RLX_WAVE);
            first = o < first ? o : first;
        }
        if (lane == 0 && first < S) atomicMin(&s_first, first);
        __syncthreads();
    }
    const long long eos = S - 1 - (s_first == S ? 0 : s_first);
    double carry = 0.0;                // sum of r over everything to the right of the current tile
    double cnt = 0.0, sum = 0.0, sq = 0.0;
    for (long long tile = n_tiles - 1; tile >= 0; --tile) {
        const long long t0 = tile * TILE + (long long)tid * 4;
        const Tile cur = ahead ? nxt : fetch(tile);
        if (ahead && tile > 0) nxt = fetch(tile - 1);
        float r[4];
        uint8_t mk[4];
        if (vec_ok && t0 + 4 <= S) {
            if (has_kl) {
                r[0] = -fmul(kl_beta, kl_value(kl_kind, cur.a.x, cur.c.x)), r[1] = -fmul(kl_beta, kl_value(kl_kind, cur.a.y, cur.c.y));
                r[2] = -fmul(kl_beta, kl_value(kl_kind, cur.a.z, cur.c.z)), r[3] = -fmul(kl_beta, kl_value(kl_kind, cur.a.w, cur.c.w));
            } else {
                r[0] = r[1] = r[2] = r[3] = 0.f;
            }
            const uint32_t mw = cur.mw;
            mk[0] = mw & 0xff, mk[1] = (mw >> 8) & 0xff, mk[2] = (mw >> 16) & 0xff, mk[3] = (mw >> 24) & 0xff;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long t = t0 + j;
                const bool in = t < S;
                r[j] = (in && has_kl) ? -fmul(kl_beta, kl_value(kl_kind, lp[t], rf[t])) : 0.f;
                mk[j] = in ? m[t] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (t0 + j == eos) r[j] = has_kl ? fadd(reward, r[j]) : reward;  // r_matrix[eos] = reward, then -= beta * kl
        // suffix sums: within the lane, across the wave, across the block's waves, plus the carry from the tiles to the right
        const double l3 = (double)r[3], l2 = l3 + (double)r[2], l1 = l2 + (double)r[1], l0 = l1 + (double)r[0];
        const double incl = wave_incl_scan_from_right(l0, lane);  // lanes >= this one
        if (lane == 0) s_wave[wave] = incl;
        __syncthreads();
        double right = carry;  // everything right of this lane's 4 elements
        for (int w = wave + 1; w < RT / RLX_WAVE; ++w) right += s_wave[w];
        right += incl - l0;
        double tile_total = 0.0;
        for (int w = 0; w < RT / RLX_WAVE; ++w) tile_total += s_wave[w];
        const float v[4] = {(float)(right + l0), (float)(right + l1), (float)(right + l2), (float)(right + l3)};
        if (vec_ok && t0 + 4 <= S) {
            *reinterpret_cast<float4*>(out + t0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (t0 + j < S) out[t0 + j] = v[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (mk[j]) cnt += 1.0, sum += (double)v[j], sq += (double)v[j] * (double)v[j];
        carry += tile_total;
        __syncthreads();  // s_wave is rewritten by the next tile
    }
    double red[3] = {cnt, sum, sq};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        red[k] = wave_sum(red[k]);
        if (lane == 0) s_red[k][wave] = red[k];
    }
    __syncthreads();
    if (tid < 3) {
        double t = 0.0;
        for (int w = 0; w < RT / RLX_WAVE; ++w) t += s_red[tid][w];
        partials[b * 3 + tid] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Rows of 16-byte aligned length: the returns in REGISTERS, one wavefront per U x 256 tokens (the shape of gae_seq_reg_kernel).
// The kernel above walks a row tile by tile with two workgroup barriers per tile (wave totals through LDS): its waves alternate
// between loading and waiting (0.63-0.70 of the HBM peak).  Here a wave requests ALL operands of its segment first -- U float4s of
// each log-prob array and U mask words per lane --, turns them into per-token rewards, forms each 256-token group's suffix sums
// in f64 on DPP row shifts + lane reads (no LDS), hands its total to the row's other waves through ONE barrier, and writes
// return-to-go and the masked moments from registers.  S waves share a row (S = 1: four rows per workgroup, no barrier).
// ---------------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_or_d(double old, double src) {  // a lane whose source lies outside its row keeps `old`
    const long long o = __double_as_longlong(old), x = __double_as_longlong(src);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)x, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_value_d(double x, int l) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sums over the lanes to the RIGHT of this one (exclusive) and over the whole wave
__device__ __forceinline__ void suffix_sums(double v, int lane, double& excl, double& total) {
    double i = v;
    i += dpp_or_d<0x101>(0.0, i);  // row_shl:1, 2, 4, 8 -- inside the rows of 16 lanes
    i += dpp_or_d<0x102>(0.0, i);
    i += dpp_or_d<0x104>(0.0, i);
    i += dpp_or_d<0x108>(0.0, i);
    const double a0 = lane_value_d(i, 0), a1 = lane_value_d(i, 16), a2 = lane_value_d(i, 32), a3 = lane_value_d(i, 48);
    const double t1 = a2 + a3, t0 = a1 + t1;  // everything right of row 1 / row 0
    const int rowi = lane >> 4;
    const double t = rowi == 3 ? 0.0 : rowi == 2 ? a3 : rowi == 1 ? t1 : t0;
    excl = dpp_or_d<0x101>(0.0, i) + t;
    total = a0 + t0;
}

template <int S, int U>  // waves per row (a power of two <= 16), 256-token groups per wave
__global__ __launch_bounds__(64 * (S < 4 ? 4 : S)) void reinpp_returns_reg_kernel(
    const float* __restrict__ rewards, const uint8_t* __restrict__ mask, const float* __restrict__ logprob,
    const float* __restrict__ ref, int kl_kind, float kl_beta, float* __restrict__ ret, double* __restrict__ partials, long long B,
    int seq) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int WAVES = S < 4 ? 4 : S, ROWS = WAVES / S;
    __shared__ double sTot[WAVES];
    __shared__ double sRed[WAVES][3];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int w = wid % S;  // this wave's segment, counted from the row's end
    const long long row_raw = (long long)blockIdx.x * ROWS + wid / S;
    const bool row_live = row_raw < B;
    const long long row = row_live ? row_raw : B - 1;  // (a dead wave of the last workgroup repeats a row, stores nothing)
    const bool has_kl = kl_beta > 0.f;
    const f32x4* lp4 = reinterpret_cast<const f32x4*>(logprob + row * (long long)seq);
    const f32x4* rf4 = reinterpret_cast<const f32x4*>(ref + row * (long long)seq);
    const uint32_t* m32 = reinterpret_cast<const uint32_t*>(mask + row * (long long)seq);
    f32x4* out4 = reinterpret_cast<f32x4*>(ret + row * (long long)seq);
    const int nq = seq / 4, ngroups = (nq + 63) / 64;  // group k (from the END) holds float4s [nq - 64 (k + 1), nq - 64 k)
    const int per = (ngroups + S - 1) / S;             // <= U
    const int k0 = w * per, k1 = min(ngroups, k0 + per);
    f32x4 a[U], c[U];
    uint32_t mw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // (clamped, unconditional: all in flight together)
        const int q = min(max(nq - 64 * (k0 + u + 1) + lane, 0), nq - 1);
        if (has_kl) {
#if RLX_REINPP_NT
            a[u] = __builtin_nontemporal_load(lp4 + q);
            c[u] = __builtin_nontemporal_load(rf4 + q);
#else
            a[u] = lp4[q];
            c[u] = rf4[q];
#endif
        } else {
            a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            c[u] = a[u];
        }
        mw[u] = m32[q];
    }
    const float reward = rewards[row];
    // ---- reward position (see the header): first True of the mirrored sequence's mask; none -> argmax's 0
    long long first = 0;
    if (!mask[(B - 1 - row) * (long long)seq]) {  // (wave-uniform; response masks start with True: then eos = seq - 1, no search)
        const uint32_t* mm = reinterpret_cast<const uint32_t*>(mask + (B - 1 - row) * (long long)seq);
        first = seq;
        for (int i = lane; i < nq && first == seq; i += 64) {
            const uint32_t x = mm[i];
            if (x) first = (long long)i * 4 + (__ffs((int)x) - 1) / 8;  // bool bytes are 0 / 1: the lowest set bit names the byte
        }
        for (int off = 32; off > 0; off >>= 1) {
            const long long o = __shfl_xor(first, off, 64);
            first = o < first ? o : first;
        }
        if (first == seq) first = 0;
    }
    const long long eos = seq - 1 - first;
    float r[U][4];
    double excl[U], tot[U];
    double wsum = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = k0 + u;
        excl[u] = 0.0, tot[u] = 0.0;
        r[u][0] = r[u][1] = r[u][2] = r[u][3] = 0.f;
        if (k < k1) {  // (wave-uniform)
            const int q = nq - 64 * (k + 1) + lane;
            if (q >= 0) {
                if (has_kl) {
                    r[u][0] = -fmul(kl_beta, kl_value(kl_kind, a[u].x, c[u].x)), r[u][1] = -fmul(kl_beta, kl_value(kl_kind, a[u].y, c[u].y));
                    r[u][2] = -fmul(kl_beta, kl_value(kl_kind, a[u].z, c[u].z)), r[u][3] = -fmul(kl_beta, kl_value(kl_kind, a[u].w, c[u].w));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((long long)q * 4 + j == eos) r[u][j] = has_kl ? fadd(reward, r[u][j]) : reward;  // r_matrix[eos] = reward, then -= beta * kl
            } else {
                mw[u] = 0u;
            }
            const double l0 = (((double)r[u][3] + (double)r[u][2]) + (double)r[u][1]) + (double)r[u][0];
            suffix_sums(l0, lane, excl[u], tot[u]);
            wsum += tot[u];
        } else {
            mw[u] = 0u;
        }
    }
    double carry = 0.0;  // sum of r over everything right of this wave's segment
    if constexpr (S > 1) {
        if (lane == 0) sTot[wid] = wsum;
        __syncthreads();
        const int base = wid - w;
        for (int j = 0; j < w; ++j) carry += sTot[base + j];
    }
    double cnt = 0.0, sum = 0.0, sq = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = k0 + u;
        if (k < k1) {
            const int q = nq - 64 * (k + 1) + lane;
            const double right = carry + excl[u];
            const double l3 = (double)r[u][3], l2 = l3 + (double)r[u][2], l1 = l2 + (double)r[u][1], l0 = l1 + (double)r[u][0];
            const f32x4 v = {(float)(right + l0), (float)(right + l1), (float)(right + l2), (float)(right + l3)};
            if (q >= 0 && row_live) out4[q] = v;
            const uint32_t m = mw[u];
            if (m & 0xffu) cnt += 1.0, sum += (double)v.x, sq += (double)v.x * (double)v.x;
            if (m & 0xff00u) cnt += 1.0, sum += (double)v.y, sq += (double)v.y * (double)v.y;
            if (m & 0xff0000u) cnt += 1.0, sum += (double)v.z, sq += (double)v.z * (double)v.z;
            if (m & 0xff000000u) cnt += 1.0, sum += (double)v.w, sq += (double)v.w * (double)v.w;
            carry += tot[u];
        }
    }
    cnt = wave_sum(cnt), sum = wave_sum(sum), sq = wave_sum(sq);
    if constexpr (S > 1) {
        if (lane == 0) sRed[wid][0] = cnt, sRed[wid][1] = sum, sRed[wid][2] = sq;
        __syncthreads();
        if (w == 0 && lane < 3 && row_live) {
            double t = 0.0;
            for (int j = 0; j < S; ++j) t += sRed[wid + j][lane];
            partials[row * 3 + lane] = t;
        }
    } else if (lane == 0 && row_live) {
        partials[row * 3] = cnt, partials[row * 3 + 1] = sum, partials[row * 3 + 2] = sq;
    }
}

template <int S, int U>
void launch_returns_reg(const float* rewards, const uint8_t* mask, const float* lp, const float* ref, int kl_kind, float kl_beta,
                        float* ret, double* partials, long long B, int seq, hipStream_t st) {
    constexpr int WAVES = S < 4 ? 4 : S, ROWS = WAVES / S;
    hipLaunchKernelGGL((reinpp_returns_reg_kernel<S, U>), dim3((unsigned)((B + ROWS - 1) / ROWS)), dim3(64 * WAVES), 0, st, rewards, mask,
                       lp, ref, kl_kind, kl_beta, ret, partials, B, seq);
}

template <int U>
bool returns_reg(const float* rewards, const uint8_t* mask, const float* lp, const float* ref, int kl_kind, float kl_beta, float* ret,
                 double* partials, long long B, long long seq, hipStream_t st) {
    const long long ngroups = (seq / 4 + 63) / 64, need = (ngroups + U - 1) / U;
    if (need > 16) return false;
    if (need <= 1) launch_returns_reg<1, U>(rewards, mask, lp, ref, kl_kind, kl_beta, ret, partials, B, (int)seq, st);
    else if (need <= 2) launch_returns_reg<2, U>(rewards, mask, lp, ref, kl_kind, kl_beta, ret, partials, B, (int)seq, st);
    else if (need <= 4) launch_returns_reg<4, U>(rewards, mask, lp, ref, kl_kind, kl_beta, ret, partials, B, (int)seq, st);
    else if (need <= 8) launch_returns_reg<8, U>(rewards, mask, lp, ref, kl_kind, kl_beta, ret, partials, B, (int)seq, st);
    else launch_returns_reg<16, U>(rewards, mask, lp, ref, kl_kind, kl_beta, ret, partials, B, (int)seq, st);
    return true;
}

// partials [B][3] -> groups [G][3]: block g sums sequences g, g + G, ... (one single workgroup walking 3 B doubles was
// 50 us at 32768 sequences)
constexpr int MAX_GROUPS = 64;

__global__ __launch_bounds__(RT) void reinpp_reduce_kernel(const double* __restrict__ partials, long long B,
                                                           double* __restrict__ groups) {
    __shared__ double s_red[3][RT / RLX_WAVE];
    double acc[3] = {0.0, 0.0, 0.0};
    for (long long b = (long long)blockIdx.x * RT + threadIdx.x; b < B; b += (long long)gridDim.x * RT) {
        acc[0] += partials[b * 3], acc[1] += partials[b * 3 + 1], acc[2] += partials[b * 3 + 2];
    }
    const int lane = threadIdx.x & (RLX_WAVE - 1), wave = threadIdx.x / RLX_WAVE;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        acc[k] = wave_sum(acc[k]);
        if (lane == 0) s_red[k][wave] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double t = 0.0;
        for (int w = 0; w < RT / RLX_WAVE; ++w) t += s_red[threadIdx.x][w];
        groups[blockIdx.x * 3 + threadIdx.x] = t;
    }
}

// every workgroup folds the (at most 64) group sums itself, in the same order -- no third launch, identical statistics
__global__ __launch_bounds__(RT) void reinpp_normalize_kernel(float* __restrict__ x, long long n, const double* __restrict__ groups,
                                                              int n_groups) {
    __shared__ float s_stats[2];
    if (threadIdx.x == 0) {
        double c = 0.0, s = 0.0, q = 0.0;
        for (int g = 0; g < n_groups; ++g) c += groups[g * 3], s += groups[g * 3 + 1], q += groups[g * 3 + 2];
        // masked_mean of an all-False mask is the (zero) sum itself (utils/utils.py:327-328)
        const double mean = c > 0.0 ? s / c : 0.0;
        double var = c > 0.0 ? q / c - mean * mean : 0.0;
        var = var > 1e-8 ? var : 1e-8;
        s_stats[0] = (float)mean;
        s_stats[1] = (float)(1.0 / sqrt(var));
    }
    __syncthreads();
    const float mean = s_stats[0], rstd = s_stats[1];
    const long long n4 = n / 4;
    const long long stride = (long long)gridDim.x * RT;
    float4* x4 = reinterpret_cast<float4*>(x);
    if (reinterpret_cast<uintptr_t>(x) & 15) {  // a view at an odd offset: element-wise
        for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < n; i += stride) x[i] = fmul(fsub(x[i], mean), rstd);
        return;
    }
    for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < n4; i += stride) {
        float4 v = x4[i];
        v.x = fmul(fsub(v.x, mean), rstd), v.y = fmul(fsub(v.y, mean), rstd);
        v.z = fmul(fsub(v.z, mean), rstd), v.w = fmul(fsub(v.w, mean), rstd);
#if RLX_REINPP_NT
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 o = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(x) + i);
#else
        x4[i] = v;
#endif
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * RT + threadIdx.x; i < n; i += stride) x[i] = fmul(fsub(x[i], mean), rstd);
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_reinpp_workspace_bytes(int64_t bsz) {
    return (size_t)(bsz > 0 ? bsz : 0) * 3 * sizeof(double) + (size_t)MAX_GROUPS * 3 * sizeof(double) + 256;
}

extern "C" int rlx_reinpp_seq_adv(const float* rewards, const uint8_t* loss_mask, const float* logprob, const float* ref_logprob,
                                  int kl_type, float kl_beta, float* advantages, int64_t bsz, int64_t seq, void* workspace,
                                  size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(bsz >= 0 && seq >= 0, "rlx_reinpp_seq_adv: negative size");
    if (bsz == 0 || seq == 0) return RLX_OK;
    RLX_REQUIRE(rewards && loss_mask && advantages && workspace, "rlx_reinpp_seq_adv: NULL argument");
    RLX_REQUIRE(!(kl_beta > 0.f) || (logprob && ref_logprob && kl_type >= RLX_KL_K1 && kl_type <= RLX_KL_K3),
                "rlx_reinpp_seq_adv: kl_beta > 0 needs logprob, ref_logprob and a kl type (got %d)", kl_type);
    if (workspace_bytes < rlx_reinpp_workspace_bytes(bsz)) {
        set_error("rlx_reinpp_seq_adv: workspace too small");
        return RLX_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* partials = static_cast<double*>(workspace);
    double* groups = partials + (size_t)bsz * 3;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(loss_mask) | reinterpret_cast<uintptr_t>(advantages) |
                           (kl_beta > 0.f ? (reinterpret_cast<uintptr_t>(logprob) | reinterpret_cast<uintptr_t>(ref_logprob)) : 0);
    // (A one-wave-per-row variant for rows of <= 2048 tokens -- no LDS, no barrier, 16 / 32 tokens per lane in registers -- was
    //  measured and dropped: 0.41 / 0.23 of the HBM peak at 32768 x 1024 / 16384 x 2048 against 0.51 / 0.57 for this kernel.)
    // 16-byte aligned rows: the register kernel (RLX_REINPP_REG=0: the tile walk for every shape; 4 / 8: groups per wave)
    const int reg_mode = [] { const char* e = getenv("RLX_REINPP_REG"); return e ? atoi(e) : -1; }();
    bool done = false;
    if ((bits & 15) == 0 && seq % 4 == 0 && seq < (1ll << 31) && reg_mode != 0) {
        if (reg_mode == 8 || (reg_mode != 4 && seq > 16384))
            done = returns_reg<8>(rewards, loss_mask, logprob, ref_logprob, kl_type, kl_beta, advantages, partials, bsz, seq, st);
        else
            done = returns_reg<4>(rewards, loss_mask, logprob, ref_logprob, kl_type, kl_beta, advantages, partials, bsz, seq, st);
    }
    if (!done) {
        // lanes per sequence (a tile is 4 x lanes tokens), measured on three shapes (profiles/r02_reinpp_lanes_sweep.txt):
        // rows of <= 2048 tokens want 64 lanes (0.65-0.66 of the HBM peak against 0.58-0.62 with 256: a row is 4-8 short tiles
        // of one wave each, no cross-wave hand-off, and many rows share a CU), rows of >= 4096 tokens 256 lanes (0.69 against 0.63)
#ifdef RLX_DEV_VARIANTS
        static const int forced = getenv("RLX_REINPP_RT") ? atoi(getenv("RLX_REINPP_RT")) : 0;  // development override
#else
        constexpr int forced = 0;
#endif
        const int rtv = forced ? forced : (seq <= 2048 ? 64 : (seq < 4096 ? 128 : 256));
        if (rtv == 64) hipLaunchKernelGGL(reinpp_returns_kernel<64>, dim3((unsigned)bsz), dim3(64), 0, st, rewards, loss_mask, logprob, ref_logprob, kl_type,
                       kl_beta, advantages, partials, (long long)bsz, (long long)seq, (int)((bits & 15) == 0));
        else if (rtv == 128) hipLaunchKernelGGL(reinpp_returns_kernel<128>, dim3((unsigned)bsz), dim3(128), 0, st, rewards, loss_mask, logprob, ref_logprob, kl_type,
                       kl_beta, advantages, partials, (long long)bsz, (long long)seq, (int)((bits & 15) == 0));
        else if (rtv == 512) hipLaunchKernelGGL(reinpp_returns_kernel<512>, dim3((unsigned)bsz), dim3(512), 0, st, rewards, loss_mask, logprob, ref_logprob, kl_type,
                       kl_beta, advantages, partials, (long long)bsz, (long long)seq, (int)((bits & 15) == 0));
        else hipLaunchKernelGGL(reinpp_returns_kernel<256>, dim3((unsigned)bsz), dim3(256), 0, st, rewards, loss_mask, logprob, ref_logprob, kl_type,
                       kl_beta, advantages, partials, (long long)bsz, (long long)seq, (int)((bits & 15) == 0));
    }
    RLX_LAUNCH_CHECK();
    const int n_groups = (int)std::min<long long>(MAX_GROUPS, (bsz + RT - 1) / RT);
    hipLaunchKernelGGL(reinpp_reduce_kernel, dim3(n_groups), dim3(RT), 0, st, partials, (long long)bsz, groups);
    RLX_LAUNCH_CHECK();
    const long long n = (long long)bsz * seq;
    long long blocks = (n / 4 + RT - 1) / RT;
    const long long cap = (long long)num_cu() * 16;
    blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
    hipLaunchKernelGGL(reinpp_normalize_kernel, dim3((unsigned)blocks), dim3(RT), 0, st, advantages, n, groups, n_groups);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
