// gae_seq.hip -- GAE for reasoning (LLM PPO) batches in their own [bsz, seq] layout, gfx950.
//
// Replaces, for adv_type = "gae" on task_type = "reasoning":
//   preprocess_reasoning_advantages_inputs   rlinf/algorithms/utils.py:177-262  (transposes to [seq, bsz], a [seq, bsz]
//       reward matrix that is zero except its last row, a zero bootstrap row under the values, dones = last row only)
//   compute_gae_advantages_and_returns       rlinf/algorithms/advantages.py:24-86 (a Python loop of seq iterations)
//   postprocess_reasoning_advantages_outputs rlinf/algorithms/utils.py:265-277  (transposes back, contiguous copies)
// With that shaping the recurrence of sequence b is, for t = seq-1 .. 0,
//     delta_t = (t == seq-1 ? r_b : gamma * V[b,t+1]) - V[b,t]        g_t = delta_t + (gamma*lambda) * g_{t+1},  g_seq = 0
//     returns = g + V,   advantages = returns - V
// i.e. a first-order linear recurrence ALONG THE CONTIGUOUS AXIS: a wavefront segmented scan.  Rows whose length is a multiple
// of four tokens (16-byte aligned rows, up to 32 768 tokens) take gae_seq_reg_kernel: one wave per 2048 tokens, registers only,
// one barrier (0.72-0.77 of the HBM peak at 4096 x 8192, 0.79-0.85 where the arrays fit the memory-side cache).  Every other
// shape takes the LDS kernels: one workgroup per sequence walks it in segments of up to SEG tokens from the end; a segment is
// staged in LDS with coalesced loads, every lane owns a contiguous chunk, reduces it to the affine map g_in -> K * g_in + G, the
// 256 maps are composed right-to-left through LDS, and each lane then replays its chunk sequentially from its exact carry-in,
// writing through LDS so that the stores are coalesced too.  12 B per token (values 4 + advantages 4 + returns 4).
//
// The replay uses the reference's own operations in its order, so a result differs from the sequential loop only
// through the carry-in of its chunk (affine composition instead of a chain: a few f32 ulp).

#include "rlx_common.h"

#ifndef RLX_GAESEQ_NT
#define RLX_GAESEQ_NT 0  // dev switch: streaming (non-temporal) loads of the values and stores of both outputs
#endif

#include <stdlib.h>

namespace rlx {
namespace {

inline int dev_variant_gaeseq() {  // RLX_GAESEQ_VARIANT=1 / 2 -> the round-1 walk / one segment per workgroup with decoupled look-back
#ifdef RLX_DEV_VARIANTS             // (both measured slower; tested variants of development builds)
    const char* e = getenv("RLX_GAESEQ_VARIANT");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

__device__ __forceinline__ int pad(int i) { return i + (i >> 5); }  // chunk starts land in distinct banks

template <int ST, int SEG>  // threads per sequence; tokens staged in LDS per pass (2 * (SEG + SEG/32) floats)
__global__ __launch_bounds__(ST) void gae_seq_kernel(const float* __restrict__ values, const float* __restrict__ rewards,
                                                     float* __restrict__ adv, float* __restrict__ ret, int seq,
                                                     float gamma, float gamma_lambda) {
    extern __shared__ float lds[];
    float* sv = lds;                      // values of the segment, plus one look-ahead value
    const int seg_cap = seq < SEG ? seq : SEG;
    float* sg = lds + pad(seg_cap + 1) + 1;  // g of the segment (LDS is sized for the longest segment of THIS launch)
    __shared__ float sK[ST], sG[ST], sIn[ST];
    const long long row = blockIdx.x;
    const float* v = values + row * (long long)seq;
    const float r = rewards[row];
    float carry = 0.f;                    // g at the first token AFTER the current segment
    for (int hi = seq; hi > 0; hi -= SEG) {
        const int lo = hi > SEG ? hi - SEG : 0, len = hi - lo;
        // stage the segment: every global load is unconditional (clamped address) and issued before the first LDS store
        // -- a bounds test around each load would put an s_waitcnt vmcnt(0) between them (2 us of HBM latency apiece)
        const bool vec = ((reinterpret_cast<uintptr_t>(v + lo) & 15u) == 0) && (len % 4 == 0);
        if (vec) {
            const int nq = len / 4;
            const float4* v4 = reinterpret_cast<const float4*>(v + lo);
            constexpr int QI = 4;  // 16-byte loads in flight per lane
            for (int q0 = 0; q0 < nq; q0 += QI * ST) {
                float4 q[QI];
#pragma unroll
                for (int k = 0; k < QI; ++k) {
#if RLX_GAESEQ_NT
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v4) + min(q0 + k * ST + (int)threadIdx.x, nq - 1));
                    q[k] = make_float4(t.x, t.y, t.z, t.w);
#else
                    q[k] = v4[min(q0 + k * ST + (int)threadIdx.x, nq - 1)];
#endif
                }
#pragma unroll
                for (int k = 0; k < QI; ++k) {
                    const int qi = q0 + k * ST + threadIdx.x;
                    if (qi < nq) {
                        sv[pad(4 * qi)] = q[k].x, sv[pad(4 * qi + 1)] = q[k].y;
                        sv[pad(4 * qi + 2)] = q[k].z, sv[pad(4 * qi + 3)] = q[k].w;
                    }
                }
            }
        } else {
            constexpr int SI = 8;
            for (int i0 = 0; i0 < len; i0 += SI * ST) {
                float x[SI];
#pragma unroll
                for (int k = 0; k < SI; ++k) x[k] = v[lo + min(i0 + k * ST + (int)threadIdx.x, len - 1)];
#pragma unroll
                for (int k = 0; k < SI; ++k) {
                    const int i = i0 + k * ST + threadIdx.x;
                    if (i < len) sv[pad(i)] = x[k];
                }
            }
        }
        if (threadIdx.x == 0) sv[pad(len)] = hi < seq ? v[hi] : 0.f;  // the look-ahead V[hi] (0 past the end)
        __syncthreads();
        const int per = (len + ST - 1) / ST;
        // lane tid owns the chunk (ST-1-tid): lane order = the order in which the recurrence visits the chunks
        const int chunk = ST - 1 - (int)threadIdx.x;
        const int c0 = min(len, chunk * per), c1 = min(len, c0 + per);
        // chunk -> affine map g_in -> K * g_in + G
        float K = 1.f, G = 0.f;
        {
            float vnext = sv[pad(c1)];
            for (int i = c1 - 1; i >= c0; --i) {
                const int t = lo + i;
                const float vi = sv[pad(i)];
                const float nxt = (t == seq - 1) ? r : fmul(gamma, vnext);
                const float delta = fsub(nxt, vi);
                const float k = (t == seq - 1) ? 0.f : gamma_lambda;  // ~dones[seq] cuts the recurrence at the end
                G = fadd(delta, fmul(k, G));
                K = fmul(K, k);
                vnext = vi;
            }
        }
        // exclusive scan of the maps in lane order (later-applied map on the left): wave shuffles, then the 4 wave totals
        float iK = K, iG = G;
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float oK = __shfl_up(iK, off, 64), oG = __shfl_up(iG, off, 64);
            if (lane >= off) {
                iG = fadd(iG, fmul(iK, oG));  // mine after theirs: K_m K_o g + (K_m G_o + G_m)
                iK = fmul(iK, oK);
            }
        }
        if (lane == 63) sK[wid] = iK, sG[wid] = iG;
        __syncthreads();
        float bK = 1.f, bG = 0.f;  // composition of all earlier waves
        for (int w = 0; w < wid; ++w) {
            bG = fadd(sG[w], fmul(sK[w], bG));
            bK = fmul(sK[w], bK);
        }
        float eK = __shfl_up(iK, 1, 64), eG = __shfl_up(iG, 1, 64);  // exclusive within the wave
        if (lane == 0) eK = 1.f, eG = 0.f;
        const float tK = fmul(eK, bK), tG = fadd(eG, fmul(eK, bG));   // everything visited before this chunk
        const float g_in = fadd(tG, fmul(tK, carry));
        if (threadIdx.x == ST - 1) {  // the last lane's inclusive map gives g at the segment's first token
            const float aK = fmul(iK, bK), aG = fadd(iG, fmul(iK, bG));
            sIn[0] = fadd(aG, fmul(aK, carry));
        }
        __syncthreads();
        {
            float g = g_in, vnext = sv[pad(c1)];
            for (int i = c1 - 1; i >= c0; --i) {
                const int t = lo + i;
                const float vi = sv[pad(i)];
                const float nxt = (t == seq - 1) ? r : fmul(gamma, vnext);
                const float delta = fsub(nxt, vi);
                const float k = (t == seq - 1) ? 0.f : gamma_lambda;
                g = fadd(delta, fmul(k, g));
                sg[pad(i)] = g;
                vnext = vi;
            }
        }
        carry = sIn[0];
        __syncthreads();
        float* ro = ret + row * (long long)seq + lo;
        float* ao = adv + row * (long long)seq + lo;
        if (vec && ((reinterpret_cast<uintptr_t>(ro) | reinterpret_cast<uintptr_t>(ao)) & 15u) == 0) {
            for (int qi = threadIdx.x; qi < len / 4; qi += ST) {
                float rt[4], at[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gi = sg[pad(4 * qi + j)], vi = sv[pad(4 * qi + j)];
                    rt[j] = fadd(gi, vi);       // returns[t] = gae + values[t]
                    at[j] = fsub(rt[j], vi);    // advantages = returns - values[:-1]
                }
#if RLX_GAESEQ_NT
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 r4 = {rt[0], rt[1], rt[2], rt[3]}, a4 = {at[0], at[1], at[2], at[3]};
                __builtin_nontemporal_store(r4, reinterpret_cast<f32x4*>(ro) + qi);
                __builtin_nontemporal_store(a4, reinterpret_cast<f32x4*>(ao) + qi);
#else
                reinterpret_cast<float4*>(ro)[qi] = make_float4(rt[0], rt[1], rt[2], rt[3]);
                reinterpret_cast<float4*>(ao)[qi] = make_float4(at[0], at[1], at[2], at[3]);
#endif
            }
        } else {
            for (int i = threadIdx.x; i < len; i += ST) {
                const float gi = sg[pad(i)], vi = sv[pad(i)];
                const float rt = fadd(gi, vi);
                ro[i] = rt;
                ao[i] = fsub(rt, vi);
            }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Long sequences: ONE SEGMENT PER WORKGROUP with a decoupled look-back for the carry.
// The kernel above walks a sequence's segments one after the other inside one workgroup of 512 lanes holding 66 KB of LDS: at
// most two such workgroups fit a CU and each alternates load / scan / store phases, so the memory pipes idle most of the time
// (0.57-0.63 of the HBM peak).  Here a row of `seq` tokens is cut into nseg = ceil(seq / SEG) segments and every segment is its
// own 128-lane workgroup (17 KB of LDS: nine per CU, their phases interleave).  Segment j (counted from the END of the row, so
// that within a row the workgroups are dispatched in the order the recurrence runs) needs g at the first token to its right:
//   * it publishes its aggregate map (K_j, G_j): g_in -> K_j g_in + G_j, as soon as its chunk maps are composed;
//   * it looks back over j-1, j-2, ...: a neighbour that already knows its own carry-in has published g_first (done); one that
//     has only its aggregate contributes that map and the walk goes on (segment 0's carry-in is 0);
//   * after its replay it publishes g_first itself.
// A publication is ONE 64-bit agent-scope atomic store of (value, 1.0f): payload and validity travel in the same word, written
// through to the memory side, so no fence (= no L2 write-back) is needed and a NaN payload is still a valid publication; the
// workspace is memset to 0xFF before the launch.  Workgroups only ever wait for workgroups with a LOWER linear index, which the
// dispatcher has already started: forward progress is guaranteed (and every wait is bounded all the same: a hang would cost
// a GPU, a NaN row is merely wrong).
// ---------------------------------------------------------------------------------------------------------------------------
struct Slot {
    unsigned long long k, g;  // the segment's aggregate map, each as (value, 1.0f)
    unsigned long long fin;   // (g at the segment's first token, 1.0f)
    unsigned long long pad_;
};

__device__ __forceinline__ unsigned long long publish_word(float v) {
    return (unsigned long long)__float_as_uint(v) | ((unsigned long long)0x3F800000u << 32);
}
__device__ __forceinline__ bool word_valid(unsigned long long w) { return (unsigned)(w >> 32) == 0x3F800000u; }
__device__ __forceinline__ float word_value(unsigned long long w) { return __uint_as_float((unsigned)w); }

template <int ST, int SEG>
__global__ __launch_bounds__(ST) void gae_seq_lb_kernel(const float* __restrict__ values, const float* __restrict__ rewards,
                                                        float* __restrict__ adv, float* __restrict__ ret, int seq, int nseg,
                                                        float gamma, float gamma_lambda, Slot* __restrict__ slots) {
    extern __shared__ float lds[];
    float* sv = lds;
    float* sg = lds + pad(SEG + 1) + 1;
    __shared__ float sK[ST / 64], sG[ST / 64], sIn[2];
    const long long row = blockIdx.x / nseg;
    const int j = blockIdx.x % nseg;  // segment index from the END of the row
    const int hi = seq - j * SEG, lo = hi > SEG ? hi - SEG : 0, len = hi - lo;
    const float* v = values + row * (long long)seq;
    const float r = rewards[row];
    Slot* my = slots + row * nseg + j;
    // ---- stage (same clamped, unconditional loads as above) --------------------------------------------------------------
    const bool vec = ((reinterpret_cast<uintptr_t>(v + lo) & 15u) == 0) && (len % 4 == 0);
    if (vec) {
        const int nq = len / 4;
        const float4* v4 = reinterpret_cast<const float4*>(v + lo);
        constexpr int QI = 4;
        for (int q0 = 0; q0 < nq; q0 += QI * ST) {
            float4 q[QI];
#pragma unroll
            for (int k = 0; k < QI; ++k) q[k] = v4[min(q0 + k * ST + (int)threadIdx.x, nq - 1)];
#pragma unroll
            for (int k = 0; k < QI; ++k) {
                const int qi = q0 + k * ST + threadIdx.x;
                if (qi < nq) {
                    sv[pad(4 * qi)] = q[k].x, sv[pad(4 * qi + 1)] = q[k].y;
                    sv[pad(4 * qi + 2)] = q[k].z, sv[pad(4 * qi + 3)] = q[k].w;
                }
            }
        }
    } else {
        constexpr int SI = 8;
        for (int i0 = 0; i0 < len; i0 += SI * ST) {
            float x[SI];
#pragma unroll
            for (int k = 0; k < SI; ++k) x[k] = v[lo + min(i0 + k * ST + (int)threadIdx.x, len - 1)];
#pragma unroll
            for (int k = 0; k < SI; ++k) {
                const int i = i0 + k * ST + threadIdx.x;
                if (i < len) sv[pad(i)] = x[k];
            }
        }
    }
    if (threadIdx.x == 0) sv[pad(len)] = hi < seq ? v[hi] : 0.f;
    __syncthreads();
    const int per = (len + ST - 1) / ST;
    const int chunk = ST - 1 - (int)threadIdx.x;
    const int c0 = min(len, chunk * per), c1 = min(len, c0 + per);
    float K = 1.f, G = 0.f;
    {
        float vnext = sv[pad(c1)];
        for (int i = c1 - 1; i >= c0; --i) {
            const int t = lo + i;
            const float vi = sv[pad(i)];
            const float nxt = (t == seq - 1) ? r : fmul(gamma, vnext);
            const float delta = fsub(nxt, vi);
            const float k = (t == seq - 1) ? 0.f : gamma_lambda;
            G = fadd(delta, fmul(k, G));
            K = fmul(K, k);
            vnext = vi;
        }
    }
    float iK = K, iG = G;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float oK = __shfl_up(iK, off, 64), oG = __shfl_up(iG, off, 64);
        if (lane >= off) {
            iG = fadd(iG, fmul(iK, oG));
            iK = fmul(iK, oK);
        }
    }
    if (lane == 63) sK[wid] = iK, sG[wid] = iG;
    __syncthreads();
    float bK = 1.f, bG = 0.f;
    for (int w = 0; w < wid; ++w) {
        bG = fadd(sG[w], fmul(sK[w], bG));
        bK = fmul(sK[w], bK);
    }
    float eK = __shfl_up(iK, 1, 64), eG = __shfl_up(iG, 1, 64);
    if (lane == 0) eK = 1.f, eG = 0.f;
    const float tK = fmul(eK, bK), tG = fadd(eG, fmul(eK, bG));  // everything this segment visits before this chunk
    // ---- the last lane holds the segment's aggregate: publish it, then resolve the carry by looking back ------------------
    if (threadIdx.x == ST - 1) {
        const float aK = fmul(iK, bK), aG = fadd(iG, fmul(iK, bG));
        float carry = 0.f;
        if (nseg > 1) {
            __hip_atomic_store(&my->k, publish_word(aK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&my->g, publish_word(aG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float accK = 1.f, accG = 0.f;  // maps of the segments walked so far, nearest applied LAST: carry = accK * x + accG
            bool done = j == 0;
            for (int q = j - 1; !done; --q) {
                const Slot* s = slots + row * nseg + q;
                for (int spin = 0;; ++spin) {
                    const unsigned long long f = __hip_atomic_load(&s->fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (word_valid(f)) {  // the neighbour knows its own carry-in: g at its first token is final
                        carry = fadd(accG, fmul(accK, word_value(f)));
                        done = true;
                        break;
                    }
                    const unsigned long long wk = __hip_atomic_load(&s->k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long wg = __hip_atomic_load(&s->g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (word_valid(wk) && word_valid(wg)) {  // only its aggregate so far: fold it in, keep walking to the right
                        accG = fadd(accG, fmul(accK, word_value(wg)));
                        accK = fmul(accK, word_value(wk));
                        if (q == 0) {  // the row's last segment starts from g = 0
                            carry = accG;
                            done = true;
                        }
                        break;
                    }
                    if (spin > (1 << 22)) {  // cannot happen with in-order dispatch; never hang the GPU over it
                        carry = __uint_as_float(0x7FC00000u);
                        done = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        sIn[0] = carry;
        sIn[1] = fadd(aG, fmul(aK, carry));  // g at this segment's first token
    }
    __syncthreads();
    const float carry = sIn[0];
    if (threadIdx.x == ST - 1 && nseg > 1)
        __hip_atomic_store(&my->fin, publish_word(sIn[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        float g = fadd(tG, fmul(tK, carry)), vnext = sv[pad(c1)];
        for (int i = c1 - 1; i >= c0; --i) {
            const int t = lo + i;
            const float vi = sv[pad(i)];
            const float nxt = (t == seq - 1) ? r : fmul(gamma, vnext);
            const float delta = fsub(nxt, vi);
            const float k = (t == seq - 1) ? 0.f : gamma_lambda;
            g = fadd(delta, fmul(k, g));
            sg[pad(i)] = g;
            vnext = vi;
        }
    }
    __syncthreads();
    float* ro = ret + row * (long long)seq + lo;
    float* ao = adv + row * (long long)seq + lo;
    if (vec && ((reinterpret_cast<uintptr_t>(ro) | reinterpret_cast<uintptr_t>(ao)) & 15u) == 0) {
        for (int qi = threadIdx.x; qi < len / 4; qi += ST) {
            float rt[4], at[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float gi = sg[pad(4 * qi + jj)], vi = sv[pad(4 * qi + jj)];
                rt[jj] = fadd(gi, vi);
                at[jj] = fsub(rt[jj], vi);
            }
            reinterpret_cast<float4*>(ro)[qi] = make_float4(rt[0], rt[1], rt[2], rt[3]);
            reinterpret_cast<float4*>(ao)[qi] = make_float4(at[0], at[1], at[2], at[3]);
        }
    } else {
        for (int i = threadIdx.x; i < len; i += ST) {
            const float gi = sg[pad(i)], vi = sv[pad(i)];
            const float rt = fadd(gi, vi);
            ro[i] = rt;
            ao[i] = fsub(rt, vi);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Rows of 16-byte aligned length: REGISTERS ONLY, one wavefront per 2048-token segment.
// The kernels above stage a segment in LDS so that a lane can own a contiguous chunk: load / scan / replay / store phases with
// a workgroup barrier between each, during which that workgroup's memory pipes idle (0.55-0.63 of the HBM peak).  Here a lane
// keeps the float4s it loaded: 64 lanes x 4 tokens = one 256-token group per load instruction, a wave requests ALL (up to
// eight) groups of its segment before it touches any.  S waves of a workgroup share a row (S = 1: four rows per workgroup),
// segment 0 at the row's END.
//   pass 1 (no carry needed): per group, 4 tokens per lane -> the affine map g_in -> K g_in + G, a suffix composition over the
//     64 lanes (DPP row shifts + lane reads: no LDS) -> every lane's exclusive map and the group's aggregate; the groups' aggregates composed
//     give the segment's, which goes to LDS;
//   ONE barrier; a wave chains the aggregates of the segments to its right from g = 0: its carry-in;
//   pass 2: per group, the carry chained through the groups' aggregates, the lane's g_in from its exclusive map, the replay
//     with the reference's own operations in its order, two float4 stores.
// Every byte is read once and written once, each wave is one batch of loads and one batch of stores.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int WV_U = 8;  // groups (256 tokens each) per wave

__device__ __forceinline__ float uniform(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ float lane_value(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
// DPP move: a lane whose source lies outside its row (row_shl) / the wave (wave_shl) keeps `old`
template <int CTRL>
__device__ __forceinline__ float dpp_or(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_ROW_SHL = 0x100, DPP_WAVE_SHL1 = 0x130;

// Suffix composition of 64 per-lane maps g -> K g + G (the lane on the LEFT applies last) on the VALU: four DPP row shifts inside
// the rows of 16 lanes, the three row aggregates by v_readlane.  -> the lane's EXCLUSIVE map (lanes lane + 1 .. 63) and the
// wave's aggregate.  (The first form used six ds_bpermute steps per value: 15 LDS-pipe round trips per group on the critical
// path of every wave.)  Identity is composed as (1, 0): K is a product of finite constants, so K * 0 never makes a NaN.
__device__ __forceinline__ void suffix_maps(float K, float G, int lane, float& eK, float& eG, float& aK, float& aG) {
    float iK = K, iG = G;
#define RLX_STEP(OFF)                                                                  \
    {                                                                                  \
        const float oK = dpp_or<DPP_ROW_SHL + OFF>(1.f, iK), oG = dpp_or<DPP_ROW_SHL + OFF>(0.f, iG); \
        iG = fadd(iG, fmul(iK, oG));                                                   \
        iK = fmul(iK, oK);                                                             \
    }
    RLX_STEP(1) RLX_STEP(2) RLX_STEP(4) RLX_STEP(8)
#undef RLX_STEP
    const float k0 = lane_value(iK, 0), g0 = lane_value(iG, 0), k1 = lane_value(iK, 16), g1 = lane_value(iG, 16);
    const float k2 = lane_value(iK, 32), g2 = lane_value(iG, 32), k3 = lane_value(iK, 48), g3 = lane_value(iG, 48);
    // everything right of row 2 / 1 / 0
    const float t1K = fmul(k2, k3), t1G = fadd(g2, fmul(k2, g3));
    const float t0K = fmul(k1, t1K), t0G = fadd(g1, fmul(k1, t1G));
    const int rowi = lane >> 4;
    const float tK = rowi == 3 ? 1.f : rowi == 2 ? k3 : rowi == 1 ? t1K : t0K;
    const float tG = rowi == 3 ? 0.f : rowi == 2 ? g3 : rowi == 1 ? t1G : t0G;
    const float xK = dpp_or<DPP_ROW_SHL + 1>(1.f, iK), xG = dpp_or<DPP_ROW_SHL + 1>(0.f, iG);  // in-row exclusive
    eK = fmul(xK, tK);
    eG = fadd(xG, fmul(xK, tG));
    aK = fmul(k0, t0K);
    aG = fadd(g0, fmul(k0, t0G));
}

template <int S, bool NT>  // waves per row (a power of two <= 16)
__global__ __launch_bounds__(64 * (S < 4 ? 4 : S)) void gae_seq_reg_kernel(const float* __restrict__ values,
                                                                          const float* __restrict__ rewards, float* __restrict__ adv,
                                                                          float* __restrict__ ret, long long bsz, int seq, float gamma,
                                                                          float gamma_lambda) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int WAVES = S < 4 ? 4 : S, ROWS = WAVES / S;
    __shared__ float sK[WAVES], sG[WAVES];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int w = wid % S;  // this wave's segment, counted from the row's end
    const long long row_raw = (long long)blockIdx.x * ROWS + wid / S;
    const bool row_live = row_raw < bsz;
    const long long row = row_live ? row_raw : bsz - 1;  // (a dead wave of the last workgroup repeats a row, stores nothing)
    const f32x4* v4 = reinterpret_cast<const f32x4*>(values + row * (long long)seq);
    f32x4* r4 = reinterpret_cast<f32x4*>(ret + row * (long long)seq);
    f32x4* a4 = reinterpret_cast<f32x4*>(adv + row * (long long)seq);
    const int nq = seq / 4, ngroups = (nq + 63) / 64;  // group k (from the END) holds float4s [nq - 64 (k + 1), nq - 64 k)
    const int per = (ngroups + S - 1) / S;             // <= WV_U
    const int k0 = w * per, k1 = min(ngroups, k0 + per);
    f32x4 x[WV_U];
#pragma unroll
    for (int u = 0; u < WV_U; ++u) {  // (clamped, unconditional: all in flight together)
        const int q = min(max(nq - 64 * (k0 + u + 1) + lane, 0), nq - 1);
        if constexpr (NT) x[u] = __builtin_nontemporal_load(v4 + q);
        else x[u] = v4[q];
    }
    const float r = rewards[row];
    // V of the first token right of this segment (segment 0: the row ends there)
    const float vseg = k0 > 0 && k0 < ngroups ? values[row * (long long)seq + (long long)(nq - 64 * k0) * 4] : 0.f;
    float eK[WV_U], eG[WV_U], aK[WV_U], aG[WV_U];
    float wK = 1.f, wG = 0.f;
#pragma unroll
    for (int u = 0; u < WV_U; ++u) {
        const int k = k0 + u;
        aK[u] = 1.f, aG[u] = 0.f, eK[u] = 1.f, eG[u] = 0.f;
        if (k < k1) {  // (wave-uniform)
            const bool live = nq - 64 * (k + 1) + lane >= 0;
            const float vright = u == 0 ? vseg : uniform(x[u > 0 ? u - 1 : 0].x);  // (read with the whole wave active)
            const float vn = dpp_or<DPP_WAVE_SHL1>(vright, x[u].x);  // V of the next lane's first token; lane 63: the next group's
            const bool last = k == 0 && lane == 63;  // this lane's fourth token is the row's last: reward, recurrence cut
            const float k3 = last ? 0.f : gamma_lambda;
            const float d3 = fsub(last ? r : fmul(gamma, vn), x[u].w);
            const float d2 = fsub(fmul(gamma, x[u].w), x[u].z);
            const float d1 = fsub(fmul(gamma, x[u].z), x[u].y);
            const float d0 = fsub(fmul(gamma, x[u].y), x[u].x);
            float K = k3, G = d3;
            G = fadd(d2, fmul(gamma_lambda, G)), K = fmul(K, gamma_lambda);
            G = fadd(d1, fmul(gamma_lambda, G)), K = fmul(K, gamma_lambda);
            G = fadd(d0, fmul(gamma_lambda, G)), K = fmul(K, gamma_lambda);
            if (!live) K = 1.f, G = 0.f;
            suffix_maps(K, G, lane, eK[u], eG[u], aK[u], aG[u]);
            wG = fadd(aG[u], fmul(aK[u], wG));
            wK = fmul(aK[u], wK);
        }
    }
    float carry = 0.f;  // g at the first token right of the segment
    if constexpr (S > 1) {
        if (lane == 0) sK[wid] = wK, sG[wid] = wG;
        __syncthreads();
        const int base = wid - w;
        for (int j = 0; j < w; ++j) carry = fadd(sG[base + j], fmul(sK[base + j], carry));
    }
#pragma unroll
    for (int u = 0; u < WV_U; ++u) {
        const int k = k0 + u;
        if (k < k1) {
            const int q = nq - 64 * (k + 1) + lane;
            const float vright = u == 0 ? vseg : uniform(x[u > 0 ? u - 1 : 0].x);  // (read with the whole wave active)
            const float vn = dpp_or<DPP_WAVE_SHL1>(vright, x[u].x);  // V of the next lane's first token; lane 63: the next group's
            const bool last = k == 0 && lane == 63;
            const float k3 = last ? 0.f : gamma_lambda;
            const float d3 = fsub(last ? r : fmul(gamma, vn), x[u].w);
            const float d2 = fsub(fmul(gamma, x[u].w), x[u].z);
            const float d1 = fsub(fmul(gamma, x[u].z), x[u].y);
            const float d0 = fsub(fmul(gamma, x[u].y), x[u].x);
            float g = fadd(eG[u], fmul(eK[u], carry));  // g at the first token right of this lane's four
            carry = fadd(aG[u], fmul(aK[u], carry));
            f32x4 rt, at;
            g = fadd(d3, fmul(k3, g)), rt.w = fadd(g, x[u].w), at.w = fsub(rt.w, x[u].w);
            g = fadd(d2, fmul(gamma_lambda, g)), rt.z = fadd(g, x[u].z), at.z = fsub(rt.z, x[u].z);
            g = fadd(d1, fmul(gamma_lambda, g)), rt.y = fadd(g, x[u].y), at.y = fsub(rt.y, x[u].y);
            g = fadd(d0, fmul(gamma_lambda, g)), rt.x = fadd(g, x[u].x), at.x = fsub(rt.x, x[u].x);
            if (q >= 0 && row_live) {
                if constexpr (NT) {
                    __builtin_nontemporal_store(rt, r4 + q);
                    __builtin_nontemporal_store(at, a4 + q);
                } else {
                    r4[q] = rt;
                    a4[q] = at;
                }
            }
        }
    }
}

template <int S>
void launch_reg(bool nt, const float* values, const float* rewards, float* adv, float* ret, long long bsz, int seq, float gamma,
                float gl, hipStream_t st) {
    constexpr int WAVES = S < 4 ? 4 : S, ROWS = WAVES / S;
    const unsigned grid = (unsigned)((bsz + ROWS - 1) / ROWS);
    if (nt) hipLaunchKernelGGL((gae_seq_reg_kernel<S, true>), dim3(grid), dim3(64 * WAVES), 0, st, values, rewards, adv, ret, bsz, seq, gamma, gl);
    else hipLaunchKernelGGL((gae_seq_reg_kernel<S, false>), dim3(grid), dim3(64 * WAVES), 0, st, values, rewards, adv, ret, bsz, seq, gamma, gl);
}

}  // namespace
}  // namespace rlx

using namespace rlx;

constexpr int LB_SEG = 2048, LB_ST = 128;

extern "C" size_t rlx_gae_seq_workspace_bytes(int64_t bsz, int64_t seq) {
    if (seq <= LB_SEG || bsz <= 0) return 256;
    return (size_t)bsz * (size_t)((seq + LB_SEG - 1) / LB_SEG) * sizeof(Slot);
}

extern "C" int rlx_gae_seq(const float* values, const float* rewards, float* advantages, float* returns, int64_t bsz,
                           int64_t seq, float gamma, float gamma_lambda, void* workspace, size_t workspace_bytes,
                           rlx_stream_t stream) {
    RLX_REQUIRE(bsz >= 0 && seq >= 0 && bsz < (1ll << 31) && seq < (1ll << 31), "rlx_gae_seq: bad sizes");
    if (bsz == 0 || seq == 0) return RLX_OK;
    RLX_REQUIRE(values && rewards && advantages && returns, "rlx_gae_seq: NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto lds_for = [&](int seg) {
        const int cap = (int)(seq < seg ? seq : seg);
        return (size_t)(2 * (cap + 1 + (cap + 1) / 32) + 8) * sizeof(float);
    };
    // 16-byte aligned rows: the register kernel, one wave per 2048 tokens (rows up to 32 768 tokens).  Non-temporal accesses once
    // the three arrays together exceed the memory-side cache (profiles/r06_gae_seq_register_kernel.txt).
    // RLX_GAESEQ_REG=0 keeps the LDS kernels for every shape, 1 / 2 force ordinary / non-temporal accesses.
    const int reg_mode = [] { const char* e = getenv("RLX_GAESEQ_REG"); return e ? atoi(e) : -1; }();
    const int64_t ngroups = (seq / 4 + 63) / 64;
    const bool reg_ok = seq % 4 == 0 && ngroups <= 16 * WV_U &&
                        ((reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(advantages) |
                          reinterpret_cast<uintptr_t>(returns)) & 15u) == 0;
    if (reg_ok && reg_mode != 0) {
        const int need = (int)((ngroups + WV_U - 1) / WV_U);
        const bool nt = reg_mode == 1 ? false : reg_mode == 2 ? true : (size_t)bsz * (size_t)seq * 12 > ((size_t)256 << 20);
        if (need <= 1) launch_reg<1>(nt, values, rewards, advantages, returns, bsz, (int)seq, gamma, gamma_lambda, st);
        else if (need <= 2) launch_reg<2>(nt, values, rewards, advantages, returns, bsz, (int)seq, gamma, gamma_lambda, st);
        else if (need <= 4) launch_reg<4>(nt, values, rewards, advantages, returns, bsz, (int)seq, gamma, gamma_lambda, st);
        else if (need <= 8) launch_reg<8>(nt, values, rewards, advantages, returns, bsz, (int)seq, gamma, gamma_lambda, st);
        else launch_reg<16>(nt, values, rewards, advantages, returns, bsz, (int)seq, gamma, gamma_lambda, st);
    } else if (seq <= 2048) {
        // short sequences: one 128-lane workgroup each (longer chunks, fewer barriers per token), 17 KB of LDS -> nine per CU
        const int v = dev_variant_gaeseq();  // development: 20 = 256 lanes, 21 = 64 lanes, 22 = 1024-token segments
        if (v == 20)
            hipLaunchKernelGGL((gae_seq_kernel<256, 2048>), dim3((unsigned)bsz), dim3(256), lds_for(2048), st, values, rewards,
                               advantages, returns, (int)seq, gamma, gamma_lambda);
        else if (v == 21)
            hipLaunchKernelGGL((gae_seq_kernel<64, 2048>), dim3((unsigned)bsz), dim3(64), lds_for(2048), st, values, rewards,
                               advantages, returns, (int)seq, gamma, gamma_lambda);
        else if (v == 22)
            hipLaunchKernelGGL((gae_seq_kernel<128, 1024>), dim3((unsigned)bsz), dim3(128), lds_for(1024), st, values, rewards,
                               advantages, returns, (int)seq, gamma, gamma_lambda);
        else
        hipLaunchKernelGGL((gae_seq_kernel<128, 2048>), dim3((unsigned)bsz), dim3(128), lds_for(2048), st, values, rewards,
                           advantages, returns, (int)seq, gamma, gamma_lambda);
    } else if (dev_variant_gaeseq() != 2) {
        // long sequences: one workgroup walks the row in LDS segments from the end, 16 tokens per lane.  Measured at 4096 x 8192
        // (profiles/r02_gae_seq_segment_sweep.txt): 256 lanes x 4096-token segments (34 KB of LDS: four workgroups per CU, two
        // passes per row) 0.65 of the HBM peak; 512 x 8192 (66 KB, two per CU, one pass: the round-1 shape) 0.61;
        // 128 x 2048 0.58; 8 or 32 tokens per lane 0.43-0.51.  The look-back variant below -- one segment per workgroup, carries
        // through the memory side -- 0.50-0.59: the publish / poll round trips cost more than the idle phases they remove.
#define RLX_GAESEQ_LAUNCH(STV, SEGV)                                                                                         \
    {                                                                                                                        \
        static bool attr_set = false;                                                                                        \
        if (!attr_set) {                                                                                                     \
            RLX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gae_seq_kernel<STV, SEGV>),                      \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_for(SEGV)));              \
            attr_set = true;                                                                                                 \
        }                                                                                                                    \
        hipLaunchKernelGGL((gae_seq_kernel<STV, SEGV>), dim3((unsigned)bsz), dim3(STV), lds_for(SEGV), st, values, rewards,   \
                           advantages, returns, (int)seq, gamma, gamma_lambda);                                              \
    }
        if (dev_variant_gaeseq() == 1) RLX_GAESEQ_LAUNCH(512, 8192)  // the round-1 shape, kept tested
        else RLX_GAESEQ_LAUNCH(256, 4096)
#undef RLX_GAESEQ_LAUNCH
    } else {
        // RLX_GAESEQ_VARIANT=2 (kept tested): one LB_SEG-token segment per 128-lane workgroup, carries by decoupled look-back
        const int64_t nseg = (seq + LB_SEG - 1) / LB_SEG;
        RLX_REQUIRE(bsz * nseg < (1ll << 31), "rlx_gae_seq: too many segments for one launch");
        const size_t need = rlx_gae_seq_workspace_bytes(bsz, seq);
        RLX_REQUIRE(workspace != nullptr, "rlx_gae_seq: the look-back variant needs a workspace for rows longer than %d tokens", LB_SEG);
        if (workspace_bytes < need) {
            set_error("rlx_gae_seq: workspace %zu < %zu bytes", workspace_bytes, need);
            return RLX_ENOSPC;
        }
        RLX_HIP_CHECK(hipMemsetAsync(workspace, 0xFF, need, st));  // every slot invalid (NaN patterns)
        hipLaunchKernelGGL((gae_seq_lb_kernel<LB_ST, LB_SEG>), dim3((unsigned)(bsz * nseg)), dim3(LB_ST),
                           (size_t)(2 * (LB_SEG + 1 + (LB_SEG + 1) / 32) + 8) * sizeof(float), st, values, rewards, advantages,
                           returns, (int)seq, (int)nseg, gamma, gamma_lambda, static_cast<Slot*>(workspace));
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
