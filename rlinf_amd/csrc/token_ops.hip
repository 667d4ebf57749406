// token_ops.hip -- the reasoning (LLM) learner's per-token path over vocabulary logits, gfx950.
//
//   token_logprob_fwd/bwd  compute_logprobs_from_logits + compute_entropy_from_logits (rlinf/utils/utils.py:454-512)
//                          with the temperature division of FSDPActor.forward_batch (fsdp_actor_worker.py:476-498)
//   token_loss_fwd/bwd     the micro-batch loss of FSDPActor.training_step (fsdp_actor_worker.py:694-781)
//   grpo_seq_adv           reasoning GRPO advantages in the [bsz, seq] layout (algorithms/utils.py:177-277,
//                          advantages.py:89-121)
//
// The two logits kernels are pure HBM streams (n_tokens * vocab elements read once forward; read + written once
// backward): one workgroup per token row, 16-byte loads, four in flight per lane, an online softmax in base 2 whose
// running reference point is an INTEGER power (r = ceil(max * log2 e)), so that every rescale is an exact power of
// two and d = fma(x, log2 e, -r) carries one rounding and no systematic error.  Per row the workgroup keeps
//     s = sum 2^d,   t = sum 2^d * d          =>  lse = (r + log2 s) ln 2,   H = (log2 s - t / s) ln 2.
// The loss kernels touch 4-6 floats per token: one workgroup per sequence, fixed-order double sums, a one-block
// finalisation -- deterministic, no atomics.

#include "ppo_loss_math.h"

namespace rlx {
namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float R_INIT = -1.0e30f;  // "no element yet": finite so that (r_old - r_new) * 0 stays 0
constexpr float D_FLOOR = -1.0e30f; // x = -inf: d is clamped so that 2^d * d = 0 * finite

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;  // threads per row
#ifndef RLX_TOK_UNROLL
#define RLX_TOK_UNROLL 4
#endif
#ifndef RLX_TOK_BWD_NT_STORE
#define RLX_TOK_BWD_NT_STORE 1
#endif
#ifndef RLX_TOK_BWD_NT_LOAD
#define RLX_TOK_BWD_NT_LOAD 1
#endif
constexpr int UNROLL = RLX_TOK_UNROLL;

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ void unpack(const u32x4& q, float (&x)[4]) {
        x[0] = __uint_as_float(q.x), x[1] = __uint_as_float(q.y), x[2] = __uint_as_float(q.z), x[3] = __uint_as_float(q.w);
    }
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ float round(float v) { return v; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
    static __device__ __forceinline__ u32x4 pack(const float (&x)[4]) {
        u32x4 o = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
        return o;
    }
};
template <>
struct Elem<__bf16> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ void unpack(const u32x4& q, float (&x)[8]) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[2 * i] = __uint_as_float(w[i] << 16);
            x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ float load(const __bf16* p) { return (float)*p; }
    static __device__ __forceinline__ float round(float v) { return (float)(__bf16)v; }
    static __device__ __forceinline__ void store(__bf16* p, float v) { *p = (__bf16)v; }
    static __device__ __forceinline__ u32x4 pack(const float (&x)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            bf2 v;
            v[0] = (__bf16)x[2 * i];
            v[1] = (__bf16)x[2 * i + 1];
            w[i] = *reinterpret_cast<uint32_t*>(&v);
        }
        u32x4 o = {w[0], w[1], w[2], w[3]};
        return o;
    }
};

// x / T the way torch's in-place div_ leaves it in the tensor's dtype.
//   f32 : one Newton step on x * (1/T) gives the correctly rounded quotient (x / T bit for bit);
//   bf16: x * (1/T) rounded to bf16 -- equal to bf16(x / T) unless the f32 product falls within half an f32 ulp of a
//         bf16 rounding boundary (~2^-15 of the elements, one bf16 ulp on one logit when it happens).
template <typename T, bool SCALE>
__device__ __forceinline__ float prep(float x, float temp, float rtemp);
template <>
__device__ __forceinline__ float prep<float, false>(float x, float, float) { return x; }
template <>
__device__ __forceinline__ float prep<__bf16, false>(float x, float, float) { return x; }
template <>
__device__ __forceinline__ float prep<float, true>(float x, float temp, float rtemp) {
    const float q = fmul(x, rtemp);
    const float q2 = fmaf(fmaf(-q, temp, x), rtemp, q);
    return fabsf(q) < __builtin_inff() ? q2 : q;
}
template <>
__device__ __forceinline__ float prep<__bf16, true>(float x, float, float rtemp) {
    return (float)(__bf16)fmul(x, rtemp);
}
// N elements at once; bf16 pairs share one v_cvt_pk_bf16_f32
template <typename T, bool SCALE, int N>
__device__ __forceinline__ void prep_vec(float (&x)[N], float temp, float rtemp) {
    if (!SCALE) return;
    if constexpr (sizeof(T) == 2 && (N % 2) == 0) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < N; i += 2) {
            bf2 v;
            v[0] = (__bf16)fmul(x[i], rtemp);
            v[1] = (__bf16)fmul(x[i + 1], rtemp);
            const uint32_t w = *reinterpret_cast<uint32_t*>(&v);
            x[i] = __uint_as_float(w << 16);
            x[i + 1] = __uint_as_float(w & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = prep<T, SCALE>(x[i], temp, rtemp);
    }
}

struct Stat {
    float r, s, t;
};
__device__ __forceinline__ Stat stat_merge(const Stat& a, const Stat& b) {
    const float R = fmaxf(a.r, b.r);
    const float da = a.r - R, db = b.r - R;  // integers <= 0: the factors are exact powers of two
    const float fa = __builtin_amdgcn_exp2f(da), fb = __builtin_amdgcn_exp2f(db);
    Stat o;
    o.r = R;
    o.s = fmaf(a.s, fa, b.s * fb);
    o.t = fmaf(fa, fmaf(da, a.s, a.t), fb * fmaf(db, b.s, b.t));
    return o;
}
__device__ __forceinline__ Stat stat_shfl_xor(const Stat& a, int off) {
    Stat o;
    o.r = __shfl_xor(a.r, off, RLX_WAVE), o.s = __shfl_xor(a.s, off, RLX_WAVE), o.t = __shfl_xor(a.t, off, RLX_WAVE);
    return o;
}

// fold N prepared elements into the running statistics
template <int N, bool ENT>
__device__ __forceinline__ void stat_add(Stat& st, const float (&x)[N]) {
    float cm = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) cm = fmaxf(cm, x[i]);
    const float rn = fmaxf(st.r, ceilf(fmul(cm, LOG2E)));
    const float dr = st.r - rn;
    const float f = __builtin_amdgcn_exp2f(dr);
    if (ENT) st.t = fmul(f, fmaf(dr, st.s, st.t));
    st.s = fmul(st.s, f);
    st.r = rn;
    float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float d = fmaf(x[i], LOG2E, -rn);
        if (ENT) d = fmaxf(d, D_FLOOR);
        const float e = __builtin_amdgcn_exp2f(d);
        if (i & 1) {
            s1 += e;
            if (ENT) t1 = fmaf(e, d, t1);
        } else {
            s0 += e;
            if (ENT) t0 = fmaf(e, d, t0);
        }
    }
    st.s += s0 + s1;
    if (ENT) st.t += t0 + t1;
}

struct RowGeom {
    long long n_tokens;
    int vocab;
    long long rows_per_seq, seq_stride, row_stride;
    float temp, rtemp;
    int round_outputs;
};
__device__ __forceinline__ long long row_offset(const RowGeom& g, long long i, long long seq_stride, long long row_stride) {
    return (i / g.rows_per_seq) * seq_stride + (i % g.rows_per_seq) * row_stride;
}

// PACKED (rlx_token_logprob_fwd_packed): the rows are the tokens of a PACKED stream (sequences back to back, no padding) and the
// results leave in the unpacked [bsz, response_len] layout -- row t's log-prob goes to logprob[lp_dst[t]], its entropy to
// entropy[ent_dst[t]] (-1: nowhere); a row with no destination at all is not read.  lse stays per packed row (the backward's).
template <typename T, bool ENT, bool SCALE, bool PACKED = false>
__global__ __launch_bounds__(NT) void token_logprob_fwd_kernel(const T* __restrict__ logits,
                                                               const int64_t* __restrict__ labels, RowGeom g,
                                                               float* __restrict__ logprob, float* __restrict__ entropy,
                                                               float* __restrict__ lse_out, const int32_t* __restrict__ lp_dst = nullptr,
                                                               const int32_t* __restrict__ ent_dst = nullptr) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ Stat s_part[NT / RLX_WAVE];
    const int tid = threadIdx.x;
    for (long long row = blockIdx.x; row < g.n_tokens; row += gridDim.x) {
        long long dl = row, de = row;
        if constexpr (PACKED) {
            dl = lp_dst[row];
            de = ENT ? ent_dst[row] : -1;
            if (dl < 0 && de < 0) {  // (uniform over the workgroup: nobody reaches this iteration's barrier)
                if (tid == 0) lse_out[row] = 0.f;
                continue;
            }
        }
        const T* __restrict__ base = logits + row_offset(g, row, g.seq_stride, g.row_stride);
        const int V = g.vocab;
        const int head = min(V, (int)(((16u - (unsigned)((uintptr_t)base & 15u)) & 15u) / sizeof(T)));
        const int nvec = (V - head) / VEC;
        const int tail0 = head + nvec * VEC;
        const u32x4* __restrict__ vb = reinterpret_cast<const u32x4*>(base + head);
        Stat st{R_INIT, 0.f, 0.f};

        // full groups: UNROLL 16-byte loads in flight per lane, no guards
        const int group = NT * UNROLL;
        const int nfull = nvec / group;
        for (int gi = 0; gi < nfull; ++gi) {
            u32x4 q[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) q[u] = __builtin_nontemporal_load(vb + (size_t)gi * group + u * NT + tid);
            float x[UNROLL * VEC];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float xv[VEC];
                Elem<T>::unpack(q[u], xv);
                prep_vec<T, SCALE, VEC>(xv, g.temp, g.rtemp);
#pragma unroll
                for (int i = 0; i < VEC; ++i) x[u * VEC + i] = xv[i];
            }
            stat_add<UNROLL * VEC, ENT>(st, x);
        }
        // remaining vectors (< UNROLL per lane)
        for (int v = nfull * group + tid; v < nvec; v += NT) {
            const u32x4 q = __builtin_nontemporal_load(vb + v);
            float xv[VEC];
            Elem<T>::unpack(q, xv);
            prep_vec<T, SCALE, VEC>(xv, g.temp, g.rtemp);
            stat_add<VEC, ENT>(st, xv);
        }
        // unaligned head and the tail, element by element
        const int nscalar = head + (V - tail0);
        for (int i = tid; i < nscalar; i += NT) {
            const int idx = i < head ? i : tail0 + (i - head);
            float xs[1] = {prep<T, SCALE>(Elem<T>::load(base + idx), g.temp, g.rtemp)};
            stat_add<1, ENT>(st, xs);
        }

#pragma unroll
        for (int off = 32; off > 0; off >>= 1) st = stat_merge(st, stat_shfl_xor(st, off));
        if ((tid & 63) == 0) s_part[tid >> 6] = st;
        __syncthreads();
        if (tid == 0) {
            Stat a = s_part[0];
#pragma unroll
            for (int w = 1; w < NT / RLX_WAVE; ++w) a = stat_merge(a, s_part[w]);
            const float log2s = log2f(a.s);
            const long long lab = labels[row];
            float lp;
            if (lab >= 0 && lab < V) {
                const float xy = prep<T, SCALE>(Elem<T>::load(base + lab), g.temp, g.rtemp);
                lp = fmul(fmaf(xy, LOG2E, -a.r) - log2s, LN2);
            } else {
                lp = lab == -100 ? -0.f : __builtin_nanf("");
            }
            if (g.round_outputs) lp = Elem<T>::round(lp);
            if (!PACKED || dl >= 0) logprob[dl] = lp;
            lse_out[row] = fmul(a.r + log2s, LN2);
            if (ENT && (!PACKED || de >= 0)) entropy[de] = fmul(log2s - a.t / a.s, LN2);
        }
        __syncthreads();
    }
}

// d_logits = (glp * (onehot - p) - gH * p * (log p + H)) / T
// PACKED: d_logprob / d_entropy / entropy are the UNPACKED [bsz, response_len] tensors and lp_dst / ent_dst the forward's maps --
// row t takes the upstream gradient of the element it fed (0 where it fed none: the row is then written as zeros unread).
template <typename T, bool ENT, bool SCALE, bool PACKED = false>
__global__ __launch_bounds__(NT) void token_logprob_bwd_kernel(const T* logits, const int64_t* __restrict__ labels,
                                                               RowGeom g, const float* __restrict__ lse,
                                                               const float* __restrict__ entropy,
                                                               const float* __restrict__ d_logprob,
                                                               const float* __restrict__ d_entropy, T* d_logits,
                                                               long long d_seq_stride, long long d_row_stride,
                                                               const int32_t* __restrict__ lp_dst = nullptr,
                                                               const int32_t* __restrict__ ent_dst = nullptr) {
    constexpr int VEC = Elem<T>::VEC;
    const int tid = threadIdx.x;
    __shared__ float s_xy;
    for (long long row = blockIdx.x; row < g.n_tokens; row += gridDim.x) {
        const T* base = logits + row_offset(g, row, g.seq_stride, g.row_stride);
        T* out = d_logits + row_offset(g, row, d_seq_stride, d_row_stride);
        const int V = g.vocab;
        int dl = (int)row, de = (int)row;
        if constexpr (PACKED) {
            dl = lp_dst[row];
            de = ENT ? ent_dst[row] : -1;
        }
        const float glp = (!PACKED || dl >= 0) ? d_logprob[PACKED ? dl : row] : 0.f;
        const float gh = (ENT && (!PACKED || de >= 0)) ? d_entropy[PACKED ? de : row] : 0.f;
        const bool same_phase = (((uintptr_t)base ^ (uintptr_t)out) & 15u) == 0;
        const int head = same_phase ? min(V, (int)(((16u - (unsigned)((uintptr_t)base & 15u)) & 15u) / sizeof(T))) : V;
        const int nvec = (V - head) / VEC;
        const int tail0 = head + nvec * VEC;
        const int nscalar = head + (V - tail0);
        u32x4* ov = reinterpret_cast<u32x4*>(out + head);
        if (glp == 0.f && gh == 0.f) {  // masked-out token: zeros, the logits row is never read
            for (int v = tid; v < nvec; v += NT) ov[v] = u32x4{0, 0, 0, 0};
            for (int i = tid; i < nscalar; i += NT) Elem<T>::store(out + (i < head ? i : tail0 + (i - head)), 0.f);
            continue;
        }
        const long long lab = labels[row];
        const bool lab_ok = lab >= 0 && lab < V;
        if (tid == 0 && lab_ok) s_xy = prep<T, SCALE>(Elem<T>::load(base + lab), g.temp, g.rtemp);
        __syncthreads();  // the label's logit is read before any lane may overwrite it (in-place use)
        const float L = lse[row];
        const float nL2 = -fmul(L, LOG2E);
        const float H = (ENT && (!PACKED || de >= 0)) ? entropy[PACKED ? de : row] : 0.f;
        const float HmL = H - L;  // log p + H = x + (H - lse)
        const u32x4* vb = reinterpret_cast<const u32x4*>(base + head);

        auto grad = [&](float x) -> float {
            const float p = __builtin_amdgcn_exp2f(fmaf(x, LOG2E, nL2));
            float c = glp;
            if (ENT) c = (p > 0.f) ? fmaf(gh, x + HmL, glp) : glp;  // x = -inf: the reference drops the p*log p term
            return fmul(-fmul(p, c), g.rtemp);
        };

        const int group = NT * UNROLL;
        const int nfull = nvec / group;
        for (int gi = 0; gi < nfull; ++gi) {
            u32x4 q[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                q[u] = RLX_TOK_BWD_NT_LOAD ? __builtin_nontemporal_load(vb + (size_t)gi * group + u * NT + tid)
                                           : vb[(size_t)gi * group + u * NT + tid];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                float xv[VEC];
                Elem<T>::unpack(q[u], xv);
                prep_vec<T, SCALE, VEC>(xv, g.temp, g.rtemp);
#pragma unroll
                for (int i = 0; i < VEC; ++i) xv[i] = grad(xv[i]);
                if (RLX_TOK_BWD_NT_STORE) __builtin_nontemporal_store(Elem<T>::pack(xv), ov + (size_t)gi * group + u * NT + tid);
                else ov[(size_t)gi * group + u * NT + tid] = Elem<T>::pack(xv);
            }
        }
        for (int v = nfull * group + tid; v < nvec; v += NT) {
            const u32x4 q = __builtin_nontemporal_load(vb + v);
            float xv[VEC];
            Elem<T>::unpack(q, xv);
            prep_vec<T, SCALE, VEC>(xv, g.temp, g.rtemp);
#pragma unroll
            for (int i = 0; i < VEC; ++i) xv[i] = grad(xv[i]);
            __builtin_nontemporal_store(Elem<T>::pack(xv), ov + v);
        }
        for (int i = tid; i < nscalar; i += NT) {
            const int idx = i < head ? i : tail0 + (i - head);
            Elem<T>::store(out + idx, grad(prep<T, SCALE>(Elem<T>::load(base + idx), g.temp, g.rtemp)));
        }
        __syncthreads();  // every element of the row is written; now the label's entry gets its one-hot term
        if (tid == 0 && lab_ok) {
            const float x = s_xy;
            const float p = __builtin_amdgcn_exp2f(fmaf(x, LOG2E, nL2));
            float c = glp;
            if (ENT) c = (p > 0.f) ? fmaf(gh, x + HmL, glp) : glp;
            Elem<T>::store(out + lab, fmul(glp - fmul(p, c), g.rtemp));
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// token loss
// ---------------------------------------------------------------------------------------------------------
constexpr int RS = 12;  // per-sequence sums
enum { R_NM = 0, R_LOSS, R_ABS, R_RATIO, R_RABS, R_CLIPPED, R_DUAL, R_KL, R_CLIPFRAC, R_ENT, R_KLD, R_PAD };

struct TokLossArgs {
    const float *logprobs, *old_logprobs, *advantages, *ref_logprobs, *entropy;
    const uint8_t* loss_mask;
    long long bsz, seq;
    rlx_token_loss_params p;
    float *g_logp, *g_entropy, *row_weight, *out;
    double* row_sums;  // [bsz][RS]
};

__device__ __forceinline__ float kl_elem(int kind, float first, float second, float& dsecond) {
    // kl_penalty(logprob=first, ref_logprob=second): the learner passes (ref_logprobs, logprobs), so the
    // derivative wanted is the one w.r.t. `second`.
    const float diff = fsub(first, second);
    switch (kind) {
        case RLX_KL_K1: dsecond = -1.f; return diff;
        case RLX_KL_ABS: dsecond = diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f); return fabsf(diff);
        case RLX_KL_K2: dsecond = -diff; return fmul(0.5f, fmul(diff, diff));
        default: {  // k3
            const float raw = fsub(second, first);
            const float kl = fminf(fmaxf(raw, -20.f), 20.f);
            const float r = expf(kl);
            const float kld = fsub(fsub(r, kl), 1.f);
            const bool in1 = raw >= -20.f && raw <= 20.f, in2 = kld >= -10.f && kld <= 10.f;
            dsecond = (in1 && in2) ? fsub(r, 1.f) : 0.f;
            return fminf(fmaxf(kld, -10.f), 10.f);
        }
    }
}

__global__ __launch_bounds__(256) void token_loss_rows_kernel(TokLossArgs a) {
    __shared__ double scratch[RS * 4];
    __shared__ int s_row0_on;
    const long long row = blockIdx.x;
    const int tid = threadIdx.x;
    const long long S = a.seq;
    bool policy_on = true;
    if (a.p.fast_path_zero_loss_mask && a.loss_mask) {  // losses.py:206 looks at the first sequence only
        if (tid == 0) s_row0_on = 0;
        __syncthreads();
        int any = 0;
        for (long long j = tid; j < S; j += blockDim.x) any |= a.loss_mask[j];
        if (any) s_row0_on = 1;
        __syncthreads();
        policy_on = s_row0_on != 0;
    }
    const bool has_kl = a.p.kl_type != RLX_KL_NONE && a.ref_logprobs != nullptr;
    const bool has_ent = a.p.use_entropy && a.entropy != nullptr;
    double acc[loss::NS];
#pragma unroll
    for (int k = 0; k < loss::NS; ++k) acc[k] = 0.0;
    double ent_sum = 0.0, kld_sum = 0.0;
    for (long long j = tid; j < S; j += blockDim.x) {
        const long long i = row * S + j;
        const bool on = a.loss_mask ? a.loss_mask[i] != 0 : true;
        const float mf = on ? 1.f : 0.f;
        const float lp = a.logprobs[i];
        float g = loss::actor_elem(a.p.ppo, lp, a.old_logprobs[i], a.advantages[i], on, 1.f, false, acc);
        if (!policy_on) g = 0.f;
        acc[loss::S_NM] += on ? 1.0 : 0.0;
        if (has_kl) {
            float dk;
            const float kld = kl_elem(a.p.kl_type, a.ref_logprobs[i], lp, dk);
            kld_sum += (double)fmul(kld, mf);
            g = fmaf(a.p.kl_beta, dk * mf, g);
        }
        a.g_logp[i] = g;
        if (has_ent) {
            ent_sum += (double)fmul(a.entropy[i], mf);
            if (a.g_entropy) a.g_entropy[i] = -a.p.entropy_bonus * mf;
        }
    }
    double v[RS] = {acc[loss::S_NM],      acc[loss::S_LOSS], acc[loss::S_ABS], acc[loss::S_RATIO],
                    acc[loss::S_RABS],    acc[loss::S_CLIPPED], acc[loss::S_DUAL], acc[loss::S_KL],
                    acc[loss::S_CLIPFRAC], ent_sum,          kld_sum,          policy_on ? 1.0 : 0.0};
    block_sum<RS>(v, scratch);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < RS; ++k) a.row_sums[row * RS + k] = v[k];
        a.row_sums[row * RS + R_PAD] = policy_on ? 1.0 : 0.0;
    }
}

// aggregate one per-sequence quantity the way loss_agg_func does (rlinf/utils/utils.py:323-356)
__device__ __forceinline__ double agg_term(int agg, double row_sum, double row_cnt) {
    return agg == RLX_AGG_SEQ_MEAN_TOKEN_MEAN ? row_sum / row_cnt : row_sum;
}

__global__ __launch_bounds__(256) void token_loss_finalize_kernel(TokLossArgs a) {
    __shared__ double scratch[16 * 4];
    const int tid = threadIdx.x;
    const int agg = a.p.loss_agg;
    // slots: 0 count, 1 loss, 2 abs, 3 ent, 4 kld (aggregated per the rule), 5.. global metric sums
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0.0;
    for (long long r = tid; r < a.bsz; r += blockDim.x) {
        const double* rs = a.row_sums + r * RS;
        const double cnt = rs[R_NM];
        v[0] += cnt;
        v[1] += agg_term(agg, rs[R_LOSS], cnt);
        v[2] += agg_term(agg, rs[R_ABS], cnt);
        v[3] += agg_term(agg, rs[R_ENT], cnt);
        v[4] += agg_term(agg, rs[R_KLD], cnt);
        v[5] += rs[R_RATIO], v[6] += rs[R_RABS], v[7] += rs[R_CLIPPED], v[8] += rs[R_DUAL];
        v[9] += rs[R_KL], v[10] += rs[R_CLIPFRAC];
    }
    block_sum<16>(v, scratch);
    __shared__ double s_den;
    if (tid == 0) {
        const double nm = v[0];
        const bool policy_on = a.row_sums[R_PAD] != 0.0;
        // token-mean == masked_mean: sum / count, or the plain (zero) sum when nothing is on
        const double den = agg == RLX_AGG_TOKEN_MEAN ? (nm > 0 ? nm : 1.0) : (double)a.bsz;
        s_den = den;
        const double cnt1 = nm > 0 ? nm : 1.0;  // loss_mask.count_nonzero() or 1; masked_mean's metric denominator
        const float pol = (policy_on && !a.p.ppo.critic_warmup) ? (float)(v[1] / den) : 0.f;
        const float ent = (a.p.use_entropy && a.entropy) ? (float)(v[3] / den) : 0.f;
        const bool has_kl = a.p.kl_type != RLX_KL_NONE && a.ref_logprobs != nullptr;
        const float kl = has_kl ? (float)(v[4] / den) : 0.f;
        float total = pol;
        if (a.p.use_entropy && a.entropy && a.p.entropy_bonus != 0.f) total = fsub(total, fmul(a.p.entropy_bonus, ent));
        if (has_kl) total = fadd(total, fmul(kl, a.p.kl_beta));
        float* o = a.out;
        o[RLX_TOK_LOSS] = total;
        o[RLX_TOK_POLICY_LOSS] = pol;
        o[RLX_TOK_POLICY_LOSS_ABS] = policy_on ? (float)(v[2] / den) : 0.f;
        o[RLX_TOK_RATIO] = policy_on ? (float)(v[5] / cnt1) : 0.f;
        o[RLX_TOK_RATIO_ABS] = policy_on ? (float)(v[6] / cnt1) : 0.f;
        o[RLX_TOK_CLIPPED_RATIO] = policy_on ? (float)(v[7] / cnt1) : 0.f;
        o[RLX_TOK_DUAL_CLIPPED_RATIO] = policy_on ? (float)(v[8] / cnt1) : 0.f;
        o[RLX_TOK_APPROX_KL] = policy_on ? (float)(-v[9] / cnt1) : 0.f;
        o[RLX_TOK_CLIP_FRACTION] = policy_on ? (float)(v[10] / cnt1) : 0.f;
        o[RLX_TOK_ENTROPY_LOSS] = ent;
        o[RLX_TOK_KL_LOSS] = kl;
        o[RLX_TOK_TOKEN_NUM] = (float)nm;
        o[RLX_TOK_POLICY_ON] = policy_on ? 1.f : 0.f;
        o[13] = o[14] = o[15] = 0.f;
    }
    __syncthreads();
    const double den = s_den;
    for (long long r = tid; r < a.bsz; r += blockDim.x) {
        const double cnt = a.row_sums[r * RS + R_NM];
        a.row_weight[r] = agg == RLX_AGG_SEQ_MEAN_TOKEN_MEAN ? (float)(1.0 / (den * cnt)) : (float)(1.0 / den);
    }
}

__global__ __launch_bounds__(256) void token_loss_bwd_kernel(const float* __restrict__ g_logp,
                                                             const float* __restrict__ g_entropy,
                                                             const float* __restrict__ row_weight,
                                                             const float* __restrict__ grad_out,
                                                             float* __restrict__ d_logprobs, float* __restrict__ d_entropy,
                                                             long long S) {
    const long long row = blockIdx.x;
    const float w = fmul(row_weight[row], grad_out[0]);
    for (long long j = (long long)blockIdx.y * blockDim.x + threadIdx.x; j < S; j += (long long)gridDim.y * blockDim.x) {
        const long long i = row * S + j;
        // a masked token has g == 0: keep it 0 even when the row weight is inf (all-masked sequence under
        // seq-mean-token-mean is NaN in the reference's forward already; the gradient follows 0 * inf = NaN there
        // too, which fmul reproduces)
        d_logprobs[i] = fmul(g_logp[i], w);
        if (d_entropy) d_entropy[i] = g_entropy ? fmul(g_entropy[i], w) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// GRPO advantages in the sequence-major layout
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grpo_seq_adv_kernel(const float* __restrict__ rewards,
                                                           const uint8_t* __restrict__ m, float* __restrict__ adv,
                                                           long long S, int G, float eps) {
    const long long b = blockIdx.x;
    const long long g0 = (b / G) * G;
    // same evaluation order as grpo_broadcast (grpo_adv.hip): sequential f32 sums over the group
    float sum = 0.f;
    for (int j = 0; j < G; ++j) sum += rewards[g0 + j];
    const float mean = sum / (float)G;
    float ss = 0.f;
    for (int j = 0; j < G; ++j) {
        const float dlt = rewards[g0 + j] - mean;
        ss += dlt * dlt;
    }
    const float sd = sqrtf(ss / (float)(G - 1));
    const float a = fsub(rewards[b], mean) / fadd(sd, eps);
    for (long long j = (long long)blockIdx.y * blockDim.x + threadIdx.x; j < S; j += (long long)gridDim.y * blockDim.x)
        adv[b * S + j] = fmul(a, m[b * S + j] ? 1.f : 0.f);
}

// ---------------------------------------------------------------------------------------------------------
// categorical action sampling (K2): one wave per row of K <= 64*EPL action-bin logits
// ---------------------------------------------------------------------------------------------------------
struct CatArgs {
    const void* logits;
    const void* noise;  // Exp(1) draws, same dtype and [n, K] dense layout; NULL = argmax (do_sample False)
    RowGeom g;
    int top_k;
    int lanes;  // 16 / 8: f32 lanes of the reference host's vector kernels (fixes the order of the row sum's additions)
    const float* bin_centers;
    int n_centers;
    long long* tokens;
    float* logprob;
    float* actions;
};

__device__ __forceinline__ uint32_t order_key(float f) {  // monotone float -> uint (NaN-free inputs)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RLX_WAVE);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, RLX_WAVE));
    return v;
}

// ---- the reference's CPU softmax, operation for operation --------------------------------------------------------------------
// north_star: "bit-exact for action indices".  The reference samples with torch.multinomial(softmax(x), 1), i.e. argmax(softmax(x) / q)
// for its internal Exp(1) draw q, and softmax on the host is ATen's vectorised last-dim kernel (aten/src/ATen/native/cpu/
// SoftMaxKernel.cpp, _vec_softmax_lastdim): e = Sleef_expf{16,8}_u10(x - max) lane-wise, the row sum accumulated per SIMD lane over
// consecutive W-element chunks and folded by a butterfly of W lanes, p = e * (1 / sum).  Every one of those steps is IEEE
// arithmetic with a fixed order, so it can be replayed exactly: W = 16 (AVX-512 builds of torch, what the golden files were
// written on) or 8 (AVX2) is `softmax_lanes`.  Verified element for element against torch.softmax (f32 and bf16, K = 7 .. 1024,
// with and without chunk tails) by the numpy restatement in tests/test_token_host.py::test_cpu_softmax_restatement.
//   f32 rows   : all K elements through the vector exp; lane c sums e[c], e[W + c], ... in order (a partial last chunk adds to
//                its first K % W lanes); K < W: one sequential sum e0 + e1 + ...
//   bf16 rows  : the K - K % W leading elements as above (accumulators start at 0), the tail through the scalar expf
//                (correctly rounded) and added one by one BEHIND the butterfly; p rounded to bf16.
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }
__device__ __forceinline__ float sleef_expf_u10(float d) {
    const float dc = fmaxf(d, -120.f);  // (-inf / hugely negative: the result is forced to 0 below; keeps q a small integer)
    const float q = rintf(fmul(dc, 1.442695040888963407359924681001892137426645954152985934135449406931f));
    float s = fmaf(q, -0.693145751953125f, dc);
    s = fmaf(q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    u = fadd(1.0f, fmaf(fmul(s, s), u, s));
    const int qi = (int)q;
    u = fmul(fmul(u, pow2i(qi >> 1)), pow2i(qi - (qi >> 1)));  // vldexp2
    if (d < -104.f) u = 0.f;
    if (d > 100.f) u = __builtin_inff();
    return u;
}

template <typename T, int EPL>
__global__ __launch_bounds__(256) void categorical_sample_kernel(CatArgs a) {
    constexpr bool REDUCED = sizeof(T) == 2;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.g.n_tokens) return;
    const int K = a.g.vocab, W = a.lanes;
    const T* base = static_cast<const T*>(a.logits) + row_offset(a.g, row, a.g.seq_stride, a.g.row_stride);
    const float NEG_INF = -__builtin_inff();
    const bool sample = a.noise != nullptr, scale = sample && a.g.temp != 1.0f;
    float x[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int idx = lane + 64 * j;
        float raw = Elem<T>::load(base + (idx < K ? idx : 0));
        if (scale) raw = Elem<T>::round(raw / a.g.temp);  // tensor / python scalar: a true division, rounded to the tensor's dtype
        x[j] = idx < K ? raw : NEG_INF;
    }
    if (sample && a.top_k > 0 && a.top_k < K) {
        // k-th largest score = the largest t with count(key >= t) >= k, built bit by bit
        uint32_t key[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) key[j] = (lane + 64 * j) < K ? order_key(x[j]) : 0u;
        uint32_t t = 0;
        for (int b = 31; b >= 0; --b) {
            const uint32_t cand = t | (1u << b);
            int c = 0;
#pragma unroll
            for (int j = 0; j < EPL; ++j) c += key[j] >= cand ? 1 : 0;
            if (wave_sum_i(c) >= a.top_k) t = cand;
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (key[j] < t) x[j] = NEG_INF;  // TopKLogitsWarper: scores < kth -> -inf (ties with the k-th are kept)
    }
    float m = x[0];
#pragma unroll
    for (int j = 1; j < EPL; ++j) m = fmaxf(m, x[j]);
    m = wave_max_f(m);
    // indices below `limit` go through the W lane accumulators, the rest is added one by one
    const int limit = REDUCED ? K - K % W : (K < W ? 0 : K);
    float e[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int idx = lane + 64 * j;
        const float d = fsub(x[j], m);
        e[j] = idx >= K ? 0.f : (REDUCED && idx >= limit) ? (float)exp((double)d) : sleef_expf_u10(d);
    }
    const int c = lane & (W - 1);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j)
        for (int mm = 0; mm < 64 / W; ++mm) {  // chunk 64 j / W + mm: element W mm + c of register j (every lane group of W agrees)
            const float v = __shfl(e[j], W * mm + c, RLX_WAVE);
            if (64 * j + W * mm + c < limit) acc = fadd(acc, v);
        }
    for (int sh = W >> 1; sh > 0; sh >>= 1) acc = fadd(acc, __shfl_xor(acc, sh, RLX_WAVE));
    float s = acc;
    for (int i = limit; i < K; ++i) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if ((i >> 6) == j) v = e[j];
        s = fadd(s, __shfl(v, i & 63, RLX_WAVE));
    }
    const float rs = 1.0f / s;
    // the race: argmax_v softmax(x)_v / q_v (torch.multinomial, num_samples = 1), or argmax_v x_v without sampling;
    // the lowest index wins ties, as torch.argmax does
    float best = NEG_INF;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        const int idx = lane + 64 * j;
        if (idx >= K) continue;
        float score;
        if (sample) {
            const float p = Elem<T>::round(fmul(e[j], rs));  // softmax output in the tensor's dtype
            const float q = Elem<T>::load(static_cast<const T*>(a.noise) + row * K + idx);
            score = Elem<T>::round(p / q);
        } else {
            score = x[j];
        }
        if (score > best || (score == best && idx < bidx)) best = score, bidx = idx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, RLX_WAVE);
        const int oi = __shfl_xor(bidx, off, RLX_WAVE);
        if (ob > best || (ob == best && oi < bidx)) best = ob, bidx = oi;
    }
    // the owner of the winning element reports
    if ((bidx & 63) == lane) {
        float xt = NEG_INF;
#pragma unroll
        for (int j = 0; j < EPL; ++j)
            if (bidx == lane + 64 * j) xt = x[j];
        a.tokens[row] = bidx;
        if (a.logprob) {
            float lp = (xt - m) - logf(s);
            if (a.g.round_outputs) lp = Elem<T>::round(lp);
            a.logprob[row] = lp;
        }
        if (a.actions) {
            int d = K - bidx - 1;
            d = d < 0 ? 0 : (d > a.n_centers - 1 ? a.n_centers - 1 : d);
            a.actions[row] = a.bin_centers[d];
        }
    }
}

template <typename T, int EPL>
int launch_cat(const CatArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.g.n_tokens + 3) / 4)), block(256);
    hipLaunchKernelGGL((categorical_sample_kernel<T, EPL>), grid, block, 0, s, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
template <typename T>
int dispatch_cat(const CatArgs& a, hipStream_t s) {
    const int K = a.g.vocab;
    if (K <= 256) return launch_cat<T, 4>(a, s);
    if (K <= 512) return launch_cat<T, 8>(a, s);
    return launch_cat<T, 16>(a, s);
}

int check_rows(const rlx_token_rows* r, const char* who) {
    RLX_REQUIRE(r != nullptr, "%s: NULL rows", who);
    RLX_REQUIRE(r->n_tokens >= 0 && r->vocab >= 1, "%s: bad sizes", who);
    RLX_REQUIRE(r->dtype == RLX_DTYPE_F32 || r->dtype == RLX_DTYPE_BF16, "%s: unsupported dtype %d", who, r->dtype);
    RLX_REQUIRE(r->rows_per_seq >= 1 && r->row_stride >= r->vocab, "%s: bad row geometry", who);
    RLX_REQUIRE(r->temperature > 0.f, "%s: temperature must be positive", who);
    return RLX_OK;
}
RowGeom geom_of(const rlx_token_rows* r) {
    RowGeom g;
    g.n_tokens = r->n_tokens, g.vocab = r->vocab, g.rows_per_seq = r->rows_per_seq, g.seq_stride = r->seq_stride;
    g.row_stride = r->row_stride, g.temp = r->temperature, g.rtemp = 1.0f / r->temperature;
    g.round_outputs = r->round_outputs;
    return g;
}
int row_grid(long long n_tokens) { return (int)(n_tokens < (1ll << 30) ? n_tokens : (1ll << 30)); }

template <typename T>
int launch_fwd(const void* logits, const int64_t* labels, const RowGeom& g, float* logprob, float* entropy, float* lse,
               hipStream_t s, const int32_t* lp_dst = nullptr, const int32_t* ent_dst = nullptr) {
    const T* x = static_cast<const T*>(logits);
    const dim3 grid(row_grid(g.n_tokens)), block(NT);
    const bool scale = g.temp != 1.0f;
#define RLX_TOK_FWD(ENT, SCALE)                                                                                                   \
    do {                                                                                                                          \
        if (lp_dst != nullptr)                                                                                                    \
            hipLaunchKernelGGL((token_logprob_fwd_kernel<T, ENT, SCALE, true>), grid, block, 0, s, x, labels, g, logprob, entropy, lse, lp_dst, ent_dst); \
        else                                                                                                                      \
            hipLaunchKernelGGL((token_logprob_fwd_kernel<T, ENT, SCALE>), grid, block, 0, s, x, labels, g, logprob, entropy, lse, lp_dst, ent_dst);       \
    } while (0)
    if (entropy) {
        if (scale) RLX_TOK_FWD(true, true); else RLX_TOK_FWD(true, false);
    } else {
        if (scale) RLX_TOK_FWD(false, true); else RLX_TOK_FWD(false, false);
    }
#undef RLX_TOK_FWD
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

template <typename T>
int launch_bwd(const void* logits, const int64_t* labels, const RowGeom& g, const float* lse, const float* entropy,
               const float* d_logprob, const float* d_entropy, void* d_logits, long long dss, long long drs, hipStream_t s,
               const int32_t* lp_dst = nullptr, const int32_t* ent_dst = nullptr) {
    const T* x = static_cast<const T*>(logits);
    T* dx = static_cast<T*>(d_logits);
    const dim3 grid(row_grid(g.n_tokens)), block(NT);
    const bool scale = g.temp != 1.0f;
#define RLX_TOK_BWD(ENT, SCALE)                                                                                                    \
    {                                                                                                                              \
        if (lp_dst != nullptr)                                                                                                     \
            hipLaunchKernelGGL((token_logprob_bwd_kernel<T, ENT, SCALE, true>), grid, block, 0, s, x, labels, g, lse, entropy,     \
                               d_logprob, d_entropy, dx, dss, drs, lp_dst, ent_dst);                                               \
        else                                                                                                                       \
            hipLaunchKernelGGL((token_logprob_bwd_kernel<T, ENT, SCALE, false>), grid, block, 0, s, x, labels, g, lse, entropy,    \
                               d_logprob, d_entropy, dx, dss, drs, lp_dst, ent_dst);                                               \
    }
    if (d_entropy) {
        if (scale) RLX_TOK_BWD(true, true) else RLX_TOK_BWD(true, false)
    } else {
        if (scale) RLX_TOK_BWD(false, true) else RLX_TOK_BWD(false, false)
    }
#undef RLX_TOK_BWD
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" int rlx_token_logprob_fwd(const void* logits, const int64_t* labels, const rlx_token_rows* rows,
                                     float* logprob, float* entropy, float* lse, rlx_stream_t stream) {
    if (int rc = check_rows(rows, "rlx_token_logprob_fwd")) return rc;
    if (rows->n_tokens == 0) return RLX_OK;
    RLX_REQUIRE(logits && labels && logprob && lse, "rlx_token_logprob_fwd: NULL argument");
    const RowGeom g = geom_of(rows);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return rows->dtype == RLX_DTYPE_BF16 ? launch_fwd<__bf16>(logits, labels, g, logprob, entropy, lse, s)
                                         : launch_fwd<float>(logits, labels, g, logprob, entropy, lse, s);
}

extern "C" int rlx_token_logprob_fwd_packed(const void* logits, const int64_t* labels, const rlx_token_rows* rows,
                                            const int32_t* lp_dst, const int32_t* ent_dst, float* logprob, float* entropy, float* lse,
                                            rlx_stream_t stream) {
    if (int rc = check_rows(rows, "rlx_token_logprob_fwd_packed")) return rc;
    if (rows->n_tokens == 0) return RLX_OK;
    RLX_REQUIRE(logits && labels && logprob && lse && lp_dst, "rlx_token_logprob_fwd_packed: NULL argument");
    RLX_REQUIRE((entropy == nullptr) == (ent_dst == nullptr), "rlx_token_logprob_fwd_packed: entropy and ent_dst go together");
    const RowGeom g = geom_of(rows);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return rows->dtype == RLX_DTYPE_BF16 ? launch_fwd<__bf16>(logits, labels, g, logprob, entropy, lse, s, lp_dst, ent_dst)
                                         : launch_fwd<float>(logits, labels, g, logprob, entropy, lse, s, lp_dst, ent_dst);
}

extern "C" int rlx_token_logprob_bwd(const void* logits, const int64_t* labels, const rlx_token_rows* rows,
                                     const float* lse, const float* entropy, const float* d_logprob,
                                     const float* d_entropy, void* d_logits, int64_t d_seq_stride, int64_t d_row_stride,
                                     rlx_stream_t stream) {
    if (int rc = check_rows(rows, "rlx_token_logprob_bwd")) return rc;
    if (rows->n_tokens == 0) return RLX_OK;
    RLX_REQUIRE(logits && labels && lse && d_logprob && d_logits, "rlx_token_logprob_bwd: NULL argument");
    RLX_REQUIRE(d_entropy == nullptr || entropy != nullptr, "rlx_token_logprob_bwd: d_entropy needs the forward's entropy");
    RLX_REQUIRE(d_row_stride >= rows->vocab, "rlx_token_logprob_bwd: bad d_logits row stride");
    const RowGeom g = geom_of(rows);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return rows->dtype == RLX_DTYPE_BF16
               ? launch_bwd<__bf16>(logits, labels, g, lse, entropy, d_logprob, d_entropy, d_logits, d_seq_stride, d_row_stride, s)
               : launch_bwd<float>(logits, labels, g, lse, entropy, d_logprob, d_entropy, d_logits, d_seq_stride, d_row_stride, s);
}

extern "C" int rlx_token_logprob_bwd_packed(const void* logits, const int64_t* labels, const rlx_token_rows* rows, const float* lse,
                                            const float* entropy, const int32_t* lp_dst, const int32_t* ent_dst,
                                            const float* d_logprob, const float* d_entropy, void* d_logits, int64_t d_seq_stride,
                                            int64_t d_row_stride, rlx_stream_t stream) {
    if (int rc = check_rows(rows, "rlx_token_logprob_bwd_packed")) return rc;
    if (rows->n_tokens == 0) return RLX_OK;
    RLX_REQUIRE(logits && labels && lse && d_logprob && d_logits && lp_dst, "rlx_token_logprob_bwd_packed: NULL argument");
    RLX_REQUIRE(d_entropy == nullptr || (entropy != nullptr && ent_dst != nullptr),
                "rlx_token_logprob_bwd_packed: d_entropy needs the forward's entropy and ent_dst");
    RLX_REQUIRE(d_row_stride >= rows->vocab, "rlx_token_logprob_bwd_packed: bad d_logits row stride");
    const RowGeom g = geom_of(rows);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return rows->dtype == RLX_DTYPE_BF16
               ? launch_bwd<__bf16>(logits, labels, g, lse, entropy, d_logprob, d_entropy, d_logits, d_seq_stride, d_row_stride, s, lp_dst, ent_dst)
               : launch_bwd<float>(logits, labels, g, lse, entropy, d_logprob, d_entropy, d_logits, d_seq_stride, d_row_stride, s, lp_dst, ent_dst);
}

extern "C" size_t rlx_token_loss_workspace_bytes(int64_t bsz, int64_t seq) {
    (void)seq;
    return bsz > 0 ? (size_t)bsz * RS * sizeof(double) : 8;
}

extern "C" int rlx_token_loss_fwd(const float* logprobs, const float* old_logprobs, const float* advantages,
                                  const float* ref_logprobs, const float* entropy, const uint8_t* loss_mask, int64_t bsz,
                                  int64_t seq, const rlx_token_loss_params* params, float* g_logp, float* g_entropy,
                                  float* row_weight, float* out, void* workspace, size_t workspace_bytes,
                                  rlx_stream_t stream) {
    RLX_REQUIRE(params != nullptr, "rlx_token_loss_fwd: NULL params");
    RLX_REQUIRE(bsz >= 1 && seq >= 1 && bsz < (1ll << 31), "rlx_token_loss_fwd: bad sizes");
    RLX_REQUIRE(logprobs && old_logprobs && advantages && g_logp && row_weight && out && workspace,
                "rlx_token_loss_fwd: NULL argument");
    RLX_REQUIRE(params->loss_agg >= RLX_AGG_TOKEN_MEAN && params->loss_agg <= RLX_AGG_SEQ_MEAN_TOKEN_MEAN,
                "rlx_token_loss_fwd: unknown loss_agg %d", params->loss_agg);
    RLX_REQUIRE(params->kl_type >= RLX_KL_NONE && params->kl_type <= RLX_KL_K3, "rlx_token_loss_fwd: unknown kl_type %d",
                params->kl_type);
    RLX_REQUIRE(!params->ppo.use_dual_clip || params->ppo.clip_ratio_c > 1.0f,
                "rlx_token_loss_fwd: clip_ratio_c must be greater than 1.0");
    RLX_REQUIRE(!(params->use_entropy && entropy && params->entropy_bonus != 0.f) || g_entropy,
                "rlx_token_loss_fwd: entropy_bonus needs g_entropy");
    if (workspace_bytes < rlx_token_loss_workspace_bytes(bsz, seq)) {
        set_error("rlx_token_loss_fwd: workspace too small");
        return RLX_ENOSPC;
    }
    TokLossArgs a;
    a.logprobs = logprobs, a.old_logprobs = old_logprobs, a.advantages = advantages, a.ref_logprobs = ref_logprobs;
    a.entropy = entropy, a.loss_mask = loss_mask, a.bsz = bsz, a.seq = seq, a.p = *params;
    a.g_logp = g_logp, a.g_entropy = (params->use_entropy && entropy && params->entropy_bonus != 0.f) ? g_entropy : nullptr;
    a.row_weight = row_weight, a.out = out, a.row_sums = static_cast<double*>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(token_loss_rows_kernel, dim3((unsigned)bsz), dim3(256), 0, s, a);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(token_loss_finalize_kernel, dim3(1), dim3(256), 0, s, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_token_loss_bwd(const float* g_logp, const float* g_entropy, const float* row_weight,
                                  const float* grad_out, float* d_logprobs, float* d_entropy, int64_t bsz, int64_t seq,
                                  rlx_stream_t stream) {
    RLX_REQUIRE(bsz >= 1 && seq >= 1 && bsz < (1ll << 31), "rlx_token_loss_bwd: bad sizes");
    RLX_REQUIRE(g_logp && row_weight && grad_out && d_logprobs, "rlx_token_loss_bwd: NULL argument");
    const int gy = (int)((seq + 1023) / 1024 < 64 ? (seq + 1023) / 1024 : 64);
    hipLaunchKernelGGL(token_loss_bwd_kernel, dim3((unsigned)bsz, gy), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g_logp, g_entropy, row_weight, grad_out, d_logprobs, d_entropy, (long long)seq);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_grpo_seq_adv(const float* rewards, const uint8_t* loss_mask, float* advantages, int64_t bsz,
                                int64_t seq, int group_size, float eps, rlx_stream_t stream) {
    RLX_REQUIRE(bsz >= 0 && seq >= 0 && bsz < (1ll << 31), "rlx_grpo_seq_adv: bad sizes");
    RLX_REQUIRE(group_size >= 1 && bsz % group_size == 0, "rlx_grpo_seq_adv: bsz %lld %% group_size %d != 0",
                (long long)bsz, group_size);
    if (bsz == 0 || seq == 0) return RLX_OK;
    RLX_REQUIRE(rewards && loss_mask && advantages, "rlx_grpo_seq_adv: NULL argument");
    const int gy = (int)((seq + 1023) / 1024 < 64 ? (seq + 1023) / 1024 : 64);
    hipLaunchKernelGGL(grpo_seq_adv_kernel, dim3((unsigned)bsz, gy), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rewards, loss_mask, advantages, (long long)seq, group_size, eps);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_categorical_sample(const void* logits, const rlx_token_rows* rows, const void* noise, int top_k, int softmax_lanes,
                                      const float* bin_centers, int n_centers, int64_t* tokens, float* logprob,
                                      float* actions, rlx_stream_t stream) {
    if (int rc = check_rows(rows, "rlx_categorical_sample")) return rc;
    RLX_REQUIRE(rows->vocab <= 1024, "rlx_categorical_sample: at most 1024 categories per row (got %d)", rows->vocab);
    if (rows->n_tokens == 0) return RLX_OK;
    RLX_REQUIRE(logits && tokens, "rlx_categorical_sample: NULL argument");
    RLX_REQUIRE(actions == nullptr || (bin_centers != nullptr && n_centers >= 1), "rlx_categorical_sample: actions need bin_centers");
    RLX_REQUIRE(softmax_lanes == 0 || softmax_lanes == 8 || softmax_lanes == 16, "rlx_categorical_sample: softmax_lanes must be 16 (AVX-512 hosts; 0 = 16) or 8 (AVX2), got %d", softmax_lanes);
    CatArgs a;
    a.lanes = softmax_lanes == 0 ? 16 : softmax_lanes;
    a.logits = logits, a.noise = noise, a.g = geom_of(rows), a.top_k = top_k, a.bin_centers = bin_centers;
    a.n_centers = n_centers, a.tokens = reinterpret_cast<long long*>(tokens), a.logprob = logprob, a.actions = actions;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return rows->dtype == RLX_DTYPE_BF16 ? dispatch_cat<__bf16>(a, s) : dispatch_cat<float>(a, s);
}
