// ppo_step_bf16_parts.h -- device pieces shared by the bf16 fused launches (ppo_step_bf16.hip: column-split workgroups with the
// weights streamed to registers; ppo_step_bf16_rows.hip: row-split waves with the weights staged through an LDS ring).
#pragma once

#include "ppo_step_common.h"

namespace rlx {
namespace b16 {

using namespace loss;
using namespace step;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int XSB = 272;  // bf16 slab row stride (elements): 136 dwords == 8 (mod 16) -> conflict-free b128 fragment reads

typedef float f32x2 __attribute__((ext_vector_type(2)));

// tanh for two values at once, bf16 path: 1 - 2 / (exp(2x) + 1) with v_exp_f32 / v_rcp_f32 and packed f32 arithmetic
// (v_pk_mul / v_pk_add / v_pk_fma).  ~6e-8 absolute error: far inside the 2^-9 relative rounding the result gets next.  The
// f32 kernels keep the polynomial form (2e-7 RELATIVE near zero); here the epilogues are VALU-bound (phase stamps: three of
// them were 18 k of the kernel's 65 k cycles at ~27 issue slots per element) and this is ~10 slots per element.
__device__ __forceinline__ f32x2 tanh2_b(f32x2 x) {
    const f32x2 t = x * 2.8853900817779268f;  // 2 log2(e)
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2 d = e + 1.f;
    f32x2 q;
    q.x = __builtin_amdgcn_rcpf(d.x);
    q.y = __builtin_amdgcn_rcpf(d.y);
    return __builtin_elementwise_fma(q, f32x2{-2.f, -2.f}, f32x2{1.f, 1.f});
}

// A wave's tile values in the accumulator layout (lane (r16, kq) holds rows 4 kq .. 4 kq + 3 of column c = col0 + r16) -> the
// row-major bf16 slab, as 32-BIT words.  As four ds_write_b16 per tile this was the kernel's main source of LDS bank conflicts
// (round-2 counters: SQ_LDS_BANK_CONFLICT 30 % of the LDS-active cycles): stores bank on (a / 4) mod 32 within 32-lane groups,
// the 136-dword row stride puts rows r and r + 4 (lanes kq and kq + 1 of one group) on the same banks, and two lanes share every
// dword.  Instead neighbouring lanes swap halves (one DPP quad_perm [1, 0, 3, 2]): the even lane then holds columns (c, c + 1) of
// rows 4 kq, 4 kq + 1 and its odd neighbour the same columns of rows 4 kq + 2, 4 kq + 3 -- two ds_write_b32 per tile.  Which of its
// two rows a lane stores FIRST alternates with kq: one instruction's 32-lane group then covers rows {0, 2, 5, 7} (mod 8), i.e. row
// offsets {0, 16, 8, 24} (mod 32 dwords) x 8 dwords each = all 32 banks once.  Same values, same addresses: bit-identical slab.
__device__ __forceinline__ void store_slab_quad(__bf16* Xb, int row_base /* rt * 16 */, int col0 /* multiple of 16 */, bf16x4 v) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4, odd = lane & 1;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 p = __builtin_bit_cast(u32x2, v);                     // p.x = rows 0, 1 of this column; p.y = rows 2, 3
    const unsigned keep = odd ? p.y : p.x, send = odd ? p.x : p.y;
    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    const unsigned lo = odd ? recv : keep, hi = odd ? keep : recv;   // lo: the even column's two rows, hi: the odd column's
    const unsigned w0 = __builtin_amdgcn_perm(hi, lo, 0x05040100u);  // first row of the pair:  (lo.low16, hi.low16)
    const unsigned w1 = __builtin_amdgcn_perm(hi, lo, 0x07060302u);  // second row of the pair: (lo.high16, hi.high16)
    const int row = row_base + 4 * kq + 2 * odd, flip = kq & 1;
    unsigned* base = reinterpret_cast<unsigned*>(Xb + col0 + (r16 & ~1));
    base[(row + flip) * (XSB / 2)] = flip ? w1 : w0;
    base[(row + 1 - flip) * (XSB / 2)] = flip ? w0 : w1;
}

// tanh'(z) = 1 - h^2 from the rounded activation the forward sweep kept (two bf16 -> f32 shifts and one fma per element:
// cheaper than carrying 96 f32 registers through the whole kernel, which pushed it into scratch)
__device__ __forceinline__ float dtanh_b(__bf16 h) {
    const float hf = (float)h;
    return fmaf(-hf, hf, 1.f);
}

// ---- the f32 heads on the matrix pipe ----------------------------------------------------------------------------------
// The heads keep f32 weights and f32 outputs (file header).  An f32 number is exactly the sum of three bf16 numbers
// (hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid): 3 x 8 significand bits, the differences are exact in f32), and a
// product of two bf16 numbers is exact in f32 -- so  x . w  with x in bf16 is three bf16 MFMAs with f32 accumulation: the same
// real-number sum as the f32 dot product, only the order of the f32 additions differs.  As VALU dot products the head phases
// were LDS-bound (every (row, output) pair re-read its 1 KiB weight row: ~6 k cycles of b128 reads per tile).
struct Split3 {
    bf16x8 hi, mid, lo;
};
__device__ __forceinline__ Split3 split3(const float (&w)[8]) {
    Split3 q;
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
__device__ __forceinline__ f32x4 mfma3(bf16x8 a, const Split3& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.mid, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.lo, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma3(const Split3& a, bf16x8 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b, acc, 0, 0, 0);
}

struct TileGeom {
    int nrb;
    __host__ __device__ size_t mat() const { return (size_t)16 * nrb * 512; }  // elements per [M][256] image
};

// STAMPS: the development build of the same kernel with phase stamps (tools/phase_times.py); reading the cycle counter orders
// the surrounding memory operations (+5 us measured), so the product instantiation carries none.
template <bool ON>
struct StampsT {
    long long* buf;
    int n;
    __device__ __forceinline__ void mark() {
        if constexpr (ON) {
            if (buf != nullptr && blockIdx.x == 0 && blockIdx.y == 1 && threadIdx.x == 0) buf[n] = (long long)clock64();  // the policy network's tile 0
            ++n;
        }
    }
};

// The small parameter inputs of a tile: the three hidden layers' biases, the f32 head image + bias, the policy's log-std.
// issue() requests them into registers (clamped, unconditional), commit() writes them to LDS: sBias [3][HID], W4s / b4s, and
// sStd = std | var | log(std) per output (the same expf / fmul / logf the per-element code used to repeat).
template <int NT>
struct SmallInputsB {
    static constexpr int NB = (3 * HID + NT - 1) / NT, NH = (MAX_OUT * 64 + NT - 1) / NT;
    float bv[NB];
    f32x4 hw[NH];
    float b4v, lsv;
    bool has_b4;
    __device__ __forceinline__ void issue(const float* __restrict__ params, const rlx_mlp_layout& lay, int y, int n_out) {
        const int tid = threadIdx.x;
        const int wave0 = __builtin_amdgcn_readfirstlane(tid) & ~63;  // (first lane of this wave, in a SCALAR register: the predicates
        //                                                                 below are scalar branches, no exec-mask divergence)
        // (lay lives in the kernel arguments: indexing it with a per-lane value would turn into a VECTOR load of the argument
        //  block and a dependent round trip -- the three offsets are read as scalars and selected per lane.)
        const long long ob0 = lay.off_b[y][0], ob1 = lay.off_b[y][1], ob2 = lay.off_b[y][2];
        // A vector-memory instruction occupies the CU's address unit for ~16 cycles whatever its width, and every wave of the
        // (two) resident workgroups requests its inputs at the same moment: round 6's phase stamps put 3.6-4.6 k cycles of pure
        // ISSUE in front of the fused launch's first wait.  So a wave none of whose lanes holds a useful index skips the
        // instruction (a scalar branch: no lane-level divergence, the clamped addresses stay for the partially useful waves).
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            bv[u] = 0.f;
            if (wave0 + u * NT < 3 * HID) {
                const int i = min(tid + u * NT, 3 * HID - 1), l = i / HID;
                bv[u] = params[(l == 0 ? ob0 : l == 1 ? ob1 : ob2) + i % HID];
            }
        }
        const float* W4 = params + lay.off_w[y][3];
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            hw[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (wave0 + u * NT < n_out * 64) {
                const int f = min(tid + u * NT, n_out * 64 - 1), o = f >> 6, c4 = (f & 63) * 4;
                hw[u] = *reinterpret_cast<const f32x4*>(W4 + (size_t)o * HID + c4);
            }
        }
        // unconditional within the wave, clamped (a load under a LANE-level branch gets its wait under the branch too)
        // b4 / log-std: every wave loads them (two instructions), unconditionally -- under the wave-0 predicate the compiler moved
        // commit()'s expf(lsv) up into the predicated block and put a full wait in front of wave 0's weight requests
        const long long ob3 = lay.off_b[y][3];
        has_b4 = ob3 >= 0;  // (applied in commit(): a select on the loaded value here would put its wait here)
        const int oc = min(tid, n_out - 1);
        b4v = params[(has_b4 ? ob3 : 0) + oc];
        lsv = params[y == 1 ? lay.off_logstd + oc : 0];
    }
    __device__ __forceinline__ void commit(int n_out, float* sBias, float* W4s, float* b4s, float* sStd) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (tid + u * NT < 3 * HID) sBias[tid + u * NT] = bv[u];
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int f = tid + u * NT;
            if (f < n_out * 64) *reinterpret_cast<f32x4*>(W4s + (f >> 6) * W4S + (f & 63) * 4) = hw[u];
        }
        if (tid < n_out) {  // (the value network's lanes compute the std terms of a junk word: never read)
            b4s[tid] = has_b4 ? b4v : 0.f;
            const float stdv = expf(lsv);
            sStd[tid] = stdv;
            sStd[MAX_OUT + tid] = fmul(stdv, stdv);
            sStd[2 * MAX_OUT + tid] = logf(stdv);
        }
    }
};


// ---- the loss pass of the fused optimizer-step launches (bf16: ppo_step_bf16.hip; f32 through bf16 splits: ppo_step_f32x.hip) ----
// In: the head's two k-half partials in sG / sLp, the tile's loss inputs in sOld / sAct / sAdv / sRet, std | var | log(std) in sStd.
// Out: d(loss)/d(head output) in sHead (policy: d/d mean; value: d/d value), d/d logstd per row in sLp, the tile's metric partial row in
// a.loss_part.  Ends with an LDS barrier.  sLacc: [NS][64] doubles of scratch (the dead h3 slab).
// Lane l's value of lane l + J of the same 16-lane DPP row (0 past the row's end): one VALU slot, where __shfl_down is a
// ds_bpermute, i.e. a round trip through the LDS pipe.  The loss pass's row sums use it: a row's n_out lanes are neighbours and
// (n_out dividing 16) never straddle a DPP row.
template <int J>
__device__ __forceinline__ float dpp_row_shl(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + J /* row_shl:J */, 0xF, 0xF, true));
}
// x[k] + (the row-mates to the right, in lane order): lp_0 + lp_1 + ... + lp_{n-1} with the additions in that order
template <int J, int OP>
__device__ __forceinline__ void row_sums(float& lp, float lpe, float& old, float olde, float& px, float pxe, int n_out, bool with_px) {
    if constexpr (J < OP) {
        if (J < n_out) {  // (wave-uniform)
            lp = fadd(lp, dpp_row_shl<J>(lpe));
            old = fadd(old, dpp_row_shl<J>(olde));
            if (with_px) px = fadd(px, dpp_row_shl<J>(pxe));
        }
        row_sums<J + 1, OP>(lp, lpe, old, olde, px, pxe, n_out, with_px);
    }
}
// Sum of a double over the wave's 64 lanes, result in every lane ... of the LAST row (lane 63 holds the total; callers read it
// there): DPP row shifts inside the 16-lane rows, two row broadcasts across them.  12 VALU moves + 6 f64 additions instead of
// six butterfly steps of two ds_bpermute each.
__device__ __forceinline__ double dpp_wave_sum_to_lane63(double v) {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
#define RLX_DPP_ADD(CTRL, ROWMASK)                                                                                          \
    {                                                                                                                        \
        const i32x2 b = __builtin_bit_cast(i32x2, v);                                                                         \
        i32x2 s;                                                                                                             \
        s[0] = __builtin_amdgcn_update_dpp(0, b[0], CTRL, ROWMASK, 0xF, false);                                               \
        s[1] = __builtin_amdgcn_update_dpp(0, b[1], CTRL, ROWMASK, 0xF, false);                                               \
        v += __builtin_bit_cast(double, s);                                                                                   \
    }
    RLX_DPP_ADD(0x111, 0xF)  // row_shr:1
    RLX_DPP_ADD(0x112, 0xF)  // row_shr:2
    RLX_DPP_ADD(0x114, 0xF)  // row_shr:4
    RLX_DPP_ADD(0x118, 0xF)  // row_shr:8   -> lane 15 of every row holds the row's sum
    RLX_DPP_ADD(0x142, 0xA)  // row_bcast:15 into rows 1 and 3: lane 31 = rows 0 + 1, lane 63 = rows 2 + 3
    RLX_DPP_ADD(0x143, 0xC)  // row_bcast:31 into rows 2 and 3: lane 63 = everything
#undef RLX_DPP_ADD
    return v;
}

struct LossLds {
    float *b4s, *sHead, *sLp, *sG, *sD, *sOld, *sAct, *sAdv, *sRet, *sStd;
    double* sLacc;
    const double* sNm;
};
template <int BM, int NW, bool DEC, typename TS>
__device__ __forceinline__ void fused_loss_pass(const StepArgs& a, int y, int tile, long long m0, const LossLds& L, TS& ts) {
    constexpr int OP = MAX_OUT;  // bound of the unrolled row sums
    constexpr int NT = 64 * NW;
    const rlx_mlp_layout& lay = a.lay;
    const rlx_ppo_loss_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long M = a.M;
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const int K = p.raw_per_adv, S = p.sub_per_adv, R = K / S;
    const int npr = lay.act_dim / K;
    const long long n_adv = M * npr;
    const bool has_mask = a.loss_mask != nullptr, has_msum = a.loss_mask_sum != nullptr;
    const bool ratio_mode = p.max_episode_steps > 0 && has_mask && has_msum;
    float *b4s = L.b4s, *sHead = L.sHead, *sLp = L.sLp, *sG = L.sG, *sD = L.sD, *sOld = L.sOld, *sAct = L.sAct, *sAdv = L.sAdv,
          *sRet = L.sRet, *sStd = L.sStd;
    double* sLacc = L.sLacc;
    const double* sNm = L.sNm;
    const double nm = has_mask ? sNm[0] : 0.0;
    const Denoms den = denominators(p, n_adv, nm, has_mask, has_msum);
    const float half_delta = (float)(0.5 * (double)p.huber_delta);
    double lacc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) lacc[k] = 0.0;
    DecoupledMode dmode{};
    if constexpr (DEC) dmode = decoupled_mode_now(a.dec);

    // Fast path (the embodied shapes: one loss element per row -- action_level log-probs, one sub-group --, or a value head of
    // at most 64 / BM outputs): element math, loss element and dOut in ONE pass with no barrier in between.  A row's n_out lanes
    // are neighbours (idx = row * n_out + o), so the row leader (o == 0) collects the per-dimension log-probs with n_out - 1
    // lane shifts and adds them in the reference's order (j ascending, starting from 0.f), evaluates the loss element and hands
    // d(loss)/d(logprob) back to its lanes.  Slot r of the metric partials is row r's element, exactly what wave 0's lane r
    // held in the general path: the butterfly sums -- and so the metrics -- are bit-identical between the two paths.
    const bool fast = a.merged_loss_pass && BM * n_out <= NT && (y == 0 ? BM * n_out <= 64 : (npr == 1 && S == 1 && 64 % n_out == 0));
    if (fast) {
        for (int i = tid; i < NS * 64; i += NT)
            if ((i & 63) >= (y == 0 ? BM * n_out : BM)) sLacc[i] = 0.0;  // the slots no element owns
        const bool mine = tid < BM * n_out;
        const int row = mine ? tid / n_out : 0, o = mine ? tid % n_out : 0;
        const bool valid = mine && m0 + row < M;
        float sv = fadd(sG[row * MAX_OUT + o], sLp[row * MAX_OUT + o]);
        if (lay.off_b[y][3] >= 0) sv = fadd(sv, b4s[o]);
        if (y == 1) {
            const float d = fsub(sAct[row * MAX_OUT + o], sv);
            const float var = sStd[MAX_OUT + o], log_scale = sStd[2 * MAX_OUT + o];
            const float lpe = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
            const float olde = sOld[row * MAX_OUT + o];
            float lp = fadd(0.f, lpe), old = fadd(0.f, olde);
            float pxe = 0.f, px = 0.f;  // decoupled, given proximal policy: its per-dimension log-probs, summed like `old`
            if constexpr (DEC) {
                if (dmode.mode == RLX_PROX_GIVEN && mine) pxe = a.dec.proximal[(size_t)min(m0 + row, M - 1) * lay.act_dim + o];
                px = fadd(0.f, pxe);
            }
            row_sums<1, OP>(lp, lpe, old, olde, px, pxe, n_out, DEC);  // only the leaders' sums are used
            float gs = 0.f;
            if (valid && o == 0) {
                const long long e = m0 + row;
                const bool on = has_mask ? a.loss_mask[e] != 0 : true;
                float w = 1.f;
                if (ratio_mode) w = ((float)a.loss_mask_sum[e] * 1.0f) / (float)p.max_episode_steps;
                lacc[S_NM] += on ? 1.0 : 0.0;
                if constexpr (DEC) {  // sum form: the denominator is out[RLX_PPO_ACTOR_GRAD_SCALE] of the finished row
                    const float vb = a.dec.versions != nullptr ? a.dec.versions[(size_t)e * lay.act_dim] : 0.f;
                    gs = a.grad_out * decoupled_actor_elem(p, dmode, lp, old, px, vb, sAdv[row * MAX_OUT], on, w, ratio_mode, lacc, S_VLOSS);
                } else {
                    const float g = actor_elem(p, lp, old, sAdv[row * MAX_OUT], on, w, ratio_mode, lacc);
                    gs = (a.grad_out * (float)(1.0 / den.actor)) * g;
                }
            }
            gs = __shfl(gs, (lane - o) & 63, 64);  // from the row leader
            float dmu = 0.f, dls = 0.f;
            if (valid) {
                dmu = gs * d / var;
                dls = gs * (d * d / var - 1.f);
            }
            if (mine) {
                sHead[row * MAX_OUT + o] = dmu;
                sLp[row * MAX_OUT + o] = dls;
                if (o == 0) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) sLacc[k * 64 + row] = lacc[k];
                }
            }
        } else if (mine) {
            float gv = 0.f;
            if (valid && p.has_critic) {
                const long long e = (m0 + row) * n_out + o;
                const bool on = has_mask ? a.loss_mask[e] != 0 : true;
                float w = 1.f;
                if (ratio_mode) w = ((float)a.loss_mask_sum[e] * 1.0f) / (float)p.max_episode_steps;
                gv = (a.grad_out * (float)(1.0 / den.critic)) *
                     critic_elem(p, sv, sAdv[row * MAX_OUT + o], sRet[row * MAX_OUT + o], on, w, ratio_mode, half_delta, lacc);
            }
            sHead[row * MAX_OUT + o] = gv;
#pragma unroll
            for (int k = 0; k < NS; ++k) sLacc[k * 64 + tid] = lacc[k];
        }
        lds_barrier();
        ts.mark();
        ts.mark();  // (the general path's two intermediate stamps)
    } else {
        for (int idx = tid; idx < BM * n_out; idx += NT) {
            const int row = idx / n_out, o = idx % n_out;
            float s = fadd(sG[row * MAX_OUT + o], sLp[row * MAX_OUT + o]);
            if (lay.off_b[y][3] >= 0) s = fadd(s, b4s[o]);
            sHead[row * MAX_OUT + o] = s;
            if (y == 1) {
                const float d = fsub(sAct[row * MAX_OUT + o], s);
                const float var = sStd[MAX_OUT + o];
                const float log_scale = sStd[2 * MAX_OUT + o];
                sLp[row * MAX_OUT + o] = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
                sD[row * MAX_OUT + o] = d;
            }
        }
        lds_barrier();
        ts.mark();
        // wave 0 walks the tile's loss elements (a fixed lane <-> element assignment keeps the f64 metric sums reproducible) and
        // parks its 16 per-lane partial sums in the slab (h3 is dead: the head gradients and dZ3 work from registers and sHead);
        // behind the barrier every wave butterfly-sums two of the 16 slots while all lanes run the dOut pass -- the same 64-lane
        // butterfly wave 0 used to run 16 times in a row (~7 k cycles with the other seven waves parked at the barrier).
        if (y == 1) {
            for (int idx = tid; idx < BM * npr && wave == 0; idx += 64) {
                const int row = idx / npr, c = idx % npr;
                if (m0 + row >= M) continue;
                const long long e = (m0 + row) * npr + c;
                const bool on = has_mask ? a.loss_mask[e] != 0 : true;
                float w = 1.f;
                if (ratio_mode) w = ((float)a.loss_mask_sum[e] * 1.0f) / (float)p.max_episode_steps;
                const float adv = sAdv[row * MAX_OUT + c];
                lacc[S_NM] += on ? 1.0 : 0.0;
                const float* olp = sOld + row * MAX_OUT + c * K;
                for (int s = 0; s < S; ++s) {
                    float lp = 0.f, old = 0.f;
                    for (int j = 0; j < R; ++j) {
                        lp = fadd(lp, sLp[row * MAX_OUT + c * K + s * R + j]);
                        old = fadd(old, olp[s * R + j]);
                    }
                    if constexpr (DEC) {  // the slice's raw entries are [e * K + s * R, + R) of the [M, act_dim] arrays
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
        } else {
            for (int idx = tid; idx < BM * n_out && wave == 0; idx += 64) {
                const int row = idx / n_out, o = idx % n_out;
                float gv = 0.f;
                if (m0 + row < M && p.has_critic) {
                    const long long e = (m0 + row) * n_out + o;
                    const bool on = has_mask ? a.loss_mask[e] != 0 : true;
                    float w = 1.f;
                    if (ratio_mode) w = ((float)a.loss_mask_sum[e] * 1.0f) / (float)p.max_episode_steps;
                    gv = (a.grad_out * (float)(1.0 / den.critic)) *
                         critic_elem(p, sHead[row * MAX_OUT + o], sAdv[row * MAX_OUT + o], sRet[row * MAX_OUT + o], on, w, ratio_mode, half_delta, lacc);
                }
                sHead[row * MAX_OUT + o] = gv;
            }
        }
        if (wave == 0) {
    #pragma unroll
            for (int k = 0; k < NS; ++k) sLacc[k * 64 + lane] = lacc[k];
        }
        lds_barrier();
        ts.mark();
        if (y == 1) {
            for (int idx = tid; idx < BM * n_out; idx += NT) {
                const int row = idx / n_out, o = idx % n_out;
                float dmu = 0.f, dls = 0.f;
                if (m0 + row < M) {
                    const float dlp = sG[row * MAX_OUT + (o / K) * S + (o % K) / R];
                    const float var = sStd[MAX_OUT + o], d = sD[row * MAX_OUT + o];
                    dmu = dlp * d / var;
                    dls = dlp * (d * d / var - 1.f);
                }
                sHead[row * MAX_OUT + o] = dmu;
                sLp[row * MAX_OUT + o] = dls;
            }
        }
    }
    {
        double* lp = a.loss_part + ((size_t)tile * 2 + y) * NS;
        for (int k = wave; k < NS; k += NW) {
            const double v = dpp_wave_sum_to_lane63(sLacc[k * 64 + lane]);
            if (lane == 63) lp[k] = v;
        }
    }
    lds_barrier();
    ts.mark();
}

}  // namespace b16
}  // namespace rlx
