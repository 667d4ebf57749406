// ppo_step_common.h -- pieces shared by the f32 (ppo_step.hip) and bf16 (ppo_step_bf16.hip) fused launches:
// geometry, LDS barrier, fast tanh, kernel argument structs, host-side planning.
#pragma once

#include <stdlib.h>

#include <algorithm>

#include "ppo_loss_math.h"
#include "rlx_common.h"

namespace rlx {
namespace step {

using namespace loss;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HID = 256;   // hidden width (fixed by the reference: hidden_sizes=(256,256,256))
constexpr int XS = 264;    // activation slab row stride (floats); stride % 16 == 8 -> conflict-free b128 fragment reads
constexpr int KPAD = 32;    // the K loop advances 32 at a time: the slab's k tail is zero-padded to a multiple of it
constexpr int W4S = 260;   // head weight image row stride
constexpr int MAX_OUT = 16;  // head outputs supported by the fused kernels (act_dim, val_dim)
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

template <int RT, int NW>
struct Geo {
    static constexpr int BM = 16 * RT;          // rows per workgroup tile
    static constexpr int NT = 64 * NW;          // threads
    static constexpr int CT = HID / (16 * NW);  // 16-column tiles per wave
    // LDS: slab [BM][XS] | head image [MAX_OUT][W4S] + bias [MAX_OUT] | 4 x [BM][MAX_OUT] loss scratch | 4 KiB reduction scratch
    static constexpr int AUX_FLOATS = MAX_OUT * W4S + MAX_OUT + 4 * BM * MAX_OUT;
    static constexpr size_t LDS_BYTES = (size_t)(BM * XS + AUX_FLOATS) * sizeof(float) + 4096;
};

// LDS-only workgroup barrier.  __syncthreads() also fences global memory, i.e. waits vmcnt(0): after a flush of the
// slab to HBM every barrier would stall for the stores' full round trip.  Nothing a workgroup writes to global memory
// is read back by it, so only the LDS traffic needs ordering.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// tanh to ~2e-7 absolute: odd polynomial below 0.3, 1 - 2 / (exp(2x) + 1) above (v_exp_f32 + v_rcp_f32), select instead
// of a branch.  libm's tanhf costs ~4x as many VALU slots; 32-64 of them per lane per layer were a third of the kernel.
__device__ __forceinline__ float fast_tanh(float x) {
    const float x2 = x * x;
    const float poly = x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
    const float e = __expf(2.f * x);
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    return fabsf(x) < 0.3f ? poly : big;
}

// Weight image in MFMA fragment order ("tiles"), rebuilt from the flat parameters after every optimizer step:
// a 16 (n) x 16 (k) tile is 1 KiB contiguous, float4 number l = kq*16 + r16 of it holds W[n0 + r16][k0 + 4*kq .. +3], so a
// wave's fragment load is ONE fully coalesced 1 KiB read and consecutive tiles of a column block are consecutive in
// memory.  Per network: W1 (k zero-padded to 64), W2, W3, W2^T, W3^T (the transposes feed the backward-data GEMMs).
struct Tiles {
    static constexpr int K1P = 64;  // first-layer K after padding (obs_dim <= 64 here)
    __host__ __device__ static constexpr size_t per_net() { return (size_t)HID * K1P + 4 * (size_t)HID * HID; }
    // m: 0 = W1, 1 = W2, 2 = W3, 3 = W2^T, 4 = W3^T
    __host__ __device__ static constexpr size_t mat(int y, int m) {
        return y * per_net() + (m == 0 ? 0 : (size_t)HID * K1P + (size_t)(m - 1) * HID * HID);
    }
};

// Phase stamps (development only): block 0 / thread 0 writes the shader clock at phase boundaries into a buffer set
// with rlx_dev_set_timing_buffer(); the product never sets one.
struct Stamps {
    long long* buf;
    int n;
    __device__ __forceinline__ void mark() {
        if (buf != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) buf[n] = (long long)clock64();
        ++n;
    }
};
// Ablation switches (development only: tools/bench_step.py times them; the product always runs ABL == 0)
constexpr int ABL_NO_TANH = 1, ABL_NO_FLUSH = 2, ABL_NO_WLOAD = 4, ABL_NO_MFMA = 8;

// stage the head weight [n_out][256] (+ bias) into LDS with a padded row stride
__device__ __forceinline__ void stage_head(const float* __restrict__ W4, const float* __restrict__ b4, int n_out, float* W4s,
                                           float* b4s, int nthreads) {
    for (int f = threadIdx.x; f < n_out * 64; f += nthreads) {
        const int o = f >> 6, c4 = (f & 63) * 4;
        *reinterpret_cast<f32x4*>(W4s + o * W4S + c4) = *reinterpret_cast<const f32x4*>(W4 + (size_t)o * HID + c4);
    }
    if ((int)threadIdx.x < n_out) b4s[threadIdx.x] = b4 ? b4[threadIdx.x] : 0.f;
}

struct ValueJob {
    const float* states;
    long long m;
    float* values;         // [m, val_dim] or nullptr
    float* rewards;        // [m, chunk] in place, or nullptr
    const uint8_t* flags;  // [m, chunk]
    int chunk;
    float gamma;
    const float* env_rewards;  // optional fused env-row store (see rlx_value_job)
    const uint8_t* env_term;
    const uint8_t* env_trunc;
    uint8_t* done_row;
    uint8_t* term_row;
    uint8_t* trunc_row;
    int flag_is_trunc;
};

// value-only job epilogue for output o of row `row` (global row index): value store, bootstrap fold, env-row store
// The env-row inputs of a value job's row, requested at kernel start (chunk == 1, the embodied case) so that the job's tail does
// not add two dependent memory round trips behind the head.
struct ValuePre {
    float r;
    unsigned b0, b1;  // env path: termination / truncation byte; flags path: b0 = the flag
};
// Straight-line, unconditional loads (pointer selects, no branches): a load under a branch -- or a merge with a constant -- gets
// its wait placed right behind it.  `safe` is any readable address for the lanes / jobs that have nothing to fetch.
__device__ __forceinline__ ValuePre value_job_prefetch(const ValueJob& v, bool live, size_t row, const float* safe) {
    const bool on = live && v.rewards != nullptr, env = on && v.env_rewards != nullptr;
    const size_t i = on ? row * v.chunk + (v.chunk - 1) : 0;
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(safe);
    ValuePre q;
    q.b0 = (env ? v.env_term : on ? v.flags : sb)[i];
    q.b1 = (env ? v.env_trunc : sb)[i];
    q.r = (env ? v.env_rewards : on ? (const float*)v.rewards : safe)[i];
    return q;
}
__device__ __forceinline__ void value_job_output(const ValueJob& v, size_t g, size_t row, int o, float s, const ValuePre* pre = nullptr) {
    if (v.values) v.values[g] = s;
    if (o != 0 || v.rewards == nullptr) return;
    if (v.env_rewards != nullptr) {
        for (int c = 0; c < v.chunk; ++c) {
            const size_t i = row * v.chunk + c;
            const bool last = c == v.chunk - 1, have = pre != nullptr && last;
            const uint8_t te = have ? pre->b0 != 0 : v.env_term[i] != 0, tr = have ? pre->b1 != 0 : v.env_trunc[i] != 0;
            v.term_row[i] = te;
            v.trunc_row[i] = tr;
            v.done_row[i] = te | tr;  // dones = terminations | truncations (maniskill_env.py:343-350)
            float r = have ? pre->r : v.env_rewards[i];
            const bool flag = v.flag_is_trunc ? tr != 0 : (te | tr) != 0;
            if (last && flag) r = fadd(r, fmul(v.gamma, s));  // env_worker.py:744-758
            v.rewards[i] = r;
        }
    } else {  // r[:, -1] += gamma * V(final_obs)[:, 0] where flags[:, -1]
        const size_t i = row * v.chunk + (v.chunk - 1);
        if (pre != nullptr) {
            if (pre->b0) v.rewards[i] = fadd(pre->r, fmul(v.gamma, s));
        } else if (v.flags[i]) {
            v.rewards[i] = fadd(v.rewards[i], fmul(v.gamma, s));
        }
    }
}
struct RolloutArgs {
    long long* stamps;
    const float* params;
    const float* tiles;
    rlx_mlp_layout lay;
    const float* states;   // policy job (may be nullptr with M == 0)
    const float* eps;
    long long M;
    float* action;
    float* logprob;
    float* value;
    float* states_copy;
    ValueJob vj[2];
    int tiles_policy, tiles_vj0, tiles_vj1;
};

// Decoupled (asynchronous PPO) actor loss inside the fused step (rlx_ppo_step_args.decoupled): the element math is
// loss::decoupled_actor_elem, the actor gradients leave in sum form (their denominator -- the count of the behaviour mask --
// needs every tile's forward), the version sum rides in slot S_VLOSS of the actor network's partial row.
struct DecoupledArgs {
    int on;
    loss::DecoupledMode mode;
    const float* v_theta_dev;   // optional: current version read at execution time (graph replay under the next version)
    const float* proximal;      // [M, act_dim] (RLX_PROX_GIVEN)
    const float* versions;      // [M, act_dim] or nullptr
};
#if defined(__HIPCC__)
__device__ __forceinline__ loss::DecoupledMode decoupled_mode_now(const DecoupledArgs& d) {
    loss::DecoupledMode m = d.mode;
    if (d.v_theta_dev != nullptr) m.v_theta = *d.v_theta_dev;
    return m;
}
#endif

struct StepArgs {
    long long* stamps;
    const float* params;
    const float* tiles;         // fragment-tile weight image (struct Tiles), built by pack_tiles_kernel just before
    rlx_mlp_layout lay;
    const float* states;        // [M, D]
    const float* action;        // [M, act_dim]
    const float* old_logprobs;  // [M, act_dim]
    const float* advantages;    // [M * adv_per_row]
    const float* prev_values;   // [M * adv_per_row] (has_critic)
    const float* returns;
    const uint8_t* loss_mask;   // [M * adv_per_row] or nullptr
    const int64_t* loss_mask_sum;
    long long M;
    rlx_ppo_loss_params p;
    float grad_out;             // d(total)/d(loss) of this micro-batch (1 / gradient_accumulation)
    int merged_loss_pass;       // bf16 launch: element math + loss element + dOut in one pass where the shapes allow (development: RLX_FUSED_MERGED=0)
    int xcd_rows;               // bf16 launch: workgroup -> row-tile map that gives every XCD one CONTIGUOUS eighth of the rows (see the kernel)
    float* h;                   // [2 nets][2][M][256]  hidden activations 1, 2   (B operands of the weight gradients)
    float* dz;                  // [2 nets][3][M][256]  pre-activation gradients  (A operands)
    float* head_part;           // [head_parts][2][head_stride]  per-32-row head gradients: dW4 [n_out][256], db4, dlogstd [n_out]
    double* loss_part;          // [tiles][2][NS]
    int head_stride;
    DecoupledArgs dec;          // dec.on: the decoupled actor loss (the DEC instantiations)
};

struct DwArgs {
    long long* stamps;
    rlx_mlp_layout lay;
    const float* states;
    const float* h;          // [2][2][M][256]
    const float* dz;         // [2][3][M][256]
    const float* head_part;  // [tiles][2][head_stride]
    const double* loss_part; // [tiles][2][NS]
    long long M;
    int rows_per_slab;       // multiple of 32
    int slabs;
    int tiles;               // loss partial slots per network (StepPlan.loss_slots)
    int head_parts;          // head-gradient partial slots per network
    int head_stride;
    int gemm_items;          // slabs * 20, the grid holds round_up(gemm_items, 8) GEMM blocks
    int repeat;              // 1; development (RLX_DW_REPEAT = 2, TIMING ONLY -- the sums come out doubled): the bf16 LDS-DMA kernel walks
                             // its k-blocks twice, the second time with its operands warm in its XCD's L2
    float* grads;            // [slabs][n_params]
    rlx_ppo_loss_params p;
    int has_mask, has_msum;
    float* out;              // metric row (RLX_PPO_OUT_FLOATS)
    int decoupled;           // the row is the rlx_dppo_out one (finalize_row_decoupled)
    int dec_use_threshold;
};

#if defined(__HIPCC__)
// Metric block of the weight-gradient launches: the per-tile partial rows of both networks -> the metric row.  `scratch` holds
// NS doubles per wave of the block.  (Decoupled loss: the actor network's rows -- odd i -- carry the version sum in slot S_VLOSS.)
__device__ __forceinline__ void metric_block(const DwArgs& a, double* scratch, int tid, int nthreads) {
    using namespace loss;
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    const long long n_adv = a.M * (a.lay.act_dim / a.p.raw_per_adv);
    if (!a.decoupled) {
        for (int i = tid; i < a.tiles * 2; i += nthreads) {
#pragma unroll
            for (int k = 0; k < NS; ++k) acc[k] += a.loss_part[(size_t)i * NS + k];
        }
        block_sum<NS>(acc, scratch);
        if (tid == 0) finalize_row(a.p, n_adv, a.has_mask != 0, a.has_msum != 0, acc, a.out);
        return;
    }
    double ver[1] = {0.0};
    for (int i = tid; i < a.tiles * 2; i += nthreads) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const double v = a.loss_part[(size_t)i * NS + k];
            if (k == S_VLOSS && (i & 1)) ver[0] += v;
            else acc[k] += v;
        }
    }
    block_sum<NS>(acc, scratch);
    block_sum<1>(ver, scratch);
    if (tid == 0) finalize_row_decoupled(a.p, a.dec_use_threshold != 0, n_adv, a.has_mask != 0, a.has_msum != 0, acc, ver[0], a.out);
}

// Head-gradient block of the weight-gradient launches: slab s takes the 32-row partials t == s (mod slabs), thread = hidden
// column j.  The sums run in ascending t like a plain loop, but the loads do not: a loop of `acc += part[t]` costs one
// memory round trip PER PARTIAL (the partials were written by other XCDs' workgroups: ~0.4 us each from the memory-side
// cache; 8 outputs x 11 partials = 35 us -- this tail, not the GEMM blocks, set the launch's duration), so HB partials of OB
// outputs are requested together (clamped, unconditional loads) and only then added.
template <int OB = 4, int HB = 12>
__device__ __forceinline__ void head_reduce_block(const DwArgs& a, int s, int y, int j) {
    const rlx_mlp_layout& lay = a.lay;
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    float* slab = a.grads + (size_t)s * lay.n_params;
    const int cnt = (a.head_parts - s + a.slabs - 1) / a.slabs;  // partials of this slab (>= 1: slabs <= head_parts)
    const size_t pstride = (size_t)a.slabs * 2 * a.head_stride;  // from partial t to t + slabs
    const float* base = a.head_part + ((size_t)s * 2 + y) * a.head_stride;
    for (int o0 = 0; o0 < n_out; o0 += OB) {
        float acc[OB];
#pragma unroll
        for (int q = 0; q < OB; ++q) acc[q] = 0.f;
        for (int t0 = 0; t0 < cnt; t0 += HB) {
            float x[OB][HB];
#pragma unroll
            for (int q = 0; q < OB; ++q)
#pragma unroll
                for (int u = 0; u < HB; ++u)
                    x[q][u] = base[(size_t)min(t0 + u, cnt - 1) * pstride + (size_t)min(o0 + q, n_out - 1) * HID + j];
#pragma unroll
            for (int u = 0; u < HB; ++u)
                if (t0 + u < cnt) {
#pragma unroll
                    for (int q = 0; q < OB; ++q) acc[q] += x[q][u];
                }
        }
#pragma unroll
        for (int q = 0; q < OB; ++q)
            if (o0 + q < n_out) slab[lay.off_w[y][3] + (size_t)(o0 + q) * HID + j] = acc[q];
    }
    if (j < n_out) {
        float sb = 0.f, sl = 0.f;
        for (int t0 = 0; t0 < cnt; t0 += HB) {
            float xb[HB], xl[HB];
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const float* part = base + (size_t)min(t0 + u, cnt - 1) * pstride;
                xb[u] = part[n_out * HID + j];
                xl[u] = part[n_out * HID + n_out + j];
            }
#pragma unroll
            for (int u = 0; u < HB; ++u)
                if (t0 + u < cnt) {
                    sb += xb[u];
                    sl += xl[u];
                }
        }
        if (lay.off_b[y][3] >= 0) slab[lay.off_b[y][3] + j] = sb;
        if (y == 1) slab[lay.off_logstd + j] = sl;
    }
}
#endif


// ---- host side ---------------------------------------------------------------------------------------------------------
inline int check_layout(const rlx_mlp_layout* lay, const char* who) {
    RLX_REQUIRE(lay != nullptr, "%s: NULL layout", who);
    RLX_REQUIRE(lay->hidden == HID, "%s: hidden=%d is not supported (the reference's MLP policy is 256 wide)", who, lay->hidden);
    RLX_REQUIRE(lay->obs_dim >= 1 && lay->obs_dim <= Tiles::K1P, "%s: obs_dim=%d out of range [1,%d]", who, lay->obs_dim, Tiles::K1P);
    RLX_REQUIRE(lay->act_dim >= 1 && lay->act_dim <= MAX_OUT && lay->val_dim >= 1 && lay->val_dim <= MAX_OUT,
                "%s: act_dim=%d / val_dim=%d out of range [1,%d]", who, lay->act_dim, lay->val_dim, MAX_OUT);
    for (int y = 0; y < 2; ++y)
        for (int l = 0; l < 4; ++l) {
            RLX_REQUIRE(lay->off_w[y][l] >= 0 && lay->off_w[y][l] < lay->n_params, "%s: weight offset out of range", who);
            RLX_REQUIRE(lay->off_w[y][l] % 4 == 0 || l == 0, "%s: weight offsets of layers 2-4 must be 16-byte aligned", who);
            RLX_REQUIRE(l == 3 || lay->off_b[y][l] >= 0, "%s: hidden layers need a bias", who);
        }
    return RLX_OK;
}

// once per kernel and process (not a stream operation: keep it out of hipGraph capture regions)
template <typename K>
int set_lds(K kern, size_t bytes) {
    static thread_local const void* done[32] = {};
    const void* key = reinterpret_cast<const void*>(kern);
    for (const void* d : done)
        if (d == key) return RLX_OK;
    RLX_HIP_CHECK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    for (auto& d : done)
        if (d == nullptr) { d = key; break; }
    return RLX_OK;
}

// Development switches.  The PRODUCT build reads no kernel-variant environment variable at all: every switch returns its default --
// the measured-best path -- and the variant kernels behind the switches are not even compiled (the `#ifdef RLX_DEV_VARIANTS`
// blocks of the .hip files).  A development build (RLX_CXXFLAGS=-DRLX_DEV_VARIANTS RLX_BUILD_TAG=dev python -m rlinf_amd.csrc.build,
// loaded with RLX_LIB_TAG=dev) reads the variable at every call; rlx_dev_variants() tells a caller which build it loaded.
#ifdef RLX_DEV_VARIANTS
inline int dev_variant(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
inline int dev_variant(const char*, int dflt) { return dflt; }
#endif

inline int head_stride_of(const rlx_mlp_layout* lay) {
    const int n = std::max(lay->act_dim, lay->val_dim);
    return round_up(n * HID + 2 * n, 4);
}

struct StepPlan {
    int tiles;        // row tiles of the fused kernel (its grid)
    int loss_slots;   // metric partial slots per network (one per tile)
    int head_parts;   // head-gradient partial slots per network (one per 32 rows; the row-split bf16 launch: one per 64-row tile)
    int slabs, rows_per_slab, head_stride;
    size_t off_h, off_dz, off_head, off_loss, off_tiles, off_st, bytes;
};
constexpr int STEP_BM = 64;
// rows per workgroup of the bf16 fused launch: 32 (default: two workgroups per CU at 8192 rows -- their phases, matrix pipe /
// VALU epilogues / barrier and memory latency, overlap; 58 KB of LDS and 128 VGPRs each) or 64 (RLX_FUSED_RT=4: one per CU,
// half the weight-fragment traffic, 157 KB of LDS).  Measured (profiles/r02_fused_rows_per_workgroup.txt): 25.6 against 27.7 us.
// rows per workgroup of the bf16 rollout launch (every workgroup pulls its network's whole 0.29 MB tile image through the L2)
inline int rollout_bm_bf16() { return dev_variant("RLX_ROLLOUT_RT", 1) == 2 ? 32 : 16; }
inline int dw_nbuf() { return dev_variant("RLX_DW_NBUF", 3); }  // LDS ring depth of the bf16 weight-gradient launch (3, 4, 5, 6, 9)
inline int fused_bm_bf16() { return dev_variant("RLX_FUSED_RT", 2) == 4 ? 64 : 32; }
// The row-split bf16 launch (ppo_step_bf16_rows.hip: 64 rows per workgroup, one wave per 16 rows, weights through an LDS ring) for
// the shapes it covers (fused_rows_eligible), behind RLX_FUSED_ROWS=1 while it is being tuned.
inline bool fused_rows_bf16() { return dev_variant("RLX_FUSED_ROWS", 0) != 0; }
bool fused_rows_eligible(const rlx_mlp_layout& lay, const rlx_ppo_loss_params& p);
// precision "32": true (default) = the f32 launches run on the bf16 matrix pipe with three-plane operands (ppo_step_f32x.hip),
// false (RLX_F32_EXACT_MFMA=1, read once per process) = the exact-f32-MFMA launches of ppo_step.hip.  The f32 weight image
// (rlx_mlp_pack_tiles / the optimizer's tile refresh) follows the same switch.
bool f32_split();
size_t f32x_tiles_bytes();
// rows per workgroup of the f32x fused launch: 32 (the three-plane slab allows one workgroup per CU either way; 64 rows -- half the
// weight bytes per row, ONE round of 256 workgroups at 8192 rows -- measured the same 82 us: development builds only, RLX_F32X_RT=4)
inline int f32x_bm() { return dev_variant("RLX_F32X_RT", 2) == 4 ? 64 : 32; }
inline StepPlan plan_step(const rlx_mlp_layout* lay, int64_t m, bool bf16 = false, bool rows = false) {
    StepPlan pl{};
    const bool f32x = !bf16 && f32_split();
    const int bm = (bf16 && rows) ? 64 : bf16 ? fused_bm_bf16() : f32x ? f32x_bm() : STEP_BM;
    pl.tiles = ceil_div(m, bm);
    pl.head_parts = (bf16 && rows) ? pl.tiles : pl.tiles * (bm / 32);
    pl.loss_slots = pl.tiles;
    // 20 GEMM items per slab.  Every slab is 1.15 MB written here and read back by the slab reduce through the memory side:
    // (round 2, the LDS-DMA kernel with the copies in the compute waves: 16 slabs = 1.25 workgroups per CU was its best trade --
    // 16.2 us + slab reduce 9.1 us, against 15.8 + 10.4 us with 24 slabs)
    // (the exact-f32 launch is bound by the f32 matrix pipe, not by slab bytes: it keeps 2 workgroups per CU in one round --
    //  measured 71 us with 24 slabs against 97 us with 16)
    // bf16, round 4 (ring weight-gradient launch: loader waves, one 64 KiB workgroup per CU): every workgroup of the launch -- 20 GEMM
    // items + 2 head-reduce blocks per slab + 1 -- resident at once, NONE sharing a CU: 10 slabs on 256 CUs.  Measured on one box
    // (profiles/r04_dw_ring_slab_sweep.txt): 16 slabs 19.9 us (353 blocks: a CU with two of them sets the launch's duration),
    // 12: 21.8 (265 blocks), 11: 15.6, 10: 15.6, 9: 16.4, 8: 17.2 -- against 16.8 for the previous kernel at its best (16 slabs);
    // the slab reduce behind it 8.84 -> 8.16 us with 10.
    // f32 through bf16 planes: the loader-wave ring at every size (144 KiB of LDS: one workgroup per CU) -> the same 10-slab plan
    int want = bf16 ? (m >= 4096 ? std::max(1, (num_cu() - 1) / 25) : 5 * num_cu() / (4 * 20))   // (below 4096 rows: the previous kernel and its plan)
                    : f32x ? std::max(1, (num_cu() - 1) / 25) : 2 * num_cu() / 20;
    want = std::max(1, dev_variant("RLX_DW_SLABS", want));  // development: tools/bench_step.py sweeps it
    // ... and at least 256 rows (8 k-blocks) per slab: a data-parallel rank's small minibatch should not pay 16 slabs of traffic
    int slabs = std::max(1, std::min(want, ceil_div(m, 256)));
    pl.rows_per_slab = round_up(ceil_div(m, slabs), 32);
    pl.slabs = ceil_div(m, pl.rows_per_slab);
    pl.head_stride = head_stride_of(lay);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    if (bf16) {  // k-tiled transposed bf16 images: 16 column blocks x ceil(m / 32) row blocks x 1 KiB per [m][256] matrix
        const size_t img = (size_t)16 * ceil_div(m, 32) * 1024;
        pl.off_h = take(4 * img);
        pl.off_dz = take(6 * img);
        pl.off_st = take(img / 4);  // states: 4 column blocks
        pl.off_tiles = take(2 * Tiles::per_net() * 2);
    } else if (f32x) {  // three planes of every k-tiled image
        const size_t img = (size_t)16 * ceil_div(m, 32) * 1024;
        pl.off_h = take(3 * 4 * img);
        pl.off_dz = take(3 * 6 * img);
        pl.off_st = take(3 * img / 4);
        pl.off_tiles = take(f32x_tiles_bytes());
    } else {
        pl.off_h = take((size_t)4 * m * HID * sizeof(float));
        pl.off_dz = take((size_t)6 * m * HID * sizeof(float));
        pl.off_st = off;
        pl.off_tiles = take(2 * Tiles::per_net() * sizeof(float));
    }
    pl.off_head = take((size_t)pl.head_parts * 2 * pl.head_stride * sizeof(float));
    pl.off_loss = take((size_t)pl.loss_slots * 2 * NS * sizeof(double));
    pl.bytes = off;
    return pl;
}

extern long long* g_timing_buffer;  // development: phase stamps (rlx_dev_set_timing_buffer)

// bf16 launches (ppo_step_bf16.hip)
int pack_tiles_bf16(const float* params, const rlx_mlp_layout& lay, void* tiles, hipStream_t st);
int launch_rollout_bf16(const RolloutArgs& a, int blocks, hipStream_t st);
int launch_step_bf16(const StepArgs& a, const DwArgs& d, void* st_tiles, int tiles64, int dw_blocks, bool op8, bool rows, hipStream_t st);
int launch_fused_rows_bf16(const StepArgs& a, void* st_tiles, int tiles64, hipStream_t st);  // ppo_step_bf16_rows.hip
// f32 on the bf16 matrix pipe (ppo_step_f32x.hip)
int pack_tiles_f32x(const float* params, const rlx_mlp_layout& lay, void* tiles, hipStream_t st);
int launch_rollout_f32x(const RolloutArgs& a, int blocks, hipStream_t st);
int launch_step_f32x(const StepArgs& a, const DwArgs& d, void* st_tiles, int tiles32, int dw_blocks, hipStream_t st);

}  // namespace step
}  // namespace rlx
