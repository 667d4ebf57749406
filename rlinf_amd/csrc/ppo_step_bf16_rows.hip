// ppo_step_bf16_rows.hip -- the fused optimizer-step launch (1), bf16, ROW-SPLIT: one workgroup = 64 minibatch rows of one
// network, one wave = 16 of those rows through the whole chain (forward, head, loss, backward-data), gfx950.
//
// Replaces, like ppo_step_fused_bf16_kernel (ppo_step_bf16.hip), the training forward + PPO loss + backward of
//   MLPPolicy.default_forward            rlinf/models/embodiment/mlp_policy/mlp_policy.py:202-236
//   compute_ppo_actor_loss / critic_loss rlinf/algorithms/losses.py:170-380 (element math: ppo_loss_math.h)
//   EmbodiedFSDPActor.train_micro_batch  rlinf/workers/actor/embodied_fsdp_actor_worker.py:591-700
// and writes the same k-tiled bf16 images / head partials / metric partials for the weight-gradient launch behind it.
//
// Why a second decomposition.  The column-split kernel gives every wave 32 output columns of a 32-row tile: each layer ends in
// a workgroup barrier (the next layer needs everybody's columns), every wave pulls ITS weight fragments L2 -> registers, and a
// CU that hosts two such workgroups streams the network's 0.53 MB tile image twice.  Its phase stamps (profiles/
// r02_fused_phase_stamps_bf16_32row.txt) show a 47 k-cycle dependent chain of ~30 barriers for 544 cycles of matrix work per
// layer and wave -- and the launch still takes 18.6 us at one eighth of the rows (profiles/r03_per_rank_share_*.txt): the
// chain, not the throughput, is the bound.  Here:
//   * a wave owns 16 rows and ALL 256 columns: its activations go wave-private through its own LDS slab (accumulator layout
//     -> row-major bf16 -> A fragments), so NO workgroup barrier separates the layers;
//   * the weights are shared by the workgroup's four waves: the network's fragment tiles are streamed ONCE per workgroup,
//     global -> LDS by `global_load_lds_dwordx4` (no registers), one 16 KiB slot per k-step (16 column tiles x 1 KiB, already in
//     B-fragment lane order), through a ring of NSLOT slots that runs ahead across layer boundaries (34 slots: W1 2, W2 8,
//     W3 8, W3^T 8, W2^T 8); the only workgroup barriers are the ring hand-offs (counted vmcnt + raw s_barrier);
//   * one wave per SIMD: 16 MFMAs (16 independent accumulators) per k-step and wave, 512 VGPRs -- the rounded activations
//     h1..h3 stay in registers for the backward sweep, the next k-step's fragments are read while the current one multiplies.
// Same arithmetic as the column-split kernel element by element (same fragment tiles, same k order, same f32 head on three
// bf16 planes, same loss math): a sample's log-prob at rollout time and the first epoch's recomputation stay bit-identical.
// The head-gradient partial covers the workgroup's 64 rows (the column-split kernel: 32), the metric partials one wave each.
//
// All ordinary global loads happen before the ring starts (hipcc waits vmcnt(0) at the first use of an ordinary load while
// an LDS-DMA is in flight: cdna_hip_programming.md, "Pipelining across barriers"); behind that point the waves only store.
// The counted waits assume nothing about those stores: `vmcnt(4 x slots issued behind the awaited one)` is conservative when
// stores sit between the DMAs (a wave's memory operations retire in order: the awaited slot is older than the N youngest).

#include <type_traits>

#include "ppo_step_bf16_parts.h"

namespace rlx {
namespace {

using namespace loss;
using namespace step;
using namespace b16;

constexpr int RW_NW = 4, RW_NT = 64 * RW_NW, RW_BM = 16 * RW_NW;  // compute waves
constexpr int RW_NL = 2, RW_THREADS = RW_NT + 64 * RW_NL;           // + loader waves
constexpr int RW_SLOT = 16 * 1024;            // one k-step of a layer: 16 column tiles x 1 KiB
constexpr int RW_STEPS = 2 + 8 + 8 + 8 + 8;   // W1 | W2 | W3 | W3^T | W2^T
constexpr int RW_C_L1 = 0, RW_C_L2 = 2, RW_C_L3 = 10, RW_C_B3 = 18, RW_C_B2 = 26;

__host__ __device__ constexpr int rw_mat(int s) { return s < 2 ? 0 : s < 10 ? 1 : s < 18 ? 2 : s < 26 ? 4 : 3; }
__host__ __device__ constexpr int rw_ks(int s) { return s < 2 ? s : (s - 2) & 7; }

// LDS map (bytes): ring | 4 wave slabs [16][XSB] bf16 | hidden biases [3][HID] f32 | per wave 3 x [16][MAX_OUT] f32 scratch
// | head image [W4R][W4S] + bias [MAX_OUT] + std / var / log std [3][MAX_OUT] (+ pad) | reduction scratch
template <int NSLOT, int W4R>
struct RowsLds {
    static constexpr size_t SLAB_BYTES = (size_t)16 * XSB * sizeof(__bf16);
    static constexpr size_t SCR_BYTES = (size_t)3 * 16 * MAX_OUT * sizeof(float);
    static constexpr size_t RING = 0;
    static constexpr size_t SLAB = RING + (size_t)NSLOT * RW_SLOT;
    static constexpr size_t BIAS = SLAB + RW_NW * SLAB_BYTES;
    static constexpr size_t SCR = BIAS + (size_t)3 * HID * sizeof(float);
    static constexpr size_t W4 = SCR + RW_NW * SCR_BYTES;
    static constexpr size_t RED = W4 + (size_t)(W4R * W4S + MAX_OUT + 4 * MAX_OUT) * sizeof(float);
    static constexpr size_t BYTES = RED + 1024;
    static_assert(BYTES <= 160 * 1024, "LDS budget of a CU");
    static_assert(SLAB_BYTES >= (size_t)16 * 64 * sizeof(bf16x4), "the head-gradient exchange parks 16 tiles x 64 lanes x 8 B in a slab");
};

// lgkmcnt(0) as the BUILTIN (simm16 0xC07F: vmcnt / expcnt left at their maxima), not inline asm: hipcc's wait-count pass then
// knows that every LDS read issued so far has landed.  With an opaque asm wait it protected the MFMAs of k-step ks (operands read
// one k-step earlier) with its own lgkmcnt -- and because the 17 reads of k-step ks + 1 already in flight exceed the 4-bit
// counter, that wait came out as lgkmcnt(0): every other k-step waited for the fragments it had just requested.
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void ring_barrier() {
    wave_lds_fence();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The weight ring, consumer side: slot C is complete for every wave once everybody passed hand-off C (the loader waves arrive
// only after their copies of slot C have landed); the slot behind it (C - 1) is then free and the loaders refill it.
template <int NSLOT>
struct WeightRing {
    char* ring;
    template <int C>
    __device__ __forceinline__ void acquire() const {
        ring_barrier();
    }
};

// Loader waves (RW_NL of them, nothing else to do): wave lw copies column tiles 8 lw .. 8 lw + 7 of every slot, global -> LDS.
// A compute wave that issues its own LDS-DMAs stalls ~100-185 cycles per instruction while the LDS serves fragment reads
// (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"; round-4 stamps of the first version of this kernel: 680 cycles per k-step
// with four DMAs per compute wave and k-step against 272 cycles of MFMA time) -- a wave that ONLY copies issues a slot in a few
// hundred cycles and its memory queue holds nothing but copies, so its counted waits are exact.
template <int N>
__device__ __forceinline__ void wait_slots_behind(int behind) {  // vmcnt(8 x behind): s_waitcnt takes an immediate
    if constexpr (N > 0) {
        if (behind >= N) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * N) : "memory");
            return;
        }
        wait_slots_behind<N - 1>(behind);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
template <int NSLOT>
__device__ __forceinline__ void loader_main(const __bf16* __restrict__ tiles_y, char* ring, int lw, int lane) {
    constexpr int AHEAD = NSLOT - 1;
    static_assert(8 * (AHEAD - 1) < 64, "vmcnt is a 6-bit counter");
    const __bf16* src = tiles_y + lane * 8;
    auto issue = [&](int s, int buf) {
        int m = 0, ks = s;
        if (s >= 2) {
            const int t = s - 2, l = t >> 3;
            ks = t & 7;
            m = l == 0 ? 1 : l == 1 ? 2 : l == 2 ? 4 : 3;  // W2, W3, W3^T, W2^T
        }
        const __bf16* g = src + (m == 0 ? (size_t)0 : (size_t)HID * Tiles::K1P + (size_t)(m - 1) * HID * HID) + (size_t)ks * 16 * 512;
        char* dst = ring + buf * RW_SLOT;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int nt = 8 * lw + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)nt * 512),
                                             (__attribute__((address_space(3))) void*)(dst + nt * 1024), 16, 0, 0);
        }
    };
    int fill = 0;  // buffer of the next slot to issue
    for (int s = 0; s < AHEAD; ++s) {
        issue(s, fill);
        fill = fill + 1 == NSLOT ? 0 : fill + 1;
    }
    for (int c = 0; c < RW_STEPS; ++c) {
        wait_slots_behind<AHEAD - 1>(min(AHEAD - 1, RW_STEPS - 1 - c));
        __builtin_amdgcn_s_barrier();  // hand-off c
        asm volatile("" ::: "memory");
        if (c + AHEAD < RW_STEPS) {
            issue(c + AHEAD, fill);
            fill = fill + 1 == NSLOT ? 0 : fill + 1;
        }
        if (c == RW_C_B3 - 1) {  // the compute waves' two exchange barriers between the forward and the backward sweep
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
    }
}

template <int NSLOT, int C, int G>
__device__ __forceinline__ void read_frag_group(const char* ring_lane, const __bf16* xa, int ks, bf16x8& a, bf16x8 (&b)[16]) {
    const char* slot = ring_lane + (C % NSLOT) * RW_SLOT;
    if constexpr (G == 0) a = *reinterpret_cast<const bf16x8*>(xa + ks * 32);
#pragma unroll
    for (int nt = 4 * G; nt < 4 * G + 4; ++nt) b[nt] = *reinterpret_cast<const bf16x8*>(slot + nt * 1024);
}

// acc[nt] = X[16 rows][0 : 32 NKS] . W^T for the 16 column tiles; X = this wave's slab, W = ring slots C0 .. C0 + NKS - 1.
// Software-pipelined by one k-step, in four groups: 4 (+1) fragment reads of k-step ks + 1, then 4 MFMAs of k-step ks.  A wave
// can have at most 15 LDS operations outstanding (lgkmcnt is 4 bits): with all 17 reads of the next k-step issued in one block
// the issue stalled until most of them had returned and the matrix pipe sat idle meanwhile (round-4 stamps: 680 cycles per
// k-step against 272 of MFMA time); left to itself hipcc reads each fragment right before its MFMA and waits lgkmcnt(0) in
// between (an LDS round trip per instruction).
template <int NSLOT, int C0, int NKS, int KS>
struct GemmSteps {
    static __device__ __forceinline__ void run(const WeightRing<NSLOT>& wr, const char* ring_lane, const __bf16* xa, f32x4 (&acc)[16],
                                               bf16x8 (&a)[2], bf16x8 (&b)[2][16]) {
        constexpr bool more = KS + 1 < NKS;
        if constexpr (more) wr.template acquire<C0 + KS + 1>();
#define RLX_ROWS_GROUP(G)                                                                                                                      \
        if constexpr (more) read_frag_group<NSLOT, C0 + KS + 1, G>(ring_lane, xa, KS + 1, a[(KS + 1) & 1], b[(KS + 1) & 1]);              \
        __builtin_amdgcn_sched_barrier(0);                                                                                                 \
        _Pragma("unroll") for (int nt = 4 * G; nt < 4 * G + 4; ++nt)                                                                       \
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[KS & 1], b[KS & 1][nt], acc[nt], 0, 0, 0);                                \
        __builtin_amdgcn_sched_barrier(0);
        RLX_ROWS_GROUP(0)
        RLX_ROWS_GROUP(1)
        RLX_ROWS_GROUP(2)
        RLX_ROWS_GROUP(3)
#undef RLX_ROWS_GROUP
        if constexpr (more) GemmSteps<NSLOT, C0, NKS, KS + 1>::run(wr, ring_lane, xa, acc, a, b);
    }
};
template <int NSLOT, int C0, int NKS>
__device__ __forceinline__ void row_gemm(const WeightRing<NSLOT>& wr, const __bf16* Xb, f32x4 (&acc)[16]) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    const char* ring_lane = wr.ring + lane * 16;
    const __bf16* xa = Xb + r16 * XSB + 8 * kq;
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2], b[2][16];
    wr.template acquire<C0>();
    read_frag_group<NSLOT, C0, 0>(ring_lane, xa, 0, a[0], b[0]);
    read_frag_group<NSLOT, C0, 1>(ring_lane, xa, 0, a[0], b[0]);
    read_frag_group<NSLOT, C0, 2>(ring_lane, xa, 0, a[0], b[0]);
    read_frag_group<NSLOT, C0, 3>(ring_lane, xa, 0, a[0], b[0]);
    GemmSteps<NSLOT, C0, NKS, 0>::run(wr, ring_lane, xa, acc, a, b);
}

// this wave's 16 x 16 tiles in the accumulator layout -> the k-tiled transposed image (weight-gradient operand), 8 B per lane
__device__ __forceinline__ void store_image(const bf16x4 (&v)[16], __bf16* __restrict__ img, int nrb, int rb, int half) {
    if (rb >= nrb) return;  // wave-uniform
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    const int kblk = 2 * half + (kq >> 1);
#pragma unroll
    for (int nt = 0; nt < 16; ++nt)
        *reinterpret_cast<bf16x4*>(img + ((size_t)(nt * nrb + rb) * 64 + kblk * 16 + r16) * 8 + 4 * (kq & 1)) = v[nt];
}

template <int NSLOT, int W4R, bool STAMPS>
__global__ __launch_bounds__(RW_THREADS) void ppo_step_fused_bf16_rows_kernel(StepArgs a, __bf16* st_tiles) {
    typedef RowsLds<NSLOT, W4R> L;
    extern __shared__ __align__(16) char lds[];
    const rlx_mlp_layout& lay = a.lay;
    const rlx_ppo_loss_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x, y = blockIdx.y, D = lay.obs_dim;
    const long long M = a.M, mw = (long long)tile * RW_BM + 16 * wave;  // this wave's first row
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const long long n_adv = M * (lay.act_dim / p.raw_per_adv);
    const bool has_mask = a.loss_mask != nullptr, has_msum = a.loss_mask_sum != nullptr;
    const bool ratio_mode = p.max_episode_steps > 0 && has_mask && has_msum;
    const TileGeom tg{(int)((M + 31) / 32)};
    const int rb = (int)(mw >> 5), half = wave & 1;

    __bf16* Xb = reinterpret_cast<__bf16*>(lds + L::SLAB + wave * L::SLAB_BYTES);
    float* sBias = reinterpret_cast<float*>(lds + L::BIAS);
    float* sP0 = reinterpret_cast<float*>(lds + L::SCR + wave * L::SCR_BYTES);  // head partial 0, later d(loss)/d(log std) rows
    float* sP1 = sP0 + 16 * MAX_OUT;                                            // head partial 1
    float* sH = sP1 + 16 * MAX_OUT;                                             // dOut rows (zero outside [16][n_out])
    float* W4s = reinterpret_cast<float*>(lds + L::W4);
    float* b4s = W4s + W4R * W4S;
    float* sStd = b4s + MAX_OUT;
    double* sRed = reinterpret_cast<double*>(lds + L::RED);
    const __bf16* tiles = reinterpret_cast<const __bf16*>(a.tiles);
    __bf16* hy = reinterpret_cast<__bf16*>(a.h) + (size_t)(y * 2) * tg.mat();
    __bf16* dzy = reinterpret_cast<__bf16*>(a.dz) + (size_t)(y * 3) * tg.mat();

    StampsT<STAMPS> ts{a.stamps, 0};
    ts.mark();
    double nm = 0.0;
    if (has_mask) {  // loss_mask.sum() of the whole micro-batch (every workgroup counts it: rows stay independent)
        double cnt[1] = {0.0};
        for (long long e = tid; e < n_adv; e += RW_THREADS) cnt[0] += a.loss_mask[e] != 0 ? 1.0 : 0.0;
        block_sum<1>(cnt, sRed);
        if (tid == 0) sRed[8] = cnt[0];
        __syncthreads();
        nm = sRed[8];
    }
    if (wave >= RW_NW) {
        loader_main<NSLOT>(tiles + Tiles::mat(y, 0), lds + L::RING, wave - RW_NW, lane);
        return;
    }

    // ---- every ordinary global load of the launch, then the first ring slots -------------------------------------------
    float xs[16];  // this wave's states: row u, (padded) column `lane` -- 16 rows x 64 columns in one pass
    static_assert(Tiles::K1P == 64, "one column per lane assumes a 64-wide padded first layer");
#pragma unroll
    for (int u = 0; u < 16; ++u) xs[u] = a.states[(size_t)min(mw + u, M - 1) * D + min(lane, D - 1)];  // clamped, unconditional
    // loss inputs of element e = lane + 64 u of this wave's [16][n_out] block (row = e / n_out, o = e % n_out)
    // (n_out is a power of two: fused_rows_eligible; NPASS = W4R / 4 covers 16 x n_out <= 64 NPASS elements)
    constexpr int NPASS = W4R / 4;
    const int epw = 16 * n_out, npass = (epw + 63) >> 6, osh = 31 - __builtin_clz(n_out);
    float v_a[NPASS], v_b[NPASS], v_c[NPASS];  // policy: old log-prob, action, advantage;  critic: prev value, return, -
    unsigned m_on[NPASS];
    long long m_sum[NPASS];
    const float* safe = a.params;
    const int64_t* safe64 = reinterpret_cast<const int64_t*>(a.params);
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
        const int e = min(lane + 64 * u, epw - 1), row = e >> osh, o = e & (n_out - 1);
        const size_t gr = (size_t)min(mw + row, M - 1);
        const size_t ge = y == 1 ? gr : gr * n_out + o;  // loss element index: one per row (action_level, one sub-group) / per value output
        if (y == 1) {
            v_a[u] = a.old_logprobs[gr * lay.act_dim + o];
            v_b[u] = a.action[gr * lay.act_dim + o];
            v_c[u] = a.advantages[gr];
        } else {
            v_a[u] = (p.has_critic ? a.prev_values : safe)[p.has_critic ? ge : 0];
            v_b[u] = (p.has_critic ? a.returns : safe)[p.has_critic ? ge : 0];
            v_c[u] = 0.f;
        }
        m_on[u] = (has_mask ? a.loss_mask : reinterpret_cast<const uint8_t*>(safe))[has_mask ? ge : 0];
        m_sum[u] = (has_msum ? a.loss_mask_sum : safe64)[has_msum ? ge : 0];
    }
    SmallInputsB<RW_NT> si;
    si.issue(a.params, lay, y, n_out);
    __builtin_amdgcn_sched_barrier(0);
    WeightRing<NSLOT> wr{lds + L::RING};
#pragma unroll
    for (int u = 0; u < 16; ++u) Xb[u * XSB + lane] = (__bf16)((lane < D && mw + u < M) ? xs[u] : 0.f);  // all 64 columns: the k tail is zero
    si.commit(n_out, sBias, W4s, b4s, sStd);
    for (int i = n_out * W4S + tid; i < W4R * W4S; i += RW_NT) W4s[i] = 0.f;
    for (int i = lane; i < 16 * MAX_OUT; i += 64) sH[i] = 0.f;
    wave_lds_fence();
    ts.mark();
    if (y == 1) {  // the states' k-tiled image (B operand of the first layers' weight gradients): 4 column blocks x this wave's two 8-row groups
        for (int u = lane; u < 4 * 2 * 16; u += 64) {
            const int cb = u >> 5, ko = (u >> 4) & 1, c16 = u & 15;
            if (rb >= tg.nrb) continue;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = Xb[(8 * ko + j) * XSB + cb * 16 + c16];
            *reinterpret_cast<bf16x8*>(st_tiles + ((size_t)(cb * tg.nrb + rb) * 64 + (2 * half + ko) * 16 + c16) * 8) = v;
        }
    }

    // ---- forward -----------------------------------------------------------------------------------------------------
    f32x4 acc[16];
    // The rounded activations in the accumulator layout (1 - h^2 of the backward sweep): h2, h3 stay in registers; h1 would pin 32
    // more through the whole kernel (6 waves per CU leave 256 per wave) -- every lane reads back the 8-byte words it stored into
    // the h1 image, requested before the last GEMM and used behind it.
    bf16x4 kept[2][16], h1v[16];
    auto epilogue_tanh = [&](auto lc, bf16x4 (&keep)[16]) {
        constexpr int l = decltype(lc)::value;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const float bv = sBias[l * HID + nt * 16 + r16];
            bf16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 h = tanh2_b(f32x2{acc[nt][r] + bv, acc[nt][r + 1] + bv});
                hv[r] = (__bf16)h.x;
                hv[r + 1] = (__bf16)h.y;
            }
            store_slab_quad(Xb, 0, nt * 16, hv);
            keep[nt] = hv;
        }
    };
    row_gemm<NSLOT, RW_C_L1, Tiles::K1P / 32>(wr, Xb, acc);
    ts.mark();
    epilogue_tanh(std::integral_constant<int, 0>{}, h1v);
    store_image(h1v, hy, tg.nrb, rb, half);
    ts.mark();
    row_gemm<NSLOT, RW_C_L2, HID / 32>(wr, Xb, acc);
    ts.mark();
    epilogue_tanh(std::integral_constant<int, 1>{}, kept[0]);
    store_image(kept[0], hy + tg.mat(), tg.nrb, rb, half);
    ts.mark();
    row_gemm<NSLOT, RW_C_L3, HID / 32>(wr, Xb, acc);
    ts.mark();
    epilogue_tanh(std::integral_constant<int, 2>{}, kept[1]);
    wave_lds_fence();
    ts.mark();

    // ---- head (f32 weights as three bf16 planes) + loss element math ------------------------------------------------------
    // Two k-half partials added afterwards, each over four k-steps in ascending order: the rollout launch's order.
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < HID / 64; ++q) {
            const int k0 = (kh * (HID / 64) + q) * 32 + 8 * kq;
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(Xb + r16 * XSB + k0);
            const int wr_row = W4R == MAX_OUT ? r16 : min(r16, W4R - 1);
            f32x4 w0 = *reinterpret_cast<const f32x4*>(W4s + wr_row * W4S + k0), w1 = *reinterpret_cast<const f32x4*>(W4s + wr_row * W4S + k0 + 4);
            if (W4R != MAX_OUT && r16 >= W4R) w0 = w1 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float w[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            hacc = mfma3(av, split3(w), hacc);
        }
        float* P = kh == 0 ? sP0 : sP1;
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(4 * kq + r) * MAX_OUT + r16] = hacc[r];
    }
    wave_lds_fence();
    ts.mark();

    const Denoms den = denominators(p, n_adv, nm, has_mask, has_msum);
    const float half_delta = (float)(0.5 * (double)p.huber_delta);
    double lacc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) lacc[k] = 0.0;
    const bool has_b4 = lay.off_b[y][3] >= 0;
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
        if (u >= npass) break;  // wave-uniform
        const int e = lane + 64 * u;
        const bool mine = e < epw;
        const int row = mine ? e >> osh : 0, o = mine ? e & (n_out - 1) : 0;
        const bool valid = mine && mw + row < M;
        float sv = fadd(sP0[row * MAX_OUT + o], sP1[row * MAX_OUT + o]);
        if (has_b4) sv = fadd(sv, b4s[o]);
        const bool on = has_mask ? m_on[u] != 0 : true;
        float w = 1.f;
        if (ratio_mode) w = ((float)m_sum[u] * 1.0f) / (float)p.max_episode_steps;
        if (y == 1) {
            // A row's n_out lanes are neighbours (64 % n_out == 0: a row never straddles a pass): the row leader (o == 0) collects
            // the per-dimension log-probs with n_out - 1 lane shifts, adds them in the reference's order (ascending, from 0.f),
            // evaluates the loss element and hands d(loss)/d(logprob) back to its lanes.
            const float d = fsub(v_b[u], sv);
            const float var = sStd[MAX_OUT + o], log_scale = sStd[2 * MAX_OUT + o];
            const float lpe = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
            const float olde = v_a[u];
            float lp = fadd(0.f, lpe), old = fadd(0.f, olde);
            for (int j = 1; j < n_out; ++j) {
                lp = fadd(lp, __shfl_down(lpe, j, 64));
                old = fadd(old, __shfl_down(olde, j, 64));
            }
            float gs = 0.f;
            if (valid && o == 0) {
                lacc[S_NM] += on ? 1.0 : 0.0;
                const float g = actor_elem(p, lp, old, v_c[u], on, w, ratio_mode, lacc);
                gs = (a.grad_out * (float)(1.0 / den.actor)) * g;
            }
            gs = __shfl(gs, (lane - o) & 63, 64);
            float dmu = 0.f, dls = 0.f;
            if (valid) {
                dmu = gs * d / var;
                dls = gs * (d * d / var - 1.f);
            }
            if (mine) {
                sH[row * MAX_OUT + o] = dmu;
                sP0[row * MAX_OUT + o] = dls;  // (this lane read its sP0 word above)
            }
        } else if (mine) {
            float gv = 0.f;
            if (valid && p.has_critic)
                gv = (a.grad_out * (float)(1.0 / den.critic)) * critic_elem(p, sv, v_a[u], v_b[u], on, w, ratio_mode, half_delta, lacc);
            sH[row * MAX_OUT + o] = gv;
        }
    }
    {   // Metric partials of this wave's 16 rows.  As 64-lane butterflies of doubles this was the longest phase of the kernel
        // (round-4 stamps: ~10 k of 78 k cycles -- every step of every sum is a dependent ds_bpermute round trip).  Instead the
        // lanes park their sums lane-major in the (dead) slab, lane (k, q) = (slot, quarter) adds 16 of them in ascending lane
        // order, two shuffle steps join the quarters: a fixed tree again, one LDS round trip.
        constexpr int NSP = NS + 1;  // padded row (doubles): 2-way bank conflicts at most
        static_assert((size_t)64 * NSP * sizeof(double) <= L::SLAB_BYTES, "the metric rows are parked in the slab");
        double* sL = reinterpret_cast<double*>(Xb);
#pragma unroll
        for (int k = 0; k < NS; ++k) sL[lane * NSP + k] = lacc[k];
        wave_lds_fence();
        const int k = lane >> 2, q = lane & 3;
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += sL[(q * 16 + i) * NSP + k];
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        double* lp = a.loss_part + ((size_t)(tile * RW_NW + wave) * 2 + y) * NS;
        if (q == 0) lp[k] = sum;  // (the other network's slots are exact zeros)
        wave_lds_fence();
    }
    // ---- head parameter gradients of the workgroup's 64 rows -----------------------------------------------------------------
    // dW4[o][j] = sum_rows dOut[row][o] h3[row][j] on the matrix pipe: M = o, N = j, K = rows.  Every wave parks its h3 tiles
    // (accumulator layout: lane (r16, kq) holds rows 4 kq .. 4 kq + 3 of column nt * 16 + r16) lane-linear in its -- now dead --
    // slab; wave w then takes column tiles 4 w .. 4 w + 3 over all 64 rows: two k-steps of 32 rows, k slot jj <-> row
    // (jj >> 2) * 16 + 4 kq + (jj & 3) of the wave pair, dOut split into three bf16 planes (exact, see split3).
    {
        bf16x4* park = reinterpret_cast<bf16x4*>(Xb);
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) park[nt * 64 + lane] = kept[1][nt];
    }
    ring_barrier();
    ts.mark();
    {
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.f;
        f32x4 g[4] = {zero4, zero4, zero4, zero4}, gb = zero4, gl = zero4;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float av[8], lv[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float* scr = reinterpret_cast<const float*>(lds + L::SCR + (2 * pr + (jj >> 2)) * L::SCR_BYTES);
                const int row = 4 * kq + (jj & 3);
                av[jj] = scr[2 * 16 * MAX_OUT + row * MAX_OUT + r16];  // sH of that wave
                lv[jj] = scr[row * MAX_OUT + r16];                     // its d(loss)/d(log std) rows
            }
            const Split3 A = split3(av);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bf16x4 lo = reinterpret_cast<const bf16x4*>(lds + L::SLAB + (2 * pr) * L::SLAB_BYTES)[(4 * wave + c) * 64 + lane];
                const bf16x4 hi = reinterpret_cast<const bf16x4*>(lds + L::SLAB + (2 * pr + 1) * L::SLAB_BYTES)[(4 * wave + c) * 64 + lane];
                const bf16x8 B = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                g[c] = mfma3(A, B, g[c]);
            }
            if (wave == 0) {  // bias / log-std gradients: row sums = the same product against a fragment of ones
                gb = mfma3(A, ones, gb);
                if (y == 1) gl = mfma3(split3(lv), ones, gl);
            }
        }
        float* part = a.head_part + ((size_t)tile * 2 + y) * a.head_stride;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = (4 * wave + c) * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kq + r < n_out) part[(4 * kq + r) * HID + j] = g[c][r];
        }
        if (wave == 0 && r16 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * kq + r < n_out) {
                    part[n_out * HID + 4 * kq + r] = gb[r];
                    part[n_out * HID + n_out + 4 * kq + r] = gl[r];
                }
        }
    }
    ts.mark();
    // ---- dZ3 = (dOut . W4) * (1 - h3^2): K = n_out is tiny, f32 operands -> v_mfma_f32_16x16x4_f32 -----------------------------
    bf16x4 dv[16];
    {
        f32x4 dz[16];
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) dz[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < (n_out + 3) / 4; ++ks) {
            const float av = sH[r16 * MAX_OUT + 4 * ks + kq];
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const float bv = W4s[(4 * ks + kq) * W4S + nt * 16 + r16];
                dz[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, dz[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 16; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[nt][r] = (__bf16)(dz[nt][r] * dtanh_b(kept[1][nt][r]));
    }
    ring_barrier();  // every wave has read the parked h3 tiles and the dOut rows: the slabs may be overwritten
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) store_slab_quad(Xb, 0, nt * 16, dv[nt]);
    store_image(dv, dzy + 2 * tg.mat(), tg.nrb, rb, half);
    ts.mark();

    // ---- backward-data chain -----------------------------------------------------------------------------------------------
    row_gemm<NSLOT, RW_C_B3, HID / 32>(wr, Xb, acc);  // dH2 = dZ3 . W3
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dv[nt][r] = (__bf16)(acc[nt][r] * dtanh_b(kept[0][nt][r]));
        store_slab_quad(Xb, 0, nt * 16, dv[nt]);
    }
    store_image(dv, dzy + tg.mat(), tg.nrb, rb, half);
    {   // h1 back from its image (clamped row block: a wave past the last block reads finite junk it never stores)
        const int rbc = min(rb, tg.nrb - 1), kblk = 2 * half + (kq >> 1);
#pragma unroll
        for (int nt = 0; nt < 16; ++nt)
            h1v[nt] = *reinterpret_cast<const bf16x4*>(hy + ((size_t)(nt * tg.nrb + rbc) * 64 + kblk * 16 + r16) * 8 + 4 * (kq & 1));
    }
    ts.mark();
    row_gemm<NSLOT, RW_C_B2, HID / 32>(wr, Xb, acc);  // dH1 = dZ2 . W2
#pragma unroll
    for (int nt = 0; nt < 16; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dv[nt][r] = (__bf16)(acc[nt][r] * dtanh_b(h1v[nt][r]));
    store_image(dv, dzy, tg.nrb, rb, half);
    ts.mark();
    if constexpr (STAMPS) {
        if (a.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 1 && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            a.stamps[ts.n] = (long long)clock64();
        }
    }
}

template <typename K>
int set_lds_rows(K kern, size_t bytes) {
    static thread_local const void* done[8] = {};
    const void* key = reinterpret_cast<const void*>(kern);
    for (const void* d : done)
        if (d == key) return RLX_OK;
    RLX_HIP_CHECK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    for (auto& d : done)
        if (d == nullptr) { d = key; break; }
    return RLX_OK;
}

}  // namespace

namespace step {

// The row-split launch covers the embodied shapes: one loss element per row (action_level log-probs, one sub-group) with a
// power-of-two action width (a row's lanes are neighbours inside one 64-lane pass) and 1, 2 or 4 value outputs.
bool fused_rows_eligible(const rlx_mlp_layout& lay, const rlx_ppo_loss_params& p) {
    const int npr = lay.act_dim / p.raw_per_adv;
    const bool val_pow2 = lay.val_dim == 1 || lay.val_dim == 2 || lay.val_dim == 4;
    return npr == 1 && p.sub_per_adv == 1 && lay.act_dim <= MAX_OUT && 64 % lay.act_dim == 0 && val_pow2;
}

int launch_fused_rows_bf16(const StepArgs& a, void* st_tiles, int tiles64, hipStream_t st) {
    __bf16* stt = static_cast<__bf16*>(st_tiles);
    const bool op8 = a.lay.act_dim <= 8 && a.lay.val_dim <= 8;
#define RLX_ROWS_LAUNCH(NSLOT, W4R, ST)                                                                                              \
    {                                                                                                                                \
        const size_t bytes = RowsLds<NSLOT, W4R>::BYTES;                                                                             \
        if (int rc = set_lds_rows(ppo_step_fused_bf16_rows_kernel<NSLOT, W4R, ST>, bytes)) return rc;                                \
        hipLaunchKernelGGL((ppo_step_fused_bf16_rows_kernel<NSLOT, W4R, ST>), dim3(tiles64, 2), dim3(RW_THREADS), bytes, st, a, stt);     \
    }
    const int nslot = dev_variant("RLX_ROWS_NSLOT", op8 ? 6 : 5);
    if (a.stamps != nullptr) {
        if (op8) RLX_ROWS_LAUNCH(6, 8, true) else RLX_ROWS_LAUNCH(5, 16, true)
    } else if (op8 && nslot == 6) RLX_ROWS_LAUNCH(6, 8, false)
    else if (op8 && nslot == 4) RLX_ROWS_LAUNCH(4, 8, false)
    else if (op8 && nslot == 3) RLX_ROWS_LAUNCH(3, 8, false)
    else RLX_ROWS_LAUNCH(5, 16, false)
#undef RLX_ROWS_LAUNCH
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // namespace step
}  // namespace rlx
