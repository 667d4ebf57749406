// ppo_step_bf16_rows.hip -- the fused optimizer-step launch (1), bf16, ROW-SPLIT form (development builds only: -DRLX_DEV_VARIANTS, RLX_FUSED_ROWS=1): one workgroup =
// 64 minibatch rows of one network; its 8 compute waves are 2 row pairs x 4 column quarters -- wave (rp, cq) owns 32 rows x 64
// columns (acc[2][4] tiles of 16 x 16) through the whole chain (forward, head, loss, backward-data); the network's weight tiles are
// staged ONCE per workgroup through an LDS ring by two loader waves.
//
// Replaces, like ppo_step_fused_bf16_kernel (ppo_step_bf16.hip), the training forward + PPO loss + backward of
//   MLPPolicy.default_forward            rlinf/models/embodiment/mlp_policy/mlp_policy.py:202-236
//   compute_ppo_actor_loss / critic_loss rlinf/algorithms/losses.py:170-380 (element math: ppo_loss_math.h)
//   EmbodiedFSDPActor.train_micro_batch  rlinf/workers/actor/embodied_fsdp_actor_worker.py:591-700
// and writes the same k-tiled bf16 images / head partials / metric partials for the weight-gradient launch behind it.
//
// Why a second decomposition.  The column-split kernel gives every wave 32 output columns of a 32-row tile and streams ITS weight
// fragments L2 -> registers: a CU that hosts two such workgroups pulls the network's 0.53 MB tile image twice, and a k-step is 4
// MFMAs per wave between two LDS round trips.  Here:
//   * the weights are shared by the workgroup: fragment tiles travel global -> LDS by `global_load_lds_dwordx4` (no registers),
//     one 32 KiB slot per TWO k-steps (2 x 16 column tiles x 1 KiB, contiguous in the k-step-major image, already in B-fragment
//     lane order), through a ring of NSLOT slots (3, or 2 beside the 16-output head image) that runs ahead across layer boundaries:
//     17 ring steps -- W1 1, W2 4, W3 4, W3^T 4, W2^T 4 (RW_C_*);
//   * RLX_ROWS_NL = 2 LOADER waves do nothing but copy (half of every slot's tiles each): a compute wave that issues its own
//     LDS-DMAs stalls ~100-185 cycles per instruction while the LDS serves fragment reads (MI355X_MICROARCH.md, "LDS-DMA piece issue
//     cost"; measured here: 680 cycles per k-step with the copies in the compute waves, 465 with loaders); a loader's memory queue
//     holds nothing but copies, so its counted vmcnt waits are exact, and the compute waves never wait on vmcnt in a GEMM sweep;
//   * per k-step a compute wave issues 8 MFMAs on 8 independent accumulators, the next k-step's fragments requested in groups of at
//     most 6 reads between 4 MFMAs (a wave can have 15 LDS operations outstanding: more in one block stalls the issue until they
//     return); the tile's activations go through LDS slabs (accumulator layout -> row-major bf16 -> A fragments); one ring hand-off
//     (builtin lgkmcnt(0) + s_barrier of all ten waves) per slot;
//   * two compute waves per SIMD: while one waits for an LDS round trip or a transcendental, the other issues (the first version of
//     this kernel ran ONE 16-row x 256-column wave per SIMD: its epilogues took 75 cycles per element);
//   * the loss is an 8-wave pass (every wave evaluates its rows' elements; DPP row reductions, no ds_bpermute), one metric partial
//     row and one head-gradient partial per WORKGROUP (column-split: per 32 rows).
// Same arithmetic as the column-split kernel element by element (same fragment tiles, same k order, same f32 head on three bf16
// planes added as two k-half partials, same loss math): a sample's log-prob at rollout time and the first epoch's recomputation
// stay bit-identical.
//
// Measured (profiles/r04_rows_kernel_*): 27.6 us at 8192 rows against 26.2 us for the column-split launch, 23.7 against 18.0 us at
// 1024 rows: the ring hand-off costs ~525 cycles per k-step where the matrix pipe needs 272.  Parity-green in every fused-step
// test (tests/test_gpu_fused_step.py, launch "rows"); not the default.  No decoupled-loss instantiation.

#ifdef RLX_DEV_VARIANTS  // measured slower than the default launch (see above): compiled into development builds only

#include <type_traits>

#ifndef RLX_ROWS_NL
#define RLX_ROWS_NL 2
#endif

#include "ppo_step_bf16_parts.h"

namespace rlx {
namespace {

using namespace loss;
using namespace step;
using namespace b16;

constexpr int RW_NG = 4, RW_BM = 16 * RW_NG;                      // row groups per workgroup, rows
constexpr int RW_NW = 2 * RW_NG, RW_NT = 64 * RW_NW;              // compute waves (a pair per row group), compute threads
constexpr int RW_NL = RLX_ROWS_NL, RW_THREADS = RW_NT + 64 * RW_NL;  // + loader waves (4 = one per SIMD: measured slower, 32.7 vs 27.6 us)
constexpr int RW_CT = 4;                                          // column tiles per compute wave (x 2 row groups)
constexpr int RW_KSTEP = 16 * 1024;                               // one k-step of a layer: 16 column tiles x 1 KiB
constexpr int RW_SLOT = 2 * RW_KSTEP;                             // a ring slot = two k-steps: one hand-off (barrier) per 16 MFMAs and wave
constexpr int RW_STEPS = 1 + 4 + 4 + 4 + 4;                       // slots: W1 | W2 | W3 | W3^T | W2^T
constexpr int RW_C_L1 = 0, RW_C_L2 = 1, RW_C_L3 = 5, RW_C_B3 = 9, RW_C_B2 = 13;
// Workgroup barriers besides the 17 ring hand-offs, by the hand-off they follow (the loader waves walk the same sequence):
//   behind the last slot of L1 / L2 / B3: 1 (everybody has read the slab: the epilogue may overwrite it)
//   behind the last slot of L3: 1 (same) + head: h3 slab complete, partials complete + head-gradient exchange: parked, consumed
__host__ __device__ constexpr int rw_extra_barriers(int c) { return c == RW_C_L2 - 1 || c == RW_C_L3 - 1 || c == RW_C_B2 - 1 ? 1 : c == RW_C_B3 - 1 ? 5 : 0; }

// LDS map (bytes): ring | 4 row-group slabs [16][XSB] bf16 | hidden biases [3][HID] f32 | per row group 3 x [16][MAX_OUT] f32
// | head image [W4R][W4S] + bias [MAX_OUT] + std / var / log std [3][MAX_OUT] (+ pad) | reduction scratch
template <int NSLOT, int W4R>
struct RowsLds {
    static constexpr size_t SLAB_BYTES = (size_t)16 * XSB * sizeof(__bf16);
    static constexpr size_t SCR_BYTES = (size_t)3 * 16 * MAX_OUT * sizeof(float);
    static constexpr size_t RING = 0;
    static constexpr size_t SLAB = RING + (size_t)NSLOT * RW_SLOT;
    static constexpr size_t BIAS = SLAB + RW_NG * SLAB_BYTES;
    static constexpr size_t SCR = BIAS + (size_t)3 * HID * sizeof(float);
    static constexpr size_t W4 = SCR + RW_NG * SCR_BYTES;
    static constexpr size_t RED = W4 + (size_t)(W4R * W4S + MAX_OUT + 4 * MAX_OUT) * sizeof(float);
    static constexpr size_t BYTES = RED + 1024;
    static_assert(BYTES <= 160 * 1024, "LDS budget of a CU");
    static_assert(SLAB_BYTES >= (size_t)16 * 64 * sizeof(bf16x4), "the head-gradient exchange parks 16 tiles x 64 lanes x 8 B in a slab");
};

__device__ int g_rows_dev_skip_reads = 0;  // development switch (see read_frags)

// lgkmcnt(0) as the BUILTIN (simm16 0xC07F: vmcnt / expcnt left at their maxima), not inline asm: hipcc's wait-count pass then
// knows that every LDS read issued so far has landed and does not protect the MFMAs of k-step ks (operands read one k-step
// earlier) with a lgkmcnt(0) of its own that would also wait for the fragments of k-step ks + 1 just requested.
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

// Loader waves: wave lw copies column tiles TPL lw .. TPL lw + TPL - 1 (TPL = 16 / RW_NL) of both k-steps of every slot, global -> LDS,
// AHEAD slots in flight.
constexpr int RW_TPL = 16 / RW_NL, RW_DPS = 2 * RW_TPL;  // tiles per loader and k-step, copies per loader and slot
template <int N>
__device__ __forceinline__ void wait_slots_behind(int behind) {  // vmcnt(RW_DPS x behind): s_waitcnt takes an immediate
    if constexpr (N > 0) {
        if (behind >= N) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RW_DPS * N) : "memory");
            return;
        }
        wait_slots_behind<N - 1>(behind);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
template <int NSLOT>
__device__ __forceinline__ void loader_main(const __bf16* __restrict__ tiles_y, char* ring, int lw, int lane, int dev) {
    constexpr int AHEAD = NSLOT - 1;
    static_assert(AHEAD >= 1 && RW_DPS * (AHEAD - 1) < 64, "vmcnt is a 6-bit counter");
    const __bf16* src = tiles_y + lane * 8;
    auto issue = [&](int s, int buf) {  // slot s: W1 (its two k-steps) or k-steps 2 j, 2 j + 1 of a 256 x 256 matrix
        if (dev & 1) return;  // development (RLX_ROWS_DEV bit 0, TIMING ONLY): no copies at all
        size_t off = 0;
        if (s >= 1) {
            const int t = s - 1, l = t >> 2, m = l == 0 ? 1 : l == 1 ? 2 : l == 2 ? 4 : 3;  // W2, W3, W3^T, W2^T
            off = (size_t)HID * Tiles::K1P + (size_t)(m - 1) * HID * HID + (size_t)(t & 3) * 2 * 16 * 512;
        }
        const __bf16* g = src + off;
        char* dst = ring + buf * RW_SLOT;
#pragma unroll
        for (int q = 0; q < RW_DPS; ++q) {
            const int nt = (q / RW_TPL) * 16 + RW_TPL * lw + (q % RW_TPL);  // tile index inside the slot: k-step q / TPL, column tile TPL lw + q % TPL
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
        __builtin_amdgcn_s_barrier();  // hand-off c: slot c has landed, slot c - 1 has been read by everybody
        asm volatile("" ::: "memory");
        if (c + AHEAD < RW_STEPS) {
            issue(c + AHEAD, fill);
            fill = fill + 1 == NSLOT ? 0 : fill + 1;
        }
        for (int e = rw_extra_barriers(c); e > 0; --e) __builtin_amdgcn_s_barrier();
    }
}

// k-step KS of a layer (slot C0 + KS / 2, half KS & 1): this wave's operands -- A fragments of its two row groups (their slabs), B
// fragments of its four column tiles (the ring).
template <int NSLOT, int C0, int KS>
__device__ __forceinline__ void read_frags(const char* ring_tile, const __bf16* xa0, const __bf16* xa1, bf16x8 (&a)[2], bf16x8 (&b)[RW_CT]) {
    if (KS > 0 && g_rows_dev_skip_reads) return;  // development (RLX_ROWS_DEV bit 1, TIMING ONLY): MFMAs on stale fragments
    const char* kstep = ring_tile + ((C0 + KS / 2) % NSLOT) * RW_SLOT + (KS & 1) * RW_KSTEP;
    a[0] = *reinterpret_cast<const bf16x8*>(xa0 + KS * 32);
    a[1] = *reinterpret_cast<const bf16x8*>(xa1 + KS * 32);
#pragma unroll
    for (int t = 0; t < RW_CT; ++t) b[t] = *reinterpret_cast<const bf16x8*>(kstep + t * 1024);
}

// acc[rt][t] = X[rows of group rt][0 : 32 NKS] . W^T for this wave's 2 row groups x 4 column tiles; X = the groups' slabs, W = ring
// slots C0 .. (two k-steps each).  Software-pipelined by one k-step:  4 MFMAs of k-step ks | (hand-off when k-step ks + 1 opens a
// slot) | the 6 fragment reads of k-step ks + 1 | the other 4 MFMAs of k-step ks.  MFMAs on BOTH sides of the hand-off: their
// operands were read one k-step earlier, so the matrix pipe starts again the moment the barrier releases, while the reads issue
// in its shadow (with the reads first the pipe idled ~150 cycles per hand-off).
template <int NSLOT, int C0, int NKS, int KS>
struct GemmSteps {
    static __device__ __forceinline__ void run(const char* ring_tile, const __bf16* xa0, const __bf16* xa1, f32x4 (&acc)[2][RW_CT], bf16x8 (&a)[2][2],
                                               bf16x8 (&b)[2][RW_CT]) {
        constexpr bool more = KS + 1 < NKS;
#define RLX_ROWS_MFMAS(G)                                                                                                            \
        _Pragma("unroll") for (int t = 2 * G; t < 2 * G + 2; ++t) {                                                                  \
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[KS & 1][0], b[KS & 1][t], acc[0][t], 0, 0, 0);                     \
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[KS & 1][1], b[KS & 1][t], acc[1][t], 0, 0, 0);                     \
        }
        RLX_ROWS_MFMAS(0)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (more) {
            if constexpr (((KS + 1) & 1) == 0) ring_barrier();  // hand-off of slot C0 + (KS + 1) / 2
            read_frags<NSLOT, C0, KS + 1>(ring_tile, xa0, xa1, a[(KS + 1) & 1], b[(KS + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        RLX_ROWS_MFMAS(1)
        __builtin_amdgcn_sched_barrier(0);
#undef RLX_ROWS_MFMAS
        if constexpr (more) GemmSteps<NSLOT, C0, NKS, KS + 1>::run(ring_tile, xa0, xa1, acc, a, b);
    }
};
// `slab_reused`: one more workgroup barrier behind the last k-step -- the caller's epilogue overwrites the slabs its partners may
// still be reading A fragments from.
template <int NSLOT, int C0, int NKS>
__device__ __forceinline__ void row_gemm(const char* ring, const __bf16* Xb0, const __bf16* Xb1, int cq, bool slab_reused, f32x4 (&acc)[2][RW_CT]) {
    static_assert(NKS % 2 == 0, "whole slots");
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    const char* ring_tile = ring + (cq * RW_CT) * 1024 + lane * 16;
    const __bf16 *xa0 = Xb0 + r16 * XSB + 8 * kq, *xa1 = Xb1 + r16 * XSB + 8 * kq;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < RW_CT; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2][2], b[2][RW_CT];
    ring_barrier();  // hand-off C0 (and: the slabs the previous phase wrote are complete)
    read_frags<NSLOT, C0, 0>(ring_tile, xa0, xa1, a[0], b[0]);
    GemmSteps<NSLOT, C0, NKS, 0>::run(ring_tile, xa0, xa1, acc, a, b);
    if (slab_reused) ring_barrier();
}

// Sum of a double over the wave, on the VALU's data-parallel-primitive paths: quad swaps, row half mirror, row mirror (every lane
// then holds its 16-lane row's sum), row broadcasts 15 / 31 (lane 63 holds the total).  A fixed tree, no LDS crossbar: as
// `__shfl_xor` butterflies (ds_bpermute + a dozen address instructions per step, two per double) the 9 + 7 metric sums of a
// workgroup's waves were the longest phase of the first versions of this kernel (10 - 18 k cycles).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xF, true);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);  // rows outside ROW_MASK add 0.0
}
__device__ __forceinline__ double wave_total_dpp(double v) {  // valid in lane 63
    v = dpp_add_f64<0xB1, 0xF>(v);   // quad_perm [1, 0, 3, 2]
    v = dpp_add_f64<0x4E, 0xF>(v);   // quad_perm [2, 3, 0, 1]
    v = dpp_add_f64<0x141, 0xF>(v);  // row_half_mirror
    v = dpp_add_f64<0x140, 0xF>(v);  // row_mirror
    v = dpp_add_f64<0x142, 0xA>(v);  // row_bcast15 into rows 1, 3
    v = dpp_add_f64<0x143, 0xC>(v);  // row_bcast31 into rows 2, 3
    return v;
}
template <int K0, int K1>
__device__ __forceinline__ void metric_partials(const double (&lacc)[NS], double* row /* LDS, [NS] */) {
    const int lane = threadIdx.x & 63;
    double tot[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) tot[k] = (k >= K0 && k < K1) ? wave_total_dpp(lacc[k]) : 0.0;
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < NS; ++k) row[k] = tot[k];
    }
}
// x of lane + J inside a 16-lane row (row_shl: lanes past the row read 0), the VALU form of __shfl_down for J < 16
template <int J>
__device__ __forceinline__ float row_down(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x100 + J, 0xF, 0xF, true));
}
// lp = ((0 + x[l]) + x[l + 1]) + ... + x[l + n - 1]  (n <= 16 lanes of one row, ascending like the reference's sum over the
// action dimension); n is wave-uniform
__device__ __forceinline__ float row_sum_ascending(float x, int n) {
    float s = fadd(0.f, x);
#define RLX_ROW_ADD(J) if (n > J) s = fadd(s, row_down<J>(x));
    RLX_ROW_ADD(1) RLX_ROW_ADD(2) RLX_ROW_ADD(3) RLX_ROW_ADD(4) RLX_ROW_ADD(5) RLX_ROW_ADD(6) RLX_ROW_ADD(7) RLX_ROW_ADD(8)
    RLX_ROW_ADD(9) RLX_ROW_ADD(10) RLX_ROW_ADD(11) RLX_ROW_ADD(12) RLX_ROW_ADD(13) RLX_ROW_ADD(14) RLX_ROW_ADD(15)
#undef RLX_ROW_ADD
    return s;
}

// a row group's 16 x 16 tiles (column tiles ct0 .. ct0 + 3) in the accumulator layout -> the k-tiled transposed image, 8 B per lane
__device__ __forceinline__ void store_image(const bf16x4 (&v)[RW_CT], __bf16* __restrict__ img, int nrb, int rb, int rhalf, int ct0) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
    const int kblk = 2 * rhalf + (kq >> 1);
#pragma unroll
    for (int t = 0; t < RW_CT; ++t)
        *reinterpret_cast<bf16x4*>(img + ((size_t)((ct0 + t) * nrb + rb) * 64 + kblk * 16 + r16) * 8 + 4 * (kq & 1)) = v[t];
}

template <int NSLOT, int W4R, bool STAMPS>
__global__ __launch_bounds__(RW_THREADS) void ppo_step_fused_bf16_rows_kernel(StepArgs a, __bf16* st_tiles) {
    typedef RowsLds<NSLOT, W4R> L;
    extern __shared__ __align__(16) char lds[];
    const rlx_mlp_layout& lay = a.lay;
    const rlx_ppo_loss_params& p = a.p;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    const int tile = blockIdx.x, y = blockIdx.y, D = lay.obs_dim;
    // wave (rp, cq): row pair rp = the 32 rows of row groups 2 rp, 2 rp + 1; column quarter cq = column tiles 4 cq .. 4 cq + 3.
    // A B fragment read from the ring serves both row groups: 6 LDS reads per 8 MFMAs (a wave per 16 rows x 128 columns needed 9 --
    // the k-steps were bound by the LDS: ~530 cycles against 272 of MFMA time).
    const int rp = wave >> 2, cq = wave & 3, ct0 = cq * RW_CT;
    const long long M = a.M, mp = (long long)tile * RW_BM + 32 * rp;      // the row pair's first row
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const long long n_adv = M * (lay.act_dim / p.raw_per_adv);
    const bool has_mask = a.loss_mask != nullptr, has_msum = a.loss_mask_sum != nullptr;
    const bool ratio_mode = p.max_episode_steps > 0 && has_mask && has_msum;
    const TileGeom tg{(int)((M + 31) / 32)};
    const int rb = (int)(mp >> 5);            // the pair IS one 32-row block of the images (group rt = its half rt)
    const bool img_live = rb < tg.nrb;        // wave-uniform: a pair past the last block stores no image

    char* ring = lds + L::RING;
    auto slab = [&](int g) { return reinterpret_cast<__bf16*>(lds + L::SLAB + g * L::SLAB_BYTES); };
    auto scratch = [&](int g) { return reinterpret_cast<float*>(lds + L::SCR + g * L::SCR_BYTES); };  // sP0 | sP1 | sH, [16][MAX_OUT] each
    __bf16 *Xb0 = slab(2 * rp), *Xb1 = slab(2 * rp + 1);
    float* sBias = reinterpret_cast<float*>(lds + L::BIAS);
    float* W4s = reinterpret_cast<float*>(lds + L::W4);
    float* b4s = W4s + W4R * W4S;
    float* sStd = b4s + MAX_OUT;
    double* sRed = reinterpret_cast<double*>(lds + L::RED);
    const __bf16* tiles = reinterpret_cast<const __bf16*>(a.tiles);
    __bf16* hy = reinterpret_cast<__bf16*>(a.h) + (size_t)(y * 2) * tg.mat();
    __bf16* dzy = reinterpret_cast<__bf16*>(a.dz) + (size_t)(y * 3) * tg.mat();

    touch_kernargs<(int)(sizeof(StepArgs) + 8)>();  // (the loss parameters, partial-buffer pointers and strides are first read far into the kernel)
    StampsT<STAMPS> ts{a.stamps, 0};
    ts.mark();
    double nm = 0.0;
    if (has_mask) {  // loss_mask.sum() of the whole micro-batch (every workgroup counts it: rows stay independent)
        double cnt[1] = {0.0};
        for (long long e = tid; e < n_adv; e += RW_THREADS) cnt[0] += a.loss_mask[e] != 0 ? 1.0 : 0.0;
        block_sum<1>(cnt, sRed);
        if (tid == 0) sRed[RW_NW + RW_NL] = cnt[0];
        __syncthreads();
        nm = sRed[RW_NW + RW_NL];
    }
    if (wave >= RW_NW) {
        loader_main<NSLOT>(tiles + Tiles::mat(y, 0), ring, wave - RW_NW, lane, a.xcd_rows);
        return;
    }

    // ---- the launch's global inputs ---------------------------------------------------------------------------------------------
    // states: the pair's four waves split its 32 rows (wave cq: rows 8 cq .. 8 cq + 7 = rows 8 (cq & 1) .. of group 2 rp + (cq >> 1)),
    // one (padded) column per lane
    static_assert(Tiles::K1P == 64, "one column per lane assumes a 64-wide padded first layer");
    float xs[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xs[u] = a.states[(size_t)min(mp + 8 * cq + u, M - 1) * D + min(lane, D - 1)];  // clamped, unconditional
    // Loss inputs.  A group's [16][n_out] elements are walked in passes of 64 (element e = lane + 64 u: row = e / n_out, o = e % n_out;
    // n_out is a power of two: fused_rows_eligible); the pair's 2 x npass (group, pass) jobs go round the four waves: wave cq takes
    // jobs cq, cq + 4 (job j: group j / npass of the pair, pass j % npass).
    constexpr int NPW = W4R / 8;  // jobs per wave: W4R / 4 passes cover 16 x n_out <= 16 W4R elements of a group
    const int epw = 16 * n_out, npass = (epw + 63) >> 6, osh = 31 - __builtin_clz(n_out), psh = 31 - __builtin_clz(npass);
    float v_a[NPW], v_b[NPW], v_c[NPW];  // policy: old log-prob, action, advantage;  critic: prev value, return, -
    unsigned m_on[NPW];
    float m_w[NPW];
    const float* safe = a.params;
    const int64_t* safe64 = reinterpret_cast<const int64_t*>(a.params);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = min(cq + 4 * i, 2 * npass - 1), jg = j >> psh, u = j & (npass - 1);
        const int e = min(lane + 64 * u, epw - 1), row = e >> osh, o = e & (n_out - 1);
        const size_t gr = (size_t)min(mp + 16 * jg + row, M - 1);
        const size_t ge = y == 1 ? gr : gr * n_out + o;  // loss element: one per row (action_level, one sub-group) / per value output
        if (y == 1) {
            v_a[i] = a.old_logprobs[gr * lay.act_dim + o];
            v_b[i] = a.action[gr * lay.act_dim + o];
            v_c[i] = a.advantages[gr];
        } else {
            v_a[i] = (p.has_critic ? a.prev_values : safe)[p.has_critic ? ge : 0];
            v_b[i] = (p.has_critic ? a.returns : safe)[p.has_critic ? ge : 0];
            v_c[i] = 0.f;
        }
        m_on[i] = (has_mask ? a.loss_mask : reinterpret_cast<const uint8_t*>(safe))[has_mask ? ge : 0];
        const long long ms = (has_msum ? a.loss_mask_sum : safe64)[has_msum ? ge : 0];
        m_w[i] = ratio_mode ? ((float)ms * 1.0f) / (float)p.max_episode_steps : 1.f;
    }
    SmallInputsB<RW_NT> si;
    si.issue(a.params, lay, y, n_out);
    {
        __bf16* Xs = slab(2 * rp + (cq >> 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) Xs[(8 * (cq & 1) + u) * XSB + lane] = (__bf16)((lane < D && mp + 8 * cq + u < M) ? xs[u] : 0.f);  // 64 columns: zero k tail
        si.commit(n_out, sBias, W4s, b4s, sStd);
        for (int i = n_out * W4S + tid; i < W4R * W4S; i += RW_NT) W4s[i] = 0.f;
        for (int i = tid; i < RW_NG * 3 * 16 * MAX_OUT; i += RW_NT) reinterpret_cast<float*>(lds + L::SCR)[i] = 0.f;  // (sH must be zero outside [16][n_out])
        // The loss inputs are first USED behind the forward sweep: hipcc would wait for them there with a vmcnt count that is safe on
        // every path -- including the one where the image stores are skipped (rows past M) -- i.e. on the normal path it waited for all
        // but the last three of the h1 / h2 image stores to reach the L2 (measured: 3 - 11 k cycles in the loss phase).  Consumed here,
        // behind the states commit, the wait costs nothing: everything above has been waited for already.
#pragma unroll
        for (int i = 0; i < NPW; ++i) asm volatile("" : "+v"(v_a[i]), "+v"(v_b[i]), "+v"(v_c[i]), "+v"(m_on[i]), "+v"(m_w[i]));
        wave_lds_fence();
        ts.mark();
        if (y == 1 && img_live) {  // the states' k-tiled image (B operand of the first layers' weight gradients): 4 column blocks x this wave's 8 rows
            const int cb = lane >> 4, c16 = lane & 15;
            bf16x8 v;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = Xs[(8 * (cq & 1) + jj) * XSB + cb * 16 + c16];
            *reinterpret_cast<bf16x8*>(st_tiles + ((size_t)(cb * tg.nrb + rb) * 64 + cq * 16 + c16) * 8) = v;
        }
    }

    // ---- forward -----------------------------------------------------------------------------------------------------
    f32x4 acc[2][RW_CT];
    // The rounded activations in the accumulator layout (1 - h^2 of the backward sweep): h2, h3 stay in registers; every lane
    // reads back the 8-byte words it stored into the h1 image (requested before the last GEMM, used behind it).
    bf16x4 kept[2][2][RW_CT], h1v[2][RW_CT];
    auto epilogue_tanh = [&](auto lc, bf16x4 (&keep)[2][RW_CT]) {
        constexpr int l = decltype(lc)::value;
#pragma unroll
        for (int t = 0; t < RW_CT; ++t) {
            const float bv = sBias[l * HID + (ct0 + t) * 16 + r16];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                bf16x4 hv;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 h = tanh2_b(f32x2{acc[rt][t][r] + bv, acc[rt][t][r + 1] + bv});
                    hv[r] = (__bf16)h.x;
                    hv[r + 1] = (__bf16)h.y;
                }
                store_slab_quad(rt == 0 ? Xb0 : Xb1, 0, (ct0 + t) * 16, hv);
                keep[rt][t] = hv;
            }
        }
    };
    auto store_images = [&](const bf16x4 (&v)[2][RW_CT], __bf16* img) {
        if (!img_live) return;
        store_image(v[0], img, tg.nrb, rb, 0, ct0);
        store_image(v[1], img, tg.nrb, rb, 1, ct0);
    };
    row_gemm<NSLOT, RW_C_L1, Tiles::K1P / 32>(ring, Xb0, Xb1, cq, true, acc);
    ts.mark();
    epilogue_tanh(std::integral_constant<int, 0>{}, h1v);
    store_images(h1v, hy);
    ts.mark();
    row_gemm<NSLOT, RW_C_L2, HID / 32>(ring, Xb0, Xb1, cq, true, acc);
    ts.mark();
    epilogue_tanh(std::integral_constant<int, 1>{}, kept[0]);
    store_images(kept[0], hy + tg.mat());
    ts.mark();
    row_gemm<NSLOT, RW_C_L3, HID / 32>(ring, Xb0, Xb1, cq, true, acc);
    ts.mark();
    epilogue_tanh(std::integral_constant<int, 2>{}, kept[1]);
    ring_barrier();  // extra 2 of 5: the groups' h3 slabs are complete
    ts.mark();

    // ---- head (f32 weights as three bf16 planes): the pair's four waves take (group, k half) = (cq & 1, cq >> 1); the two partials
    // of a group are added afterwards, each over four k-steps in ascending order -- the rollout launch's order.
    {
        const int hg = cq & 1, kh = cq >> 1;
        const __bf16* Xh = hg == 0 ? Xb0 : Xb1;
        f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < HID / 64; ++q) {
            const int k0 = (kh * (HID / 64) + q) * 32 + 8 * kq;
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(Xh + r16 * XSB + k0);
            const int wr_row = W4R == MAX_OUT ? r16 : min(r16, W4R - 1);
            f32x4 w0 = *reinterpret_cast<const f32x4*>(W4s + wr_row * W4S + k0), w1 = *reinterpret_cast<const f32x4*>(W4s + wr_row * W4S + k0 + 4);
            if (W4R != MAX_OUT && r16 >= W4R) w0 = w1 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float w[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
            hacc = mfma3(av, split3(w), hacc);
        }
        float* P = scratch(2 * rp + hg) + kh * 16 * MAX_OUT;  // sP0 / sP1 of that group
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(4 * kq + r) * MAX_OUT + r16] = hacc[r];
    }
    ring_barrier();  // extra 3 of 5: both partials of every group are complete
    ts.mark();

    // ---- loss element math (f32, identical to the column-split kernel) -----------------------------------------------------------
    const Denoms den = denominators(p, n_adv, nm, has_mask, has_msum);
    const float half_delta = (float)(0.5 * (double)p.huber_delta);
    double lacc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) lacc[k] = 0.0;
    const bool has_b4 = lay.off_b[y][3] >= 0;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int j = cq + 4 * i;
        if (j >= 2 * npass) break;  // wave-uniform
        const int jg = j >> psh, u = j & (npass - 1);
        float *sP0 = scratch(2 * rp + jg), *sP1 = sP0 + 16 * MAX_OUT, *sH = sP1 + 16 * MAX_OUT;
        const long long mg = mp + 16 * jg;
        const int e = lane + 64 * u;
        const bool mine = e < epw;
        const int row = mine ? e >> osh : 0, o = mine ? e & (n_out - 1) : 0;
        const bool valid = mine && mg + row < M;
        float sv = fadd(sP0[row * MAX_OUT + o], sP1[row * MAX_OUT + o]);
        if (has_b4) sv = fadd(sv, b4s[o]);
        const bool on = has_mask ? m_on[i] != 0 : true;
        const float w = m_w[i];
        if (y == 1) {
            // A row's n_out lanes are neighbours inside one 16-lane row of the wave (n_out divides 16): the row leader (o == 0) adds
            // the per-dimension log-probs in the reference's order (ascending, from 0.f) over n_out - 1 lane shifts, evaluates the loss
            // element and hands d(loss)/d(logprob) back to its lanes.
            const float d = fsub(v_b[i], sv);
            const float var = sStd[MAX_OUT + o], log_scale = sStd[2 * MAX_OUT + o];
            const float lpe = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
            const float lp = row_sum_ascending(lpe, n_out), old = row_sum_ascending(v_a[i], n_out);
            float gs = 0.f;
            if (valid && o == 0) {
                lacc[S_NM] += on ? 1.0 : 0.0;
                const float g = actor_elem(p, lp, old, v_c[i], on, w, ratio_mode, lacc);
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
                sP0[row * MAX_OUT + o] = dls;  // (this lane read its own sP0 word above; nobody else reads it before the exchange)
            }
        } else if (mine) {
            float gv = 0.f;
            if (valid && p.has_critic)
                gv = (a.grad_out * (float)(1.0 / den.critic)) * critic_elem(p, sv, v_a[i], v_b[i], on, w, ratio_mode, half_delta, lacc);
            sH[row * MAX_OUT + o] = gv;
        }
    }
    // Metric partials: every wave's sums -> its row of sRed (the other network's slots as zeros); behind the exchange barrier below
    // one wave adds the eight rows in wave order and writes the workgroup's ONE partial row (the weight-gradient launch's last block
    // adds every row of every tile: one row per wave made that block's serial walk the tail of its launch).
    static_assert(RW_NW * NS * sizeof(double) <= 1024, "the metric rows live in the reduction scratch");
    if (y == 1) metric_partials<0, S_VLOSS>(lacc, sRed + wave * NS);
    else metric_partials<S_VLOSS, NS>(lacc, sRed + wave * NS);
    // ---- head parameter gradients of the workgroup's 64 rows -----------------------------------------------------------------
    // dW4[o][j] = sum_rows dOut[row][o] h3[row][j] on the matrix pipe: M = o, N = j, K = rows.  Every wave parks its h3 tiles
    // (accumulator layout: lane (r16, kq) holds rows 4 kq .. 4 kq + 3 of column ct * 16 + r16) lane-linear in their groups' -- now
    // dead -- slabs; wave w then takes column tiles 2 w, 2 w + 1 over all 64 rows: two k-steps of 32 rows, k slot jj <-> row
    // (jj >> 2) * 16 + 4 kq + (jj & 3) of a group pair, dOut split into three bf16 planes (exact, see split3).
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        bf16x4* park = reinterpret_cast<bf16x4*>(rt == 0 ? Xb0 : Xb1);
#pragma unroll
        for (int t = 0; t < RW_CT; ++t) park[(ct0 + t) * 64 + lane] = kept[1][rt][t];
    }
    ring_barrier();  // extra 4 of 5: parked tiles, dOut and d(log std) rows of every group, and the metric rows are complete
    ts.mark();
    if (wave == RW_NW - 1 && lane < NS) {  // (the last wave: wave 0 carries the bias / log-std gradients below)
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < RW_NW; ++w) v += sRed[w * NS + lane];
        a.loss_part[((size_t)tile * 2 + y) * NS + lane] = v;
    }
    {
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.f;
        f32x4 g[2] = {zero4, zero4}, gb = zero4, gl = zero4;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            float av[8], lv[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float* scr = scratch(2 * pr + (jj >> 2));
                const int row = 4 * kq + (jj & 3);
                av[jj] = scr[2 * 16 * MAX_OUT + row * MAX_OUT + r16];  // sH of that group
                lv[jj] = scr[row * MAX_OUT + r16];                     // its d(loss)/d(log std) rows
            }
            const Split3 A = split3(av);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x4 lo = reinterpret_cast<const bf16x4*>(slab(2 * pr))[(2 * wave + c) * 64 + lane];
                const bf16x4 hi = reinterpret_cast<const bf16x4*>(slab(2 * pr + 1))[(2 * wave + c) * 64 + lane];
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
        for (int c = 0; c < 2; ++c) {
            const int j = (2 * wave + c) * 16 + r16;
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
    bf16x4 dv[2][RW_CT];
    {
        f32x4 dz[2][RW_CT];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < RW_CT; ++t) dz[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *sH0 = scratch(2 * rp) + 2 * 16 * MAX_OUT, *sH1 = scratch(2 * rp + 1) + 2 * 16 * MAX_OUT;
        for (int ks = 0; ks < (n_out + 3) / 4; ++ks) {
            const float av0 = sH0[r16 * MAX_OUT + 4 * ks + kq], av1 = sH1[r16 * MAX_OUT + 4 * ks + kq];
#pragma unroll
            for (int t = 0; t < RW_CT; ++t) {
                const float bv = W4s[(4 * ks + kq) * W4S + (ct0 + t) * 16 + r16];
                dz[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv, dz[0][t], 0, 0, 0);
                dz[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv, dz[1][t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < RW_CT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dv[rt][t][r] = (__bf16)(dz[rt][t][r] * dtanh_b(kept[1][rt][t][r]));
    }
    ring_barrier();  // extra 5 of 5: every wave has read the parked h3 tiles and the dOut rows: the slabs may be overwritten
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < RW_CT; ++t) store_slab_quad(rt == 0 ? Xb0 : Xb1, 0, (ct0 + t) * 16, dv[rt][t]);
    store_images(dv, dzy + 2 * tg.mat());
    ts.mark();

    // ---- backward-data chain -----------------------------------------------------------------------------------------------
    row_gemm<NSLOT, RW_C_B3, HID / 32>(ring, Xb0, Xb1, cq, true, acc);  // dH2 = dZ3 . W3
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < RW_CT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[rt][t][r] = (__bf16)(acc[rt][t][r] * dtanh_b(kept[0][rt][t][r]));
            store_slab_quad(rt == 0 ? Xb0 : Xb1, 0, (ct0 + t) * 16, dv[rt][t]);
        }
    store_images(dv, dzy + tg.mat());
    {   // h1 back from its image (clamped row block: a pair past the last block reads finite junk it never stores)
        const int rbc = min(rb, tg.nrb - 1);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < RW_CT; ++t)
                h1v[rt][t] = *reinterpret_cast<const bf16x4*>(hy + ((size_t)((ct0 + t) * tg.nrb + rbc) * 64 + (2 * rt + (kq >> 1)) * 16 + r16) * 8 + 4 * (kq & 1));
    }
    ts.mark();
    row_gemm<NSLOT, RW_C_B2, HID / 32>(ring, Xb0, Xb1, cq, false, acc);  // dH1 = dZ2 . W2
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < RW_CT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) dv[rt][t][r] = (__bf16)(acc[rt][t][r] * dtanh_b(h1v[rt][t][r]));
    store_images(dv, dzy);
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

// The row-group launch covers the embodied shapes: one loss element per row (action_level log-probs, one sub-group) with a
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
        hipLaunchKernelGGL((ppo_step_fused_bf16_rows_kernel<NSLOT, W4R, ST>), dim3(tiles64, 2), dim3(RW_THREADS), bytes, st, ad, stt); \
    }
#ifdef RLX_DEV_VARIANTS  // timing experiments that BREAK the results (no weight copies / no fragment reads): development builds only
    const int dev = dev_variant("RLX_ROWS_DEV", 0);
#else
    const int dev = 0;
#endif
    if (dev != 0 || a.xcd_rows != 0) {  // development only: timing experiments that break the results
        static int last = -1;
        if (last != dev) {
            const int skip = (dev >> 1) & 1;
            RLX_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_dev_skip_reads), &skip, sizeof(int)));
            last = dev;
        }
    }
    StepArgs ad = a;
    ad.xcd_rows = dev;
    const int nslot = dev_variant("RLX_ROWS_NSLOT", 3);  // 32 KiB ring slots (3 fit beside the 16-row head image, 2 beside the 8-row one too)
    if (a.stamps != nullptr) {
        if (op8) RLX_ROWS_LAUNCH(3, 8, true) else RLX_ROWS_LAUNCH(2, 16, true)
    } else if (op8 && nslot == 2) RLX_ROWS_LAUNCH(2, 8, false)
    else if (op8) RLX_ROWS_LAUNCH(3, 8, false)
    else RLX_ROWS_LAUNCH(2, 16, false)
#undef RLX_ROWS_LAUNCH
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // namespace step
}  // namespace rlx

#else  // product build: the row-split launch is not compiled

#include "ppo_step_common.h"

namespace rlx {
namespace step {
bool fused_rows_eligible(const rlx_mlp_layout&, const rlx_ppo_loss_params&) { return false; }
int launch_fused_rows_bf16(const StepArgs&, void*, int, hipStream_t) {
    set_error("the row-split fused launch is compiled into development builds only (-DRLX_DEV_VARIANTS)");
    return RLX_ENOSYS;
}
}  // namespace step
}  // namespace rlx

#endif  // RLX_DEV_VARIANTS
