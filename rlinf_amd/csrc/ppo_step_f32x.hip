// ppo_step_f32x.hip -- precision "32" (the precision the reference's YAML ships: examples/embodiment/config/model/mlp_policy.yaml:11,
// maniskill_ppo_mlp.yaml:98,128-131; rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:122-142) on the bf16 matrix pipe, gfx950.
//
// The f32 MFMA rate of this part is 1/16 of its bf16 rate, and the round-1 f32 launches (ppo_step.hip: 64-row tiles, 105 KB of LDS,
// v_mfma_f32_16x16x4_f32) ran at 113 + 57 us per optimizer step against 26 + 16 us in bf16 mode.  Here every f32 operand travels as
// THREE bf16 planes
//        x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)        (3 x 8 significand bits: exact)
// and every product a . b is six of the nine plane products on v_mfma_f32_16x16x32_bf16 with f32 accumulation
//        lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi          (dropped: mid.lo, lo.mid, lo.lo < 2^-24 |a||b|)
// -- the same real-number sum as the f32 dot product to below the rounding of the f32 accumulation itself
// (tests/test_bf16_split_exact.py), at 6 / 16 of the f32 MFMA time.  The split happens ONCE, at the producer: the optimizer (or
// rlx_mlp_pack_tiles) writes the weight image as three bf16 planes, the forward / backward epilogues write activations and
// pre-activation gradients as three planes (LDS slab and the k-tiled images of the weight-gradient launch), so no consumer spends
// VALU time on splitting (the round-3 f32 weight-gradient launch was VALU-bound on exactly that).
//
// Same decomposition as the bf16 launches (ppo_step_bf16.hip), whose pieces are reused (ppo_step_bf16_parts.h):
//   rollout_step_f32x_kernel    one launch per rollout step, 16-row tiles
//   ppo_step_fused_f32x_kernel  forward + loss + backward-data per 32-row tile and network; the activation slab is 3 bf16 planes
//                               (52 KB) -> one workgroup per CU, 8 waves, <= 256 VGPRs
//   ppo_step_dw_f32x_kernel     weight gradients: loader-wave LDS ring of 48 KiB k-blocks (16 tiles x 3 planes), 4 compute waves
// Replaces MLPPolicy.default_forward (mlp_policy.py:202-236), compute_ppo_actor_loss / compute_ppo_critic_loss (losses.py:170-380),
// EmbodiedFSDPActor.train_micro_batch (embodied_fsdp_actor_worker.py:591-700) at precision "32".  RLX_F32_EXACT_MFMA=1 selects the
// exact-f32-MFMA launches of ppo_step.hip instead (read once per process).

#include "ppo_step_bf16_parts.h"

namespace rlx {
namespace {

using namespace loss;
using namespace step;
using namespace b16;

constexpr int NPL = 3;  // bf16 planes per f32 operand

template <int RT, int NW>
struct GeoX {
    static constexpr int BM = 16 * RT, NT = 64 * NW, CT = HID / (16 * NW);
    static constexpr int PLANE = BM * XSB;              // bf16 elements of one slab plane
    static constexpr int SLAB_FLOATS = NPL * PLANE / 2;  // the three planes measured in floats
    static constexpr int BIAS_OFF = MAX_OUT * W4S + MAX_OUT + 8 * BM * MAX_OUT;  // same auxiliary map as GeoB
    static constexpr int AUX_FLOATS = BIAS_OFF + 3 * HID + 4 * MAX_OUT;
    static constexpr size_t LDS_BYTES = (size_t)(SLAB_FLOATS + AUX_FLOATS) * sizeof(float) + 4096;
};

constexpr size_t TILE_PLANE = 2 * Tiles::per_net();  // bf16 elements of one plane of the weight image (both networks)

// x -> (hi, mid, lo) for four values at once
struct Quad3 {
    bf16x4 p[NPL];
};
__device__ __forceinline__ Quad3 split_quad(const float (&v)[4]) {
    Quad3 q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const __bf16 h = (__bf16)v[r];
        const float r1 = fsub(v[r], (float)h);
        const __bf16 m = (__bf16)r1;
        q.p[0][r] = h;
        q.p[1][r] = m;
        q.p[2][r] = (__bf16)fsub(r1, (float)m);
    }
    return q;
}
// The six plane products in the order they are accumulated, smallest terms first (plane 0 = hi, 1 = mid, 2 = lo): product q is
// A plane PA[q] x B plane PB[q].  A GEMM step walks PRODUCT-major over its independent accumulators (mfma6_tile below), so that two
// consecutive MFMAs never share an accumulator: back-to-back MFMAs on ONE accumulator stall on its latency (MI355X_MICROARCH.md:
// +43 cycles for any extra issue slot between them), and six of them in a row per tile were what the first version of these kernels
// spent most of its k-step on.  The order per accumulator is unchanged by the interleaving: bit-identical sums.
__device__ constexpr int PA(int q) { return q == 0 ? 2 : (q == 2 || q == 3) ? 1 : 0; }
__device__ constexpr int PB(int q) { return q == 1 ? 2 : (q == 2 || q == 4) ? 1 : 0; }
template <int RH, int R0, int RT, int CT>  // row tiles R0 .. R0 + RH - 1 of acc
__device__ __forceinline__ void mfma6_tile(const bf16x8 (&a)[RH][NPL], const bf16x8 (&b)[CT][NPL], f32x4 (&acc)[RT][CT]) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int rt = 0; rt < RH; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[R0 + rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][PA(q)], b[ct][PB(q)], acc[R0 + rt][ct], 0, 0, 0);
}
// one accumulator (the heads: a handful of MFMAs per tile)
__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&a)[NPL], const bf16x8 (&b)[NPL], f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
}

// Weight image: three planes, each the bf16 k-step-major tile image of pack_tiles_bf16_kernel (ppo_step_bf16.hip):
//   plane p, element ((it * 16 + nb) * 64 + l) * 8 + j  holds plane p of  W[nb*16 + (l & 15)][it*32 + 8*(l >> 4) + j]
__global__ __launch_bounds__(256) void pack_tiles_f32x_kernel(const float* __restrict__ params, rlx_mlp_layout lay,
                                                              __bf16* __restrict__ tiles) {
    const size_t total = TILE_PLANE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / Tiles::per_net());
        size_t r = i - y * Tiles::per_net();
        int m = 0;
        if (r >= (size_t)HID * Tiles::K1P) {
            r -= (size_t)HID * Tiles::K1P;
            m = 1 + (int)(r / ((size_t)HID * HID));
            r %= (size_t)HID * HID;
        }
        const int j = (int)(r & 7), l = (int)((r >> 3) & 63);
        const int q = (int)(r >> 9), nb = q % (HID / 16), it = q / (HID / 16);
        const int n = nb * 16 + (l & 15), k = it * 32 + 8 * (l >> 4) + j;
        float v;
        if (m == 0) v = k < lay.obs_dim ? params[lay.off_w[y][0] + (size_t)n * lay.obs_dim + k] : 0.f;
        else if (m <= 2) v = params[lay.off_w[y][m] + (size_t)n * HID + k];
        else v = params[lay.off_w[y][m - 2] + (size_t)k * HID + n];
        const __bf16 h = (__bf16)v;
        const float r1 = fsub(v, (float)h);
        const __bf16 md = (__bf16)r1;
        tiles[i] = h;
        tiles[total + i] = md;
        tiles[2 * total + i] = (__bf16)fsub(r1, (float)md);
    }
}

template <int RT, int CT>
__device__ __forceinline__ void zero_acc_x(f32x4 (&acc)[RT][CT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc = X[0:BM, 0:32*nit] . W^T: weights as three planes of fragment tiles streamed L2 -> registers (ring of PD stages), the A
// operand from the three slab planes; no barrier inside the loop, one at the end.  Six MFMAs per (row tile, column tile, k-step).
template <int RT, int NW, int PD>
struct RowGemmX {
    typedef GeoX<RT, NW> G;
    static constexpr int CT = G::CT, KI = 32, MAXIT = HID / KI;
    bf16x8 bq[PD][CT][NPL];
    const __bf16* wbase;
    int nit;

    __device__ __forceinline__ void gload(int it, bf16x8 (&b)[CT][NPL]) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                b[ct][p] = *reinterpret_cast<const bf16x8*>(wbase + p * TILE_PLANE + (size_t)(it * (HID / 16) + ct) * 512);
    }
    __device__ __forceinline__ void prefetch(const __bf16* __restrict__ P, int nit_) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        nit = nit_;
        wbase = P + (size_t)(wave * CT) * 512 + lane * 8;
#pragma unroll
        for (int d = 0; d < PD; ++d)
            if (d < nit) gload(d, bq[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int R0, int RH>
    __device__ __forceinline__ void pass(const __bf16* Xb, int it, int r16, int kb, f32x4 (&acc)[RT][CT]) {
        bf16x8 a[RH][NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int rt = 0; rt < RH; ++rt)
                a[rt][p] = *reinterpret_cast<const bf16x8*>(Xb + p * G::PLANE + ((R0 + rt) * 16 + r16) * XSB + it * KI + 8 * kb);
        mfma6_tile<RH, R0, RT, CT>(a, bq[it % PD], acc);
    }
    __device__ __forceinline__ void run(const __bf16* Xb, f32x4 (&acc)[RT][CT]) {
        const int lane = threadIdx.x & 63, r16 = lane & 15, kb = lane >> 4;
        zero_acc_x(acc);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if (it >= nit) break;
            constexpr int RH = RT < 2 ? RT : 2;  // row tiles per pass: 2 x CT independent accumulators, 6 A fragments live
            pass<0, RH>(Xb, it, r16, kb, acc);
            if constexpr (RT > 2) pass<2, RH>(Xb, it, r16, kb, acc);
            if (it + PD < nit) {
                gload(it + PD, bq[it % PD]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();
    }
};

// k-tiled transposed bf16 image (the weight-gradient GEMM's operand layout, see store_tiles in ppo_step_bf16.hip), one plane
template <int RT, int CT>
__device__ __forceinline__ void store_tiles_x(const bf16x4 (&v)[RT][CT], __bf16* __restrict__ dst, int nrb, long long m0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int rb = (int)(m0 >> 5) + (rt >> 1);
        if (rb >= nrb) continue;
        const int kblk = 2 * (rt & 1) + (kq >> 1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int cb = wave * CT + ct;
            *reinterpret_cast<bf16x4*>(dst + ((size_t)(cb * nrb + rb) * 64 + kblk * 16 + r16) * 8 + 4 * (kq & 1)) = v[rt][ct];
        }
    }
}

// ... and the way back: the bf16x4 this lane stored for (rt, ct) (the backward sweep re-reads h1 / h2 for 1 - h^2 instead of
// carrying 24 f32 registers per layer through the kernel: that is what lets a workgroup own 64 rows)
template <int RT, int CT>
__device__ __forceinline__ void load_tiles_x(bf16x4 (&v)[RT][CT], const __bf16* __restrict__ src, int nrb, long long m0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int rb = min((int)(m0 >> 5) + (rt >> 1), nrb - 1);
        const int kblk = 2 * (rt & 1) + (kq >> 1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int cb = wave * CT + ct;
            v[rt][ct] = *reinterpret_cast<const bf16x4*>(src + ((size_t)(cb * nrb + rb) * 64 + kblk * 16 + r16) * 8 + 4 * (kq & 1));
        }
    }
}
// 1 - h^2 from h's three planes: hi + mid + lo is h exactly, so this is the forward sweep's own `1.f - h * h`
__device__ __forceinline__ float dtanh_x(__bf16 hi, __bf16 mid, __bf16 lo) {
    const float h = fadd(fadd((float)hi, (float)mid), (float)lo);
    return fsub(1.f, fmul(h, h));
}

// obs-preprocess into the three slab planes (k tail and rows past M zero); issue / commit split as in StatesB
template <int RT, int NW>
struct StatesX {
    typedef GeoX<RT, NW> G;
    static_assert(Tiles::K1P == 64, "one pass per lane assumes a 64-wide padded first layer");
    static constexpr int UB = G::BM * Tiles::K1P / G::NT;
    float x[UB];
    __device__ __forceinline__ void issue(const float* __restrict__ states, int D, long long m0, long long M) {
        const int kp = round_up(D, KPAD);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = min((int)threadIdx.x + u * G::NT, G::BM * kp - 1);
            const int r = i / kp, c = i % kp;
            x[u] = states[(size_t)min(m0 + r, M - 1) * D + min(c, D - 1)];
        }
    }
    __device__ __forceinline__ void commit(float* __restrict__ states_copy, int D, long long m0, long long M, __bf16* Xb) const {
        const int kp = round_up(D, KPAD);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = (int)threadIdx.x + u * G::NT;
            if (i >= G::BM * kp) continue;
            const int r = i / kp, c = i % kp;
            const bool ok = c < D && m0 + r < M;
            if (ok && states_copy) states_copy[(size_t)(m0 + r) * D + c] = x[u];
            const float v = ok ? x[u] : 0.f;
            const __bf16 h = (__bf16)v;
            const float r1 = fsub(v, (float)h);
            const __bf16 md = (__bf16)r1;
            Xb[r * XSB + c] = h;
            Xb[G::PLANE + r * XSB + c] = md;
            Xb[2 * G::PLANE + r * XSB + c] = (__bf16)fsub(r1, (float)md);
        }
    }
    // after an LDS barrier behind commit(): the three planes of the states image (4 column blocks each)
    __device__ __forceinline__ static void tiles(__bf16* __restrict__ st_tiles, int nrb, int D, long long m0, const __bf16* Xb) {
        const int kp = round_up(D, KPAD);
        const size_t plane = (size_t)4 * nrb * 512;
        for (int u = threadIdx.x; u < NPL * 4 * (G::BM / 8) * 16; u += G::NT) {
            const int p = u / (4 * (G::BM / 8) * 16), w = u % (4 * (G::BM / 8) * 16);
            const int cb = w / ((G::BM / 8) * 16), ko = (w / 16) % (G::BM / 8), c16 = w & 15;
            const int rb = (int)(m0 >> 5) + (ko >> 2);
            if (rb >= nrb) continue;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = cb * 16 + c16 < kp ? Xb[p * G::PLANE + (8 * ko + j) * XSB + cb * 16 + c16] : (__bf16)0.f;
            *reinterpret_cast<bf16x8*>(st_tiles + p * plane + ((size_t)(cb * nrb + rb) * 64 + (ko & 3) * 16 + c16) * 8) = v;
        }
    }
};

// forward hidden-layer epilogue: h = tanh(acc + bias) (f32: the polynomial / exp form of ppo_step.hip) -> three slab planes;
// hp (optional): the planes in the accumulator layout; dst (optional): the k-tiled images (plane stride img_plane) -- which the
// backward sweep also reads back for 1 - h^2.
template <int RT, int NW>
__device__ __forceinline__ void epilogue_tanh_x(const f32x4 (&acc)[RT][GeoX<RT, NW>::CT], const float* bias, __bf16* Xb,
                                                bf16x4 (*hp)[RT][GeoX<RT, NW>::CT],
                                                __bf16* __restrict__ dst, size_t img_plane, int nrb, long long m0) {
    typedef GeoX<RT, NW> G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15;
    bf16x4 loc[NPL][RT][G::CT];
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) {
        const float b = bias[wave * 16 * G::CT + ct * 16 + r16];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = fast_tanh(acc[rt][ct][r] + b);
            const Quad3 q = split_quad(h);
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                store_slab_quad(Xb + p * G::PLANE, rt * 16, wave * 16 * G::CT + ct * 16, q.p[p]);
                loc[p][rt][ct] = q.p[p];
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        if (dst != nullptr) store_tiles_x<RT, G::CT>(loc[p], dst + p * img_plane, nrb, m0);
        if (hp != nullptr) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < G::CT; ++ct) hp[p][rt][ct] = loc[p][rt][ct];
        }
    }
    lds_barrier();
}

// head forward: out[row][o] = h3[row][:] . W4[o][:] as two k-half partials P0 / P1 (see head_forward_mfma): h3 from the three slab
// planes, W4 (f32, LDS) split on the fly -- six plane products per 32-k step.  Rollout (RT = 1) and training (RT = 2) add every
// row's products in the same order: a sample's rollout log-prob and the first epoch's recomputation are bit-identical.
template <int RT, int PLANE>
__device__ __forceinline__ void head_forward_mfma_x(const __bf16* Xb, const float* W4s, float* P0, float* P1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
    if (wave >= 2 * RT) return;
    const int rt = wave % RT, kh = wave / RT;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < HID / 64; ++q) {
        const int k0 = (kh * (HID / 64) + q) * 32 + 8 * kq;
        bf16x8 a[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) a[p] = *reinterpret_cast<const bf16x8*>(Xb + p * PLANE + (rt * 16 + r16) * XSB + k0);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(W4s + r16 * W4S + k0), w1 = *reinterpret_cast<const f32x4*>(W4s + r16 * W4S + k0 + 4);
        const float w[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        const Split3 s = split3(w);
        const bf16x8 b[NPL] = {s.hi, s.mid, s.lo};
        acc = mfma6(a, b, acc);
    }
    float* P = kh == 0 ? P0 : P1;
#pragma unroll
    for (int r = 0; r < 4; ++r) P[(rt * 16 + 4 * kq + r) * MAX_OUT + r16] = acc[r];
}

// ---------------------------------------------------------------------------------------------------------------
// rollout step (RT = 1, NW = 8): the launch of rollout_step_bf16_kernel with three-plane operands
// ---------------------------------------------------------------------------------------------------------------
template <int PD>  // weight-ring depth (development: RLX_F32X_ROLLOUT_PD)
__global__ __launch_bounds__(512) void rollout_step_f32x_kernel(RolloutArgs a) {
    constexpr int RT = 1, NW = 8;
    touch_kernargs<(int)sizeof(RolloutArgs)>();
    typedef GeoX<RT, NW> G;
    extern __shared__ __align__(16) float smem[];
    __bf16* Xb = reinterpret_cast<__bf16*>(smem);
    float* W4s = smem + G::SLAB_FLOATS;
    float* b4s = W4s + MAX_OUT * W4S;
    const rlx_mlp_layout& lay = a.lay;
    const int D = lay.obs_dim, tid = threadIdx.x;
    const __bf16* tiles = reinterpret_cast<const __bf16*>(a.tiles);

    int b = blockIdx.x, y, job;
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
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    float* sBias = smem + G::SLAB_FLOATS + G::BIAS_OFF;
    float* sStd = sBias + 3 * HID;

    // every small global input of the tile is requested before anything waits (see rollout_step_bf16_kernel); the weight planes do
    // not fit in registers for all three layers at once: a PD-deep ring per layer, the next layer's first stages requested before
    // the epilogue that produces its input
    StatesX<RT, NW> st;
    st.issue(states, D, m0, M);
    SmallInputsB<G::NT> si;
    si.issue(a.params, lay, y, n_out);
    const int orow = tid / n_out, oo = tid % n_out;
    const bool olive = tid < G::BM * n_out && m0 + orow < M;
    const size_t og = (size_t)min(m0 + orow, M - 1) * n_out + oo;
    const float epsv = (a.eps != nullptr ? a.eps : a.params)[(job == 0 && y == 1 && a.eps != nullptr && tid < G::BM * n_out) ? og : 0];
    const ValuePre vp = value_job_prefetch(a.vj[max(job - 1, 0)], job != 0, (size_t)min(m0 + orow, M - 1), a.params);
    __builtin_amdgcn_sched_barrier(0);
    RowGemmX<RT, NW, PD> gemm;
    gemm.prefetch(tiles + Tiles::mat(y, 0), Tiles::K1P / 32);
    st.commit(states_copy, D, m0, M, Xb);
    si.commit(n_out, sBias, W4s, b4s, sStd);
    for (int i = n_out * W4S + tid; i < MAX_OUT * W4S; i += G::NT) W4s[i] = 0.f;
    lds_barrier();
    f32x4 acc[RT][G::CT];
    gemm.run(Xb, acc);
    gemm.prefetch(tiles + Tiles::mat(y, 1), HID / 32);
    epilogue_tanh_x<RT, NW>(acc, sBias, Xb, nullptr, nullptr, 0, 0, m0);
    gemm.run(Xb, acc);
    gemm.prefetch(tiles + Tiles::mat(y, 2), HID / 32);
    epilogue_tanh_x<RT, NW>(acc, sBias + HID, Xb, nullptr, nullptr, 0, 0, m0);
    gemm.run(Xb, acc);
    epilogue_tanh_x<RT, NW>(acc, sBias + 2 * HID, Xb, nullptr, nullptr, 0, 0, m0);

    static_assert(G::BM * MAX_OUT <= G::NT, "one head output per lane");
    float* sP0 = b4s + MAX_OUT, *sP1 = sP0 + G::BM * MAX_OUT;
    head_forward_mfma_x<RT, G::PLANE>(Xb, W4s, sP0, sP1);
    lds_barrier();
    if (tid < G::BM * n_out) {
        float s = fadd(sP0[orow * MAX_OUT + oo], sP1[orow * MAX_OUT + oo]);
        if (lay.off_b[y][3] >= 0) s = fadd(s, b4s[oo]);
        if (!olive) return;
        if (job == 0 && y == 0) {
            a.value[og] = s;
        } else if (job == 0) {
            const float mean = s;
            const float stdv = sStd[oo];
            const float act = a.eps ? fadd(fmul(epsv, stdv), mean) : mean;
            const float d = fsub(act, mean);
            const float var = sStd[MAX_OUT + oo];
            const float log_scale = sStd[2 * MAX_OUT + oo];
            a.logprob[og] = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
            a.action[og] = act;
        } else {
            value_job_output(a.vj[job - 1], og, (size_t)(m0 + orow), oo, s, &vp);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fused optimizer-step kernel (1), f32 through bf16 planes.  Images (plane-major inside every matrix, nrb = ceil(M / 32)):
//   h  : [2 nets][2][3 planes][16 cb][nrb] tiles    dz : [2 nets][3][3 planes][16 cb][nrb] tiles    st : [3 planes][4 cb][nrb] tiles
// ---------------------------------------------------------------------------------------------------------------
template <int RT, int NW, int PD, bool DEC, bool STAMPS = false>  // STAMPS: development build with phase stamps (tools/phase_times.py)
__global__ __launch_bounds__(64 * NW, 2) void ppo_step_fused_f32x_kernel(StepArgs a, __bf16* st_tiles) {
    typedef GeoX<RT, NW> G;
    constexpr int BM = G::BM, CT = G::CT;
    extern __shared__ __align__(16) float smem[];
    __bf16* Xb = reinterpret_cast<__bf16*>(smem);
    float* W4s = smem + G::SLAB_FLOATS;
    float* b4s = W4s + MAX_OUT * W4S;
    float* sHead = b4s + MAX_OUT;
    float* sLp = sHead + BM * MAX_OUT;
    float* sG = sLp + BM * MAX_OUT;
    float* sD = sG + BM * MAX_OUT;
    float* sOld = sD + BM * MAX_OUT;
    float* sAct = sOld + BM * MAX_OUT;
    float* sAdv = sAct + BM * MAX_OUT;
    float* sRet = sAdv + BM * MAX_OUT;
    double* sRed = reinterpret_cast<double*>(smem + G::SLAB_FLOATS + G::AUX_FLOATS);
    double* sNm = sRed + 256;

    const rlx_mlp_layout& lay = a.lay;
    const rlx_ppo_loss_params& p = a.p;
    const int y = blockIdx.y, tile = blockIdx.x, D = lay.obs_dim, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const long long m0 = (long long)tile * BM, M = a.M;
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const int npr = lay.act_dim / p.raw_per_adv;
    const long long n_adv = M * npr;
    const bool has_mask = a.loss_mask != nullptr;
    const TileGeom tg{(int)((M + 31) / 32)};
    const size_t img = tg.mat();  // elements of one plane of one [M][256] image
    const __bf16* tiles = reinterpret_cast<const __bf16*>(a.tiles);
    __bf16* hy = reinterpret_cast<__bf16*>(a.h) + (size_t)(y * 2) * NPL * img;
    __bf16* dzy = reinterpret_cast<__bf16*>(a.dz) + (size_t)(y * 3) * NPL * img;

    if (has_mask) {
        double cnt[1] = {0.0};
        for (long long e = tid; e < n_adv; e += G::NT) cnt[0] += a.loss_mask[e] != 0 ? 1.0 : 0.0;
        block_sum<1>(cnt, sRed);
        if (tid == 0) sNm[0] = cnt[0];
    }

    // ---- forward -----------------------------------------------------------------------------------------------------
    StampsT<STAMPS> ts{a.stamps, 0};
    ts.mark();
    float* sBias = smem + G::SLAB_FLOATS + G::BIAS_OFF;
    float* sStd = sBias + 3 * HID;
    StatesX<RT, NW> st;
    st.issue(a.states, D, m0, M);
    constexpr int PI = BM * MAX_OUT / G::NT;
    static_assert(BM * MAX_OUT % G::NT == 0, "staging assumes whole iterations");
    float v0[PI], v1[PI], v2[PI];
#pragma unroll
    for (int u = 0; u < PI; ++u) {
        const int i = tid + u * G::NT, row = i / MAX_OUT, c = i % MAX_OUT;
        const size_t gr = (size_t)min(m0 + row, M - 1);
        if (y == 1) {
            v0[u] = a.old_logprobs[gr * lay.act_dim + min(c, lay.act_dim - 1)];
            v1[u] = a.action[gr * lay.act_dim + min(c, lay.act_dim - 1)];
            v2[u] = a.advantages[gr * npr + min(c, npr - 1)];
        } else {
            v0[u] = p.has_critic ? a.prev_values[gr * n_out + min(c, n_out - 1)] : 0.f;
            v1[u] = p.has_critic ? a.returns[gr * n_out + min(c, n_out - 1)] : 0.f;
            v2[u] = 0.f;
        }
    }
    SmallInputsB<G::NT> si;
    si.issue(a.params, lay, y, n_out);
    __builtin_amdgcn_sched_barrier(0);
    RowGemmX<RT, NW, PD> gemm;
    gemm.prefetch(tiles + Tiles::mat(y, 0), Tiles::K1P / 32);
    st.commit(nullptr, D, m0, M, Xb);
#pragma unroll
    for (int u = 0; u < PI; ++u) {
        const int i = tid + u * G::NT;
        if (y == 1) {
            sOld[i] = v0[u];
            sAct[i] = v1[u];
            sAdv[i] = v2[u];
        } else {
            sAdv[i] = v0[u];
            sRet[i] = v1[u];
        }
    }
    si.commit(n_out, sBias, W4s, b4s, sStd);
    for (int i = tid; i < BM * MAX_OUT; i += G::NT) sHead[i] = 0.f;
    for (int i = n_out * W4S + tid; i < MAX_OUT * W4S; i += G::NT) W4s[i] = 0.f;
    lds_barrier();
    if (y == 1) StatesX<RT, NW>::tiles(st_tiles, tg.nrb, D, m0, Xb);
    ts.mark();
    f32x4 acc[RT][CT];
    bf16x4 h3p[NPL][RT][CT];   // h3's planes in the accumulator layout: B operand of the head parameter gradients, and 1 - h3^2
    gemm.run(Xb, acc);
    gemm.prefetch(tiles + Tiles::mat(y, 1), HID / 32);
    ts.mark();
    epilogue_tanh_x<RT, NW>(acc, sBias, Xb, nullptr, hy, img, tg.nrb, m0);
    ts.mark();
    gemm.run(Xb, acc);
    gemm.prefetch(tiles + Tiles::mat(y, 2), HID / 32);
    ts.mark();
    epilogue_tanh_x<RT, NW>(acc, sBias + HID, Xb, nullptr, hy + NPL * img, img, tg.nrb, m0);
    ts.mark();
    gemm.run(Xb, acc);
    ts.mark();
    epilogue_tanh_x<RT, NW>(acc, sBias + 2 * HID, Xb, h3p, nullptr, 0, tg.nrb, m0);
    ts.mark();

    // ---- head + loss element math (fused_loss_pass, ppo_step_bf16_parts.h: the bf16 launch's, f32 throughout) --------------------
    head_forward_mfma_x<RT, G::PLANE>(Xb, W4s, sG, sLp);
    lds_barrier();
    ts.mark();
    fused_loss_pass<BM, NW, DEC>(a, y, tile, m0, LossLds{b4s, sHead, sLp, sG, sD, sOld, sAct, sAdv, sRet, sStd, reinterpret_cast<double*>(Xb), sNm}, ts);

    // ---- head parameter gradients per 32-row half tile: dW4[o][j] = sum_rows dOut[row][o] h3[row][j] (see the bf16 kernel) -------
    {
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.f;
#pragma unroll
        for (int sub = 0; sub < BM / 32; ++sub) {
            float* part = a.head_part + ((size_t)(tile * (BM / 32) + sub) * 2 + y) * a.head_stride;
            float av[8], lv[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int row = sub * 32 + (jj >> 2) * 16 + 4 * kq + (jj & 3);
                av[jj] = sHead[row * MAX_OUT + r16];
                lv[jj] = sLp[row * MAX_OUT + r16];
            }
            const Split3 A = split3(av);
            const bf16x8 Ap[NPL] = {A.hi, A.mid, A.lo};
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bf16x8 B[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) B[pl][jj] = h3p[pl][2 * sub + (jj >> 2)][ct][jj & 3];
                const f32x4 g = mfma6(Ap, B, zero4);
                const int j = wave * 16 * CT + ct * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * kq + r < n_out) part[(4 * kq + r) * HID + j] = g[r];
            }
            if (wave == 0) {  // bias / log-std gradients: row sums, i.e. the same product against a fragment of ones
                const f32x4 gb = mfma3(A, ones, zero4);
                const f32x4 gl = y == 1 ? mfma3(split3(lv), ones, zero4) : zero4;
                if (r16 == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * kq + r < n_out) {
                            part[n_out * HID + 4 * kq + r] = gb[r];
                            part[n_out * HID + n_out + 4 * kq + r] = gl[r];
                        }
                }
            }
        }
    }

    ts.mark();
    // ---- dZ3 = (dOut . W4) * (1 - h3^2): K = n_out is tiny, f32 operands -> v_mfma_f32_16x16x4_f32 -----------------------------
    gemm.prefetch(tiles + Tiles::mat(y, 4), HID / 32);  // W3^T
    zero_acc_x(acc);
    for (int ks = 0; ks < (n_out + 3) / 4; ++ks) {
        float av[RT], bv[CT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) av[rt] = sHead[(rt * 16 + r16) * MAX_OUT + 4 * ks + kq];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bv[ct] = W4s[(4 * ks + kq) * W4S + wave * 16 * CT + ct * 16 + r16];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[ct], acc[rt][ct], 0, 0, 0);
    }
    lds_barrier();  // every read of h3's slab (the loss pass's scratch) / sHead is done: the slab may be overwritten
    // dZ_l = acc * (1 - h_l^2) -> three planes -> slab (the next GEMM's input) + the k-tiled images; one (rt, ct) quad at a time
    auto emit = [&](int l, const bf16x4 (&hpl)[NPL][RT][CT], bool to_slab) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            bf16x4 row[NPL][CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float d4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) d4[r] = acc[rt][ct][r] * dtanh_x(hpl[0][rt][ct][r], hpl[1][rt][ct][r], hpl[2][rt][ct][r]);
                const Quad3 q = split_quad(d4);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    row[pl][ct] = q.p[pl];
                    if (to_slab) store_slab_quad(Xb + pl * G::PLANE, rt * 16, wave * 16 * CT + ct * 16, q.p[pl]);
                }
            }
            // (store_tiles_x for one row tile: rt enters through m0's row block and the k block)
            const int rb = (int)(m0 >> 5) + (rt >> 1);
            if (rb < tg.nrb) {
                const int kblk = 2 * (rt & 1) + (kq >> 1);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int cb = wave * CT + ct;
                        *reinterpret_cast<bf16x4*>(dzy + (size_t)(l * NPL + pl) * img + ((size_t)(cb * tg.nrb + rb) * 64 + kblk * 16 + r16) * 8 + 4 * (kq & 1)) = row[pl][ct];
                    }
            }
        }
    };
    emit(2, h3p, true);
    lds_barrier();
    ts.mark();

    // ---- backward-data chain ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int l = 2; l >= 1; --l) {
        bf16x4 hl[NPL][RT][CT];  // h_l's planes, read back from the images the forward sweep wrote (same lanes, same words)
        if (l == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's image stores have been acknowledged
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) load_tiles_x<RT, CT>(hl[pl], hy + (size_t)((l - 1) * NPL + pl) * img, tg.nrb, m0);
        gemm.run(Xb, acc);
        if (l == 2) gemm.prefetch(tiles + Tiles::mat(y, 3), HID / 32);  // W2^T
        ts.mark();
        emit(l - 1, hl, l == 2);
        if (l == 2) lds_barrier();
        ts.mark();
    }
    if constexpr (STAMPS) {
        if (a.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 1 && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            a.stamps[ts.n] = (long long)clock64();  // ... and the stores have drained
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fused optimizer-step kernel (2): weight gradients dW_l = dZ_l^T H_{l-1} from the three-plane images.  The loader-wave ring of
// ppo_step_dw_bf16_ring_kernel (same items, same 128 x 128 tiles, 4 compute waves as 2 x 2 of 64 x 64, 2 loader waves) with
// 48 KiB k-blocks -- the 8 dZ^T tiles and the 8 H^T tiles of 32 batch rows, three planes each -- and six MFMAs per tile pair.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DWX_THREADS = 384, DWX_NBUF = 3, DWX_BUF_BYTES = NPL * 16 * 1024;

__device__ __forceinline__ void dwx_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(DWX_THREADS) void ppo_step_dw_f32x_kernel(DwArgs a, const __bf16* __restrict__ st_tiles) {
    extern __shared__ __align__(16) char dsm[];
    const rlx_mlp_layout& lay = a.lay;
    const long long M = a.M;
    const int tid = threadIdx.x;
    const int gemm_blocks = round_up(a.gemm_items, 8);
    int b = blockIdx.x;

    if (b >= gemm_blocks) {
        b -= gemm_blocks;
        if (b < a.slabs * 2) {
            if (tid < 256) head_reduce_block(a, b >> 1, b & 1, tid);
        } else {
            double* s_red = reinterpret_cast<double*>(dsm);
            metric_block(a, s_red, tid, DWX_THREADS);
        }
        return;
    }
    const int item = (b & 7) * (gemm_blocks >> 3) + (b >> 3);
    if (item >= a.gemm_items) return;
    const int s = item / 20, w = item % 20;
    int y, l, i0, j0;
    if (w < 16) {
        const int mat = w >> 2, tile = w & 3;
        y = mat >> 1; l = 1 + (mat & 1); i0 = (tile >> 1) * 128; j0 = (tile & 1) * 128;
    } else {
        y = (w - 16) >> 1; l = 0; i0 = ((w - 16) & 1) * 128; j0 = 0;
    }
    const int nrb = (int)((M + 31) / 32);
    const size_t img = (size_t)16 * nrb * 512;
    const int Kin = l == 0 ? lay.obs_dim : HID;
    const __bf16* A = reinterpret_cast<const __bf16*>(a.dz) + (size_t)(y * 3 + l) * NPL * img;
    const __bf16* Bm = l == 0 ? st_tiles : reinterpret_cast<const __bf16*>(a.h) + (size_t)(y * 2 + l - 1) * NPL * img;
    const size_t b_plane = l == 0 ? img / 4 : img;
    const int ncb_b = l == 0 ? 4 : 16;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    const int rb0 = (int)((long long)s * a.rows_per_slab / 32);
    const int rb1 = min(nrb, (int)(((long long)(s + 1) * a.rows_per_slab) / 32));
    const int nkb = max(0, rb1 - rb0);
    constexpr int AHEAD = DWX_NBUF - 1;

    if (wave >= 4) {  // ---- loader waves: wave 4 the 8 x 3 dZ^T tiles of every k-block (LDS slots p * 16 + 0..7), wave 5 the H^T ones (+ 8)
        const int op = wave - 4;
        auto issue = [&](int kb, int buf) {
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const __bf16* src = (op == 0 ? A + p * img + ((size_t)(i0 / 16 + t) * nrb + rb0) * 512
                                                 : Bm + p * b_plane + ((size_t)min(j0 / 16 + t, ncb_b - 1) * nrb + rb0) * 512) + lane * 8;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)kb * 512),
                                                     (__attribute__((address_space(3))) void*)(dsm + buf * DWX_BUF_BYTES + (p * 16 + 8 * op + t) * 1024), 16, 0, 0);
                }
        };
        int fill = 0;
        for (int kb = 0; kb < AHEAD && kb < nkb; ++kb) {
            issue(kb, fill);
            fill = fill + 1 == DWX_NBUF ? 0 : fill + 1;
        }
        for (int kb = 0; kb < nkb; ++kb) {
            // this wave's 24 copies of k-block kb have landed once only those of the k-block behind it are still in flight
            if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // hand-off kb: k-block kb has landed, k-block kb - 1 has been read by everybody
            asm volatile("" ::: "memory");
            if (kb + AHEAD < nkb) {
                issue(kb + AHEAD, fill);
                fill = fill + 1 == DWX_NBUF ? 0 : fill + 1;
            }
        }
        return;
    }

    // ---- compute waves ---------------------------------------------------------------------------------------------------
    const int wi = wave >> 1, wj = wave & 1;
    const bool live = j0 + wj * 64 < Kin;  // first layers: only the first 64-column block holds inputs (the wave still synchronises)
    f32x4 acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; ++kb) {
        dwx_barrier();  // hand-off kb
        const char* buf = dsm + (kb % DWX_NBUF) * DWX_BUF_BYTES + lane * 16;
        bf16x8 fb[4][NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[t][p] = *reinterpret_cast<const bf16x8*>(buf + (p * 16 + 8 + wj * 4 + t) * 1024);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            bf16x8 fa[NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) fa[p] = *reinterpret_cast<const bf16x8*>(buf + (p * 16 + wi * 4 + ti) * 1024);
            if (live) {
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA(q)], fb[tj][PB(q)], acc[ti][tj], 0, 0, 0);
                float t8 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) t8 += ((float)fa[0][e] + (float)fa[1][e]) + (float)fa[2][e];
                bsum[ti] += t8;
            }
        }
    }
    dwx_barrier();  // every compute wave has read its last fragments (the loaders are gone): the ring becomes the store staging area
    float* slab = a.grads + (size_t)s * lay.n_params;
    float* dW = slab + lay.off_w[y][l];
    const bool vec_ok = (lay.n_params & 3) == 0 && (reinterpret_cast<uintptr_t>(a.grads) & 15) == 0;
    if (live && l != 0 && vec_ok) {
        constexpr int SS = 68;
        float* stage = reinterpret_cast<float*>(dsm) + wave * 32 * SS;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stage[(t2 * 16 + 4 * kq + r) * SS + tj * 16 + r16] = acc[2 * hp + t2][tj][r];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = lane + 64 * u, row = idx >> 4, c4 = idx & 15;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * SS + 4 * c4);
                *reinterpret_cast<f32x4*>(dW + (size_t)(i0 + wi * 64 + hp * 32 + row) * HID + j0 + wj * 64 + 4 * c4) = v;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
    } else if (live) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                const int col = j0 + wj * 64 + tj * 16 + r16;
                if (col < Kin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i0 + wi * 64 + ti * 16 + 4 * kq + r;
                        dW[(size_t)row * Kin + col] = acc[ti][tj][r];
                    }
                }
            }
    }
    if (live && j0 == 0 && wj == 0) {  // bias gradient = column sums of dZ: lane (r16, kq) holds 8 of the 32 rows of column ti*16 + r16
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            float tot = bsum[ti];
            tot += __shfl_xor(tot, 16, 64);
            tot += __shfl_xor(tot, 32, 64);
            if (kq == 0) slab[lay.off_b[y][l] + i0 + wi * 64 + ti * 16 + r16] = tot;
        }
    }
}

template <typename K>
int set_lds_x(K kern, size_t bytes) {
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

bool f32_split() {  // read once per process
    static const bool split = !(getenv("RLX_F32_EXACT_MFMA") != nullptr && atoi(getenv("RLX_F32_EXACT_MFMA")) != 0);
    return split;
}

size_t f32x_tiles_bytes() { return NPL * TILE_PLANE * sizeof(__bf16); }

int pack_tiles_f32x(const float* params, const rlx_mlp_layout& lay, void* tiles, hipStream_t st) {
    hipLaunchKernelGGL(pack_tiles_f32x_kernel, dim3(num_cu() * 4), dim3(256), 0, st, params, lay, static_cast<__bf16*>(tiles));
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int launch_rollout_f32x(const RolloutArgs& a, int blocks, hipStream_t st) {
    const size_t lds = GeoX<1, 8>::LDS_BYTES;
#ifdef RLX_DEV_VARIANTS
    if (dev_variant("RLX_F32X_ROLLOUT_PD", 2) == 4) {
        if (int rc = set_lds_x(rollout_step_f32x_kernel<4>, lds)) return rc;
        hipLaunchKernelGGL(rollout_step_f32x_kernel<4>, dim3(blocks), dim3(512), lds, st, a);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
    if (dev_variant("RLX_F32X_ROLLOUT_PD", 2) == 8) {
        if (int rc = set_lds_x(rollout_step_f32x_kernel<8>, lds)) return rc;
        hipLaunchKernelGGL(rollout_step_f32x_kernel<8>, dim3(blocks), dim3(512), lds, st, a);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
#endif
    if (int rc = set_lds_x(rollout_step_f32x_kernel<2>, lds)) return rc;
    hipLaunchKernelGGL(rollout_step_f32x_kernel<2>, dim3(blocks), dim3(512), lds, st, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int launch_step_f32x(const StepArgs& a, const DwArgs& d, void* st_tiles, int tiles, int dw_blocks, hipStream_t st) {
    __bf16* stt = static_cast<__bf16*>(st_tiles);
#define RLX_F32X_LAUNCH(RTV, PDV, DECV, STV)                                                                                       \
    {                                                                                                                              \
        const size_t lds = GeoX<RTV, 8>::LDS_BYTES;                                                                                \
        if (int rc = set_lds_x(ppo_step_fused_f32x_kernel<RTV, 8, PDV, DECV, STV>, lds)) return rc;                                \
        hipLaunchKernelGGL((ppo_step_fused_f32x_kernel<RTV, 8, PDV, DECV, STV>), dim3(tiles, 2), dim3(512), lds, st, a, stt);     \
    }
    // `tiles` counts f32x_bm()-row tiles (plan_step).  Measured on one box (profiles/r05_f32x_*): 64-row tiles (half the weight bytes
    // per row, one round of workgroups) 82.2 us against 82.7 us for 32-row tiles; weight-ring depth 1 / 2 / 3: 79.7 / 79.9 / 81.2 us
    // -- the launch is bound by neither.  32 rows, depth 2 it is; the rest is compiled with -DRLX_DEV_VARIANTS only.
#ifdef RLX_DEV_VARIANTS
    if (a.stamps != nullptr && !a.dec.on) {  // phase stamps (tools/phase_times.py)
        if (f32x_bm() == 64) RLX_F32X_LAUNCH(4, 1, false, true)
        else RLX_F32X_LAUNCH(2, 2, false, true)
    } else if (f32x_bm() == 64) {
        if (a.dec.on) RLX_F32X_LAUNCH(4, 1, true, false)
        else RLX_F32X_LAUNCH(4, 1, false, false)
    } else
#endif
    {
        if (a.dec.on) RLX_F32X_LAUNCH(2, 2, true, false)
        else RLX_F32X_LAUNCH(2, 2, false, false)
    }
#undef RLX_F32X_LAUNCH
    RLX_LAUNCH_CHECK();
    const size_t rlds = (size_t)DWX_NBUF * DWX_BUF_BYTES;
    if (int rc = set_lds_x(ppo_step_dw_f32x_kernel, rlds)) return rc;
    hipLaunchKernelGGL(ppo_step_dw_f32x_kernel, dim3(dw_blocks), dim3(DWX_THREADS), rlds, st, d, static_cast<const __bf16*>(stt));
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // namespace step
}  // namespace rlx
