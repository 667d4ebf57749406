// ppo_step_bf16.hip -- the fused rollout / optimizer-step launches with bf16 MFMA operands (f32 accumulate), gfx950.
//
// "PPO bf16" (BASELINE.json configs[1]; the reference's amp_autocast / precision: bf16 switch, rlinf/hybrid_engines/fsdp/
// fsdp_model_manager.py:122-142, examples/embodiment/config/maniskill_ppo_mlp.yaml:98,128-131): master weights, biases,
// heads, log-probs, losses, GAE and the optimizer stay f32; only the operands of the 256-wide dense layers are bf16,
// accumulated in f32 by v_mfma_f32_16x16x32_bf16 (one instruction per 16x16 tile and 32 k: 1/16 of the f32 MFMA time).
// Same decomposition as ppo_step.hip; what changes with the element size:
//   * the activation slab lives in LDS as bf16 (row stride 272 elements: conflict-free 16-byte fragment reads);
//   * weight tiles are 16 (n) x 32 (k) bf16 = 1 KiB, a fragment load is ONE coalesced b128 per lane;
//   * activations / pre-activation gradients leave the fused kernel as bf16 in a k-tiled TRANSPOSED layout
//     (16 columns x 32 batch rows per 1 KiB tile, the weight-gradient GEMM's fragment order), written straight from the
//     accumulator registers with 512-byte coalesced stores -- the weight-gradient kernel then streams both operands
//     L2 -> registers with no LDS and no barrier at all.

#include "ppo_step_bf16_parts.h"

namespace rlx {
namespace {

using namespace loss;
using namespace step;
using namespace b16;


template <int RT, int NW>
struct GeoB {
    static constexpr int BM = 16 * RT, NT = 64 * NW, CT = HID / (16 * NW);
    static constexpr int SLAB_FLOATS = BM * XSB / 2;  // the bf16 slab measured in floats
    // head image + bias | sHead, sLp, sG, sD | sOld, sAct, sAdv, sRet (the tile's loss inputs, staged at kernel start)
    // | the three hidden layers' biases [3][HID] | std, var, log(std) of the policy [3][MAX_OUT] (+ pad)
    static constexpr int BIAS_OFF = MAX_OUT * W4S + MAX_OUT + 8 * BM * MAX_OUT;  // floats past the slab
    static constexpr int AUX_FLOATS = BIAS_OFF + 3 * HID + 4 * MAX_OUT;
    static constexpr size_t LDS_BYTES = (size_t)(SLAB_FLOATS + AUX_FLOATS) * sizeof(float) + 4096;
    // fused step only: the rounded activations h1, h2 in the accumulator layout, one 8-byte word per lane and 16x16 tile
    // (lane-linear: conflict-free b64 accesses), read back by the backward epilogues -- 32 VGPRs that no longer pin the kernel
    // at the register limit (the spills they caused were reloaded with vmcnt(0) waits, i.e. behind the weight prefetch)
    static constexpr size_t KEEP_BYTES = RT == 4 ? (size_t)2 * BM * HID * sizeof(__bf16) : 0;  // RT == 2: registers (16 VGPRs)
};

// bf16 weight tiles: same matrix list and element offsets as struct Tiles, tile = 16 (n) x 32 (k), K-STEP MAJOR:
//   element ((it * 16 + nb) * 64 + l) * 8 + j  holds  W[nb*16 + (l & 15)][it*32 + 8*(l >> 4) + j]
// (round 4; before: column-block major, (nb * nit + it).  The 16 tiles of one k-step -- what a workgroup consumes together -- are
//  now 16 KiB of consecutive addresses: column-block major put them 2 / 8 KiB apart, i.e. onto a quarter of an XCD's L2 channels
//  at a time while every CU of the XCD asks for the same k-step.)
__global__ __launch_bounds__(256) void pack_tiles_bf16_kernel(const float* __restrict__ params, rlx_mlp_layout lay,
                                                              __bf16* __restrict__ tiles) {
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
        const int j = (int)(r & 7), l = (int)((r >> 3) & 63);
        const int q = (int)(r >> 9), nb = q % (HID / 16), it = q / (HID / 16);
        const int n = nb * 16 + (l & 15), k = it * 32 + 8 * (l >> 4) + j;
        float v;
        if (m == 0) v = k < lay.obs_dim ? params[lay.off_w[y][0] + (size_t)n * lay.obs_dim + k] : 0.f;
        else if (m <= 2) v = params[lay.off_w[y][m] + (size_t)n * HID + k];
        else v = params[lay.off_w[y][m - 2] + (size_t)k * HID + n];
        tiles[i] = (__bf16)v;
    }
}

template <int RT, int CT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[RT][CT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc = Xb[0:BM, 0:32*nit] . W^T, weights as bf16 fragment tiles streamed L2 -> registers (ring of PD stages), the A operand
// from the bf16 LDS slab; no barrier inside the loop, one at the end.
template <int RT, int NW, int PD>
struct RowGemmB {
    typedef GeoB<RT, NW> G;
    static constexpr int CT = G::CT, KI = 32, MAXIT = HID / KI;
    bf16x8 bq[PD][CT];
    const char* wmat;   // the layer's image (uniform: a scalar register pair)
    unsigned woff;      // this lane's byte offset into it
    int nit;

    // (Rotating the k-tile order per workgroup, to spread identical requests over the L2 channels, was measured: no gain,
    //  and it makes results depend on the block index -- dropped.)
    // Addresses as uniform base + 32-bit lane offset: the global_load takes its base from scalar registers and a 4-byte offset per
    // lane instead of a 64-bit address per lane.
    __device__ __forceinline__ void gload(int it, bf16x8 (&b)[CT]) {
        const char* stage = wmat + (size_t)it * ((HID / 16) * 512 * sizeof(__bf16));  // (scalar arithmetic: the stage's base)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[ct] = *reinterpret_cast<const bf16x8*>(stage + ct * 512 * sizeof(__bf16) + woff);
    }
    __device__ __forceinline__ void prefetch(const __bf16* __restrict__ P, int nit_) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        nit = nit_;
        wmat = reinterpret_cast<const char*>(P);
        woff = (unsigned)((wave * CT * 512 + lane * 8) * (int)sizeof(__bf16));
#pragma unroll
        for (int d = 0; d < PD; ++d)
            if (d < nit) gload(d, bq[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // stages [D0, D1) of a layer whose PD covers all of its k-steps: a launch may spread its requests over several points
    template <int D0, int D1>
    __device__ __forceinline__ void prefetch_part(const __bf16* __restrict__ P, int nit_) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        nit = nit_;
        wmat = reinterpret_cast<const char*>(P);
        woff = (unsigned)((wave * CT * 512 + lane * 8) * (int)sizeof(__bf16));
#pragma unroll
        for (int d = D0; d < D1; ++d)
            if (d < nit) gload(d, bq[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void run(const __bf16* Xb, f32x4 (&acc)[RT][CT]) {
        const int lane = threadIdx.x & 63, r16 = lane & 15, kb = lane >> 4;
        zero_acc(acc);
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            if (it >= nit) break;
            bf16x8 a[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                a[rt] = *reinterpret_cast<const bf16x8*>(Xb + (rt * 16 + r16) * XSB + it * KI + 8 * kb);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt], bq[it % PD][ct], acc[rt][ct], 0, 0, 0);
            if (it + PD < nit) {
                gload(it + PD, bq[it % PD]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();
    }
};

// k-tiled transposed bf16 image of a [rows][256] matrix (the weight-gradient GEMM's operand layout):
//   tile (cb = col / 16, rb = row / 32) is 1 KiB at (cb * nrb + rb) * 512 elements; inside it element
//   ((row % 32) / 8 * 16 + col % 16) * 8 + row % 8.
// The accumulator layout holds 4 consecutive rows of one column per register quad -> 8 contiguous bytes; the four
// row-quads x sixteen columns of a wave instruction land in one contiguous 512-byte run.
template <int RT, int NW>
__device__ __forceinline__ void store_tiles(const bf16x4 (*v)[GeoB<RT, NW>::CT], __bf16* __restrict__ dst, int nrb, long long m0) {
    constexpr int CT = GeoB<RT, NW>::CT;
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

// obs-preprocess into the bf16 slab (k tail and rows past M zero); states_copy: f32 copy (trajectory buffer row);
// st_tiles: k-tiled transposed bf16 image of the states (B operand of the first layers' weight gradients), 4 column blocks.
// Split into issue / commit: a wave's loads retire in order (vmcnt), so WAITING for a load also waits for every load issued
// before it.  The kernels therefore request the tile's states FIRST, then every other global input of the launch, and only
// then commit the states to LDS -- one memory round trip for everything instead of one per input.
template <int RT, int NW>
struct StatesB {
    typedef GeoB<RT, NW> G;
    static_assert(Tiles::K1P == 64, "one pass per lane assumes a 64-wide padded first layer");
    // The tile's rows are CONTIGUOUS in memory (BM x D floats, starting on a multiple of 16 floats), so it travels as float4s
    // (dword alignment is all a global 16-byte load needs): one instruction for each of the first few waves, where round 5's
    // form issued BM * 64 / NT dword instructions in EVERY wave -- a vector-memory instruction costs the CU's address unit ~16
    // cycles whatever its width, and the launch's first wait sits behind all of them (32 wave-instructions per workgroup -> 6
    // at 32 x 42).  Floats past the end of the array are never read: the (at most three) valid floats behind the last whole
    // float4 of the array's last tile are fetched by three lanes in commit().
    static constexpr int UV = (G::BM * (Tiles::K1P / 4) + G::NT - 1) / G::NT;
    f32x4 xv[UV];
    __device__ __forceinline__ static int valid_floats(int D, long long m0, long long M) {  // of this tile
        return (int)(min((long long)G::BM, M - m0) * D);
    }
    __device__ __forceinline__ void issue(const float* __restrict__ states, int D, long long m0, long long M) {
        const int nfloat = valid_floats(D, m0, M);
        const int nfull4 = nfloat >> 2, wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x) & ~63;  // (scalar: see SmallInputsB)
        const f32x4* t4 = reinterpret_cast<const f32x4*>(states + (size_t)m0 * D);
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            xv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (wave0 + u * G::NT < nfull4) xv[u] = t4[min((int)threadIdx.x + u * G::NT, nfull4 - 1)];  // (wave-uniform skip)
        }
    }
    __device__ __forceinline__ void commit(const float* __restrict__ states, float* __restrict__ states_copy, int D, long long m0,
                                           long long M, __bf16* Xb) const {
        const int kp = round_up(D, KPAD), nfloat = valid_floats(D, m0, M), nfull4 = nfloat >> 2, rem = nfloat & 3;
        const float inv_d = 1.f / (float)D;
        auto row_of = [&](int e) {  // e / D for an element index e < 4096: the float quotient is off by far less than one
            int r = (int)(((float)e + 0.5f) * inv_d);
            r -= (r * D > e);
            r += ((r + 1) * D <= e);
            return r;
        };
        float* copy = states_copy ? states_copy + (size_t)m0 * D : nullptr;
#pragma unroll
        for (int u = 0; u < UV; ++u) {
            const int q = (int)threadIdx.x + u * G::NT;
            if (q >= G::BM * D / 4) continue;
            const bool ok = q < nfull4;
            if (ok && copy) *reinterpret_cast<f32x4*>(copy + 4 * q) = xv[u];
            int r = row_of(4 * q), c = 4 * q - r * D;  // element 4 q + j sits at (r, c): one division, then a carry per element
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ok || 4 * q + j >= nfloat) Xb[r * XSB + c] = (__bf16)(ok ? xv[u][j] : 0.f);  // (rows past M: zeros)
                if (++c == D) c = 0, ++r;
            }
        }
        if ((int)threadIdx.x < rem) {  // the floats behind the array's last whole float4: fetched here (a second round trip, paid by
            const int e = 4 * nfull4 + threadIdx.x, r = row_of(e);  // the last tile of an array whose size is no multiple of four floats)
            const float tail = states[(size_t)m0 * D + e];
            Xb[r * XSB + (e - r * D)] = (__bf16)tail;
            if (copy) copy[e] = tail;
        }
        const int padc = kp - D;  // the zero columns behind every row (the k tail of the first layer)
        for (int i = threadIdx.x; i < G::BM * padc; i += G::NT) Xb[(i / padc) * XSB + D + i % padc] = (__bf16)0.f;
    }
    // after an LDS barrier behind commit()
    __device__ __forceinline__ static void tiles(__bf16* __restrict__ st_tiles, int nrb, int D, long long m0, const __bf16* Xb) {
        const int kp = round_up(D, KPAD);
        for (int u = threadIdx.x; u < 4 * (G::BM / 8) * 16; u += G::NT) {
            const int cb = u / ((G::BM / 8) * 16), ko = (u / 16) % (G::BM / 8), c16 = u & 15;
            const int rb = (int)(m0 >> 5) + (ko >> 2);
            if (rb >= nrb) continue;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = cb * 16 + c16 < kp ? Xb[(8 * ko + j) * XSB + cb * 16 + c16] : (__bf16)0.f;
            *reinterpret_cast<bf16x8*>(st_tiles + ((size_t)(cb * nrb + rb) * 64 + (ko & 3) * 16 + c16) * 8) = v;
        }
    }
};

// forward hidden-layer epilogue: h = bf16(tanh(acc + bias)) -> slab; KEEP: the rounded value also goes to `kept` (registers) or
// `keep_lds` (lane-linear LDS words) for the backward sweep's 1 - h^2, and to the k-tiled image `dst`.  `bias` is an LDS image.
template <int RT, int NW, bool KEEP>
__device__ __forceinline__ void epilogue_tanh_b(const f32x4 (&acc)[RT][GeoB<RT, NW>::CT], const float* bias, __bf16* Xb,
                                                bf16x4 (*kept)[GeoB<RT, NW>::CT], bf16x4* keep_lds, __bf16* __restrict__ dst, int nrb,
                                                long long m0) {
    typedef GeoB<RT, NW> G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15;
    bf16x4 hloc[KEEP ? RT : 1][G::CT];
    float b[G::CT];
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) b[ct] = bias[wave * 16 * G::CT + ct * 16 + r16];
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            bf16x4 hv;
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 h = tanh2_b(f32x2{acc[rt][ct][r] + b[ct], acc[rt][ct][r + 1] + b[ct]});
                hv[r] = (__bf16)h.x;
                hv[r + 1] = (__bf16)h.y;
            }
            store_slab_quad(Xb, rt * 16, wave * 16 * G::CT + ct * 16, hv);
            if constexpr (KEEP) hloc[rt][ct] = hv;
        }
    }
    if constexpr (KEEP) {
        if (dst != nullptr) store_tiles<RT, NW>(hloc, dst, nrb, m0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct) {
                if (keep_lds != nullptr) keep_lds[(rt * G::CT + ct) * G::NT + threadIdx.x] = hloc[rt][ct];
                if (kept != nullptr) kept[rt][ct] = hloc[rt][ct];
            }
    }
    lds_barrier();
}


// head forward: out[row][o] = h3[row][:] . W4[o][:], written as TWO k-half partials P0 / P1 [BM][MAX_OUT] (the caller adds them
// and the bias: P0 + P1 + b).  Wave w < 2 RT takes row tile w % RT and k half w / RT -- the rollout launch (RT = 1) and the
// training launch (RT = 4) therefore add every row's products in the same order: the log-prob a sample gets at rollout time
// and the one the first training epoch recomputes are bit-identical, as with the reference.  Rows of W4s past n_out are zero.
template <int RT>
__device__ __forceinline__ void head_forward_mfma(const __bf16* Xb, const float* W4s, float* P0, float* P1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
    if (wave >= 2 * RT) return;
    const int rt = wave % RT, kh = wave / RT;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < HID / 64; ++q) {
        const int k0 = (kh * (HID / 64) + q) * 32 + 8 * kq;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Xb + (rt * 16 + r16) * XSB + k0);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(W4s + r16 * W4S + k0), w1 = *reinterpret_cast<const f32x4*>(W4s + r16 * W4S + k0 + 4);
        const float w[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
        acc = mfma3(a, split3(w), acc);
    }
    float* P = kh == 0 ? P0 : P1;
#pragma unroll
    for (int r = 0; r < 4; ++r) P[(rt * 16 + 4 * kq + r) * MAX_OUT + r16] = acc[r];
}

// ---------------------------------------------------------------------------------------------------------------
// rollout step (RT = 1, NW = 8)
// ---------------------------------------------------------------------------------------------------------------
template <int RT, int ORDER = 1345>  // 16 RT rows per workgroup; ORDER: where the weight fragments of layers 2 and 3 are requested
__global__ __launch_bounds__(512) void rollout_step_bf16_kernel(RolloutArgs a) {
    constexpr int NW = 8;
    touch_kernargs<(int)sizeof(RolloutArgs)>();
    typedef GeoB<RT, NW> G;
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

    // Every global input of the tile is requested before anything waits: states, biases / head / log-std, the noise of this
    // lane's output element, and ALL three layers' weight fragments (16 rows per workgroup leave the registers for it:
    // 16 + 64 + 64 VGPRs).  The launch then costs one memory round trip plus the compute chain, instead of one round trip per
    // layer (each layer's first fragment loads used to be waited for by the previous epilogue's bias load).
#ifdef RLX_DEV_VARIANTS
    Stamps ts{a.stamps, 0};  // development: phase stamps of block 0 (tools/phase_times.py)
#define RLX_MARK() ts.mark()
#else
#define RLX_MARK()
#endif
    RLX_MARK();
#ifdef RLX_DEV_VARIANTS
    if (a.stamps != nullptr) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // kernel arguments arrived
#endif
    RLX_MARK();
    StatesB<RT, NW> st;
    st.issue(states, D, m0, M);
    SmallInputsB<G::NT> si;
    si.issue(a.params, lay, y, n_out);
    const int orow = tid / n_out, oo = tid % n_out;
    const bool olive = tid < G::BM * n_out && m0 + orow < M;
    const size_t og = (size_t)min(m0 + orow, M - 1) * n_out + oo;
    const float epsv = (a.eps != nullptr ? a.eps : a.params)[(job == 0 && y == 1 && a.eps != nullptr && tid < G::BM * n_out) ? og : 0];
    const ValuePre vp = value_job_prefetch(a.vj[max(job - 1, 0)], job != 0, (size_t)min(m0 + orow, M - 1), a.params);
    __builtin_amdgcn_sched_barrier(0);
    RowGemmB<RT, NW, Tiles::K1P / 32> gemm1;
    RowGemmB<RT, NW, HID / 32> gemm2, gemm3;
    // WHEN the weight fragments are requested decides this launch.  A workgroup's 288 KB of fragments pass the CU's address
    // unit at 64 B / clk: 4.6 k cycles, and a wave that issues loads faster than that STALLS in issue -- with everything requested
    // up front (rounds 3-5: "one memory round trip per launch") the waves spent 3.9 k cycles issuing and the first barrier fell at
    // 7.8 k of the launch's 13.7 k (phase stamps, development build), the products and epilogues (5.9 k) queueing behind it.
    // Requested one phase ahead of their use, half a layer at a time, the same bytes stream while the earlier layers compute.
    // ORDER = four decimal digits (W2 first half, W2 second half, W3 first half, W3 second half): the point at which that half
    // layer is requested -- 0 up front, 1 behind the commits, 2 behind the first barrier, 3 behind the first layer's products,
    // 4 behind its epilogue, 5 behind the second layer's products.  Same loads, same arithmetic: bit-identical outputs.
    constexpr int P2A = ORDER / 1000 % 10, P2B = ORDER / 100 % 10, P3A = ORDER / 10 % 10, P3B = ORDER % 10, H = HID / 64;
#define RLX_POINT(N)                                                                               \
    {                                                                                              \
        if constexpr (P2A == N) gemm2.template prefetch_part<0, H>(tiles + Tiles::mat(y, 1), HID / 32);      \
        if constexpr (P2B == N) gemm2.template prefetch_part<H, 2 * H>(tiles + Tiles::mat(y, 1), HID / 32);  \
        if constexpr (P3A == N) gemm3.template prefetch_part<0, H>(tiles + Tiles::mat(y, 2), HID / 32);      \
        if constexpr (P3B == N) gemm3.template prefetch_part<H, 2 * H>(tiles + Tiles::mat(y, 2), HID / 32);  \
    }
    gemm1.prefetch(tiles + Tiles::mat(y, 0), Tiles::K1P / 32);
    RLX_POINT(0)
    RLX_MARK();  // every load issued
    st.commit(states, states_copy, D, m0, M, Xb);
    RLX_MARK();  // states arrived
    si.commit(n_out, sBias, W4s, b4s, sStd);
    for (int i = n_out * W4S + tid; i < MAX_OUT * W4S; i += G::NT) W4s[i] = 0.f;
    RLX_POINT(1)
    lds_barrier();
    RLX_MARK();  // inputs staged
    RLX_POINT(2)
    f32x4 acc[RT][G::CT];
    gemm1.run(Xb, acc);
    RLX_POINT(3)
    RLX_MARK();
    epilogue_tanh_b<RT, NW, false>(acc, sBias, Xb, nullptr, nullptr, nullptr, 0, m0);
    RLX_POINT(4)
    RLX_MARK();
    gemm2.run(Xb, acc);
    RLX_POINT(5)
#undef RLX_POINT
    RLX_MARK();
    epilogue_tanh_b<RT, NW, false>(acc, sBias + HID, Xb, nullptr, nullptr, nullptr, 0, m0);
    RLX_MARK();
    gemm3.run(Xb, acc);
    RLX_MARK();
    epilogue_tanh_b<RT, NW, false>(acc, sBias + 2 * HID, Xb, nullptr, nullptr, nullptr, 0, m0);
    RLX_MARK();

    static_assert(G::BM * MAX_OUT <= G::NT, "one head output per lane");
    float* sP0 = b4s + MAX_OUT, *sP1 = sP0 + G::BM * MAX_OUT;
    head_forward_mfma<RT>(Xb, W4s, sP0, sP1);
    lds_barrier();
    RLX_MARK();  // head
#undef RLX_MARK
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
// fused optimizer-step kernel (1), bf16: StepArgs.h / .dz point at the k-tiled transposed bf16 images
//   h  : [2 nets][2][16 cb][nrb] tiles    dz : [2 nets][3][16 cb][nrb] tiles    st : [4 cb][nrb] tiles (states)
// ---------------------------------------------------------------------------------------------------------------

template <int RT, int NW, int PD, int OP, bool STAMPS, bool DEC = false>  // DEC: the decoupled actor loss (StepArgs.dec)
__global__ __launch_bounds__(64 * NW, RT == 2 ? 4 : 2) void ppo_step_fused_bf16_kernel(StepArgs a, __bf16* st_tiles) {
    typedef GeoB<RT, NW> G;
    constexpr int BM = G::BM, CT = G::CT;
    extern __shared__ __align__(16) float smem[];
    __bf16* Xb = reinterpret_cast<__bf16*>(smem);
    float* W4s = smem + G::SLAB_FLOATS;
    float* b4s = W4s + MAX_OUT * W4S;
    float* sHead = b4s + MAX_OUT;
    float* sLp = sHead + BM * MAX_OUT;
    float* sG = sLp + BM * MAX_OUT;
    float* sD = sG + BM * MAX_OUT;
    // the tile's loss inputs, requested before the first GEMM so that their (memory-side) latency hides behind the forward
    // sweeps: old log-probs and actions [BM][act_dim]; advantages [BM][npr] (actor) or prev_values [BM][n_out] (critic); returns
    float* sOld = sD + BM * MAX_OUT;
    float* sAct = sOld + BM * MAX_OUT;
    float* sAdv = sAct + BM * MAX_OUT;
    float* sRet = sAdv + BM * MAX_OUT;
    double* sRed = reinterpret_cast<double*>(smem + G::SLAB_FLOATS + G::AUX_FLOATS);
    double* sNm = sRed + 256;
    bf16x4* sKeep = reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(smem) + G::LDS_BYTES);  // [2][RT * CT][NT] words

    const rlx_mlp_layout& lay = a.lay;
    const rlx_ppo_loss_params& p = a.p;
    // Workgroups go to the 8 XCDs round-robin (linear id % 8 = blockIdx.x % 8: gridDim.x is a multiple of 8 whenever the remap is
    // used).  The weight-gradient launch behind this one gives XCD x the split-K slabs over rows [x M / 8, (x + 1) M / 8); with the
    // identity map those rows were written by ALL XCDs and every operand fetch of that launch is a memory-side trip.  Remapped
    // (development: RLX_FUSED_XCD_ROWS=1), XCD x writes exactly the rows it reads back next -- measured, same box, 3200 launches
    // each: weight-gradient launch 15.88 -> 15.64 us, this launch 25.16 -> 25.44 us (profiles/r03_xcd_row_map_*.txt).  A launch
    // boundary writes back AND invalidates the XCD's L2, so which XCD produced a line does not matter to the next launch; only
    // producer and consumer inside ONE launch could share an L2.  The identity map stays.
    const int nt = gridDim.x, bx = blockIdx.x;
    const int y = blockIdx.y, tile = (a.xcd_rows && (nt & 7) == 0) ? (bx & 7) * (nt >> 3) + (bx >> 3) : bx, D = lay.obs_dim, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const long long m0 = (long long)tile * BM, M = a.M;
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const int npr = lay.act_dim / p.raw_per_adv;
    const long long n_adv = M * npr;
    const bool has_mask = a.loss_mask != nullptr;
    const TileGeom tg{(int)((M + 31) / 32)};
    const __bf16* tiles = reinterpret_cast<const __bf16*>(a.tiles);
    __bf16* hy = reinterpret_cast<__bf16*>(a.h) + (size_t)(y * 2) * tg.mat();
    __bf16* dzy = reinterpret_cast<__bf16*>(a.dz) + (size_t)(y * 3) * tg.mat();

    if (has_mask) {
        double cnt[1] = {0.0};
        for (long long e = tid; e < n_adv; e += G::NT) cnt[0] += a.loss_mask[e] != 0 ? 1.0 : 0.0;
        block_sum<1>(cnt, sRed);
        if (tid == 0) sNm[0] = cnt[0];
    }

    // ---- forward -----------------------------------------------------------------------------------------------------
    StampsT<STAMPS> ts{a.stamps, 0};  // development: phase stamps of block (0, 0), see tools/phase_times.py
    ts.mark();
    float* sBias = smem + G::SLAB_FLOATS + G::BIAS_OFF;
    float* sStd = sBias + 3 * HID;
    // Request order = wait order (a wave's loads retire in order): the tile's states first, then the loss inputs, the biases /
    // head image / log-std, then the first layer's weight fragments.  Nothing below the commits touches global memory except the
    // weight stream, so no epilogue waits behind a weight prefetch any more (the per-layer bias load used to: it was issued
    // after the next layer's first fragment loads and its wait covered their whole first-touch latency).
    StatesB<RT, NW> st;
    st.issue(a.states, D, m0, M);
    constexpr int PI = BM * MAX_OUT / G::NT;  // staging iterations per lane (2)
    static_assert(BM * MAX_OUT % G::NT == 0, "staging assumes whole iterations");
    float v0[PI], v1[PI], v2[PI];
#pragma unroll
    for (int u = 0; u < PI; ++u) {
        const int i = tid + u * G::NT, row = i / MAX_OUT, c = i % MAX_OUT;
        const size_t gr = (size_t)min(m0 + row, M - 1);
        if (y == 1) {  // clamped, unconditional loads; junk columns are never read back
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
    RowGemmB<RT, NW, PD> gemm;
    gemm.prefetch(tiles + Tiles::mat(y, 0), Tiles::K1P / 32);
    ts.mark();  // (every input requested)
    st.commit(a.states, nullptr, D, m0, M, Xb);
    ts.mark();
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
    if (y == 1) StatesB<RT, NW>::tiles(st_tiles, tg.nrb, D, m0, Xb);
    ts.mark();
    f32x4 acc[RT][CT];
    bf16x4 kept3[RT][CT];  // rounded h3 in the accumulator layout (h1, h2: sKeep): the backward epilogues need exactly these lanes
    constexpr bool KEEP_LDS = RT == 4;
    bf16x4 kept12[KEEP_LDS ? 1 : 2][RT][CT];  // the 32-row variant has the registers for h1, h2 as well
    gemm.run(Xb, acc);
    gemm.prefetch(tiles + Tiles::mat(y, 1), HID / 32);
    ts.mark();
    epilogue_tanh_b<RT, NW, true>(acc, sBias, Xb, KEEP_LDS ? nullptr : kept12[0], KEEP_LDS ? sKeep : nullptr, hy, tg.nrb, m0);
    ts.mark();
    gemm.run(Xb, acc);
    gemm.prefetch(tiles + Tiles::mat(y, 2), HID / 32);
    ts.mark();
    epilogue_tanh_b<RT, NW, true>(acc, sBias + HID, Xb, KEEP_LDS ? nullptr : kept12[KEEP_LDS ? 0 : 1], KEEP_LDS ? sKeep + RT * CT * G::NT : nullptr,
                                  hy + tg.mat(), tg.nrb, m0);
    ts.mark();
    gemm.run(Xb, acc);
    ts.mark();
    epilogue_tanh_b<RT, NW, true>(acc, sBias + 2 * HID, Xb, kept3, nullptr, nullptr, tg.nrb, m0);
    ts.mark();

    // ---- head + loss element math (f32, identical to ppo_step.hip): fused_loss_pass, ppo_step_bf16_parts.h -------------------------
    head_forward_mfma<RT>(Xb, W4s, sG, sLp);  // the two k-half partials land in sG / sLp (both free until the passes below)
    lds_barrier();
    ts.mark();
    fused_loss_pass<BM, NW, DEC>(a, y, tile, m0, LossLds{b4s, sHead, sLp, sG, sD, sOld, sAct, sAdv, sRet, sStd, reinterpret_cast<double*>(Xb), sNm}, ts);
    // ---- head parameter gradients per 32-row half tile: dW4[o][j] = sum_rows dOut[row][o] h3[row][j] on the matrix pipe -------
    // M = o, N = j (this wave's 2 x 16 columns), K = the half's 32 rows.  K is only a summation index, so the k slots are
    // mapped to rows the way the accumulator layout already holds h3: lane (r16, kq) slot jj <-> row (jj >> 2) * 16 + 4 kq +
    // (jj & 3) of the half -- the B fragment is then kept3[2 sub][ct] | kept3[2 sub + 1][ct] straight from registers, and the
    // A fragment (dOut, f32 in sHead) is split into three bf16 planes (exact, see split3).
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
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                bf16x8 B;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) B[jj] = kept3[2 * sub + (jj >> 2)][ct][jj & 3];
                const f32x4 g = mfma3(A, B, zero4);
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
    // ---- dZ3 = (dOut . W4) * (1 - h3^2) -> bf16 slab + tiles: K = n_out is tiny, f32 operands -> v_mfma_f32_16x16x4_f32 ------
    gemm.prefetch(tiles + Tiles::mat(y, 4), HID / 32);  // W3^T
    bf16x4 dv[RT][CT];
    {
        f32x4 dz[RT][CT];
        zero_acc(dz);
        for (int ks = 0; ks < (n_out + 3) / 4; ++ks) {
            float av[RT], bv[CT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) av[rt] = sHead[(rt * 16 + r16) * MAX_OUT + 4 * ks + kq];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bv[ct] = W4s[(4 * ks + kq) * W4S + wave * 16 * CT + ct * 16 + r16];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) dz[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[ct], dz[rt][ct], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) dv[rt][ct][r] = (__bf16)(dz[rt][ct][r] * dtanh_b(kept3[rt][ct][r]));
    }
    lds_barrier();  // every read of h3 / sHead is done: the slab may be overwritten
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            store_slab_quad(Xb, rt * 16, wave * 16 * CT + ct * 16, dv[rt][ct]);
    store_tiles<RT, NW>(dv, dzy + 2 * tg.mat(), tg.nrb, m0);
    lds_barrier();
    ts.mark();

    // ---- backward-data chain ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int l = 2; l >= 1; --l) {
        gemm.run(Xb, acc);
        if (l == 2) gemm.prefetch(tiles + Tiles::mat(y, 3), HID / 32);  // W2^T
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x4 hk = KEEP_LDS ? sKeep[((l - 1) * RT * CT + rt * CT + ct) * G::NT + tid] : kept12[KEEP_LDS ? 0 : l - 1][rt][ct];
#pragma unroll
                for (int r = 0; r < 4; ++r) dv[rt][ct][r] = (__bf16)(acc[rt][ct][r] * dtanh_b(hk[r]));
                if (l == 2) store_slab_quad(Xb, rt * 16, wave * 16 * CT + ct * 16, dv[rt][ct]);
            }
        store_tiles<RT, NW>(dv, dzy + (size_t)(l - 1) * tg.mat(), tg.nrb, m0);
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
// fused optimizer-step kernel (2), bf16: weight gradients from the k-tiled transposed images, barrier-free.
//   item = (slab, matrix, 128x128 tile); 4 waves as 2x2, each 4x4 tiles of 16x16 (v_mfma_f32_16x16x32_bf16, K = 32 batch rows)
// ---------------------------------------------------------------------------------------------------------------
#ifdef RLX_DEV_VARIANTS  // the register-streaming weight-gradient kernel of round 2 (RLX_DW_BF16_REG=1): kept for comparison only
template <int PD>
__global__ __launch_bounds__(256, 2) void ppo_step_dw_bf16_kernel(DwArgs a, const __bf16* __restrict__ st_tiles) {
    __shared__ double s_red[NS * 4];
    const rlx_mlp_layout& lay = a.lay;
    const long long M = a.M;
    const int tid = threadIdx.x;
    const int gemm_blocks = round_up(a.gemm_items, 8);
    int b = blockIdx.x;

    if (b >= gemm_blocks) {
        b -= gemm_blocks;
        if (b < a.slabs * 2) {
            head_reduce_block(a, b >> 1, b & 1, tid);
        } else {
            metric_block(a, s_red, tid, 256);
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
    const size_t mat_elems = (size_t)16 * nrb * 512;
    const int Kin = l == 0 ? lay.obs_dim : HID;
    const __bf16* A = reinterpret_cast<const __bf16*>(a.dz) + (size_t)(y * 3 + l) * mat_elems;
    const __bf16* Bm = l == 0 ? st_tiles : reinterpret_cast<const __bf16*>(a.h) + (size_t)(y * 2 + l - 1) * mat_elems;
    const int ncb_b = l == 0 ? 4 : 16;  // column blocks of the B image
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    if (j0 + wj * 64 >= Kin) return;  // first layers: only the first 64-column block holds inputs
    const int rb0 = (int)((long long)s * a.rows_per_slab / 32);
    const int rb1 = min(nrb, (int)(((long long)(s + 1) * a.rows_per_slab) / 32));
    const int nkb = max(0, rb1 - rb0);

    // fragment pointers: A tile (cb = (i0 + wi*64)/16 + ti, rb), B tile (cb = (j0 + wj*64)/16 + tj clamped, rb)
    const __bf16* ap = A + ((size_t)((i0 + wi * 64) / 16) * nrb + rb0) * 512 + lane * 8;
    int cbj[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) cbj[tj] = min((j0 + wj * 64) / 16 + tj, ncb_b - 1);
    const __bf16* bp = Bm + (size_t)rb0 * 512 + lane * 8;

    struct Frag { bf16x8 a[4], b[4]; };
    Frag ring[PD];
    auto gload = [&](int kb, Frag& f) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f.a[t] = *reinterpret_cast<const bf16x8*>(ap + ((size_t)t * nrb + kb) * 512);
            f.b[t] = *reinterpret_cast<const bf16x8*>(bp + ((size_t)cbj[t] * nrb + kb) * 512);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < nkb) gload(d, ring[d]);
    __builtin_amdgcn_sched_barrier(0);
    for (int kb0 = 0; kb0 < nkb; kb0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int kb = kb0 + d;
            if (kb < nkb) {
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[d].a[ti], ring[d].b[tj], acc[ti][tj], 0, 0, 0);
                    float t8 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t8 += (float)ring[d].a[ti][e];
                    bsum[ti] += t8;
                }
                if (kb + PD < nkb) gload(kb + PD, ring[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float* slab = a.grads + (size_t)s * lay.n_params;
    float* dW = slab + lay.off_w[y][l];
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
    if (j0 == 0 && wj == 0) {  // bias gradient = column sums of dZ: lane (r16, kq) holds 8 of the 32 rows of column ti*16 + r16
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            float tot = bsum[ti];
            tot += __shfl_xor(tot, 16, 64);
            tot += __shfl_xor(tot, 32, 64);
            if (kq == 0) slab[lay.off_b[y][l] + i0 + wi * 64 + ti * 16 + r16] = tot;
        }
    }
}

#endif  // RLX_DEV_VARIANTS

// ---------------------------------------------------------------------------------------------------------------
// weight gradients, LDS-DMA variant: the 16 operand tiles of a k-block (8 of dZ^T, 8 of H^T, 1 KiB each, already in
// lane-linear fragment order) are copied global -> LDS by `global_load_lds_dwordx4` (four 1 KiB DMAs per wave, no
// registers involved), every wave then reads its 4 + 4 fragments with conflict-free ds_read_b128.  Each operand byte
// is fetched ONCE per workgroup (the register-streaming kernel fetches it once per wave pair): the kernel is bound by
// how fast a CU can pull data that the previous launch produced (~11 B/clk), so halving the bytes is what counts.
// Ring of 3 LDS buffers, counted vmcnt waits, raw s_barrier (a __syncthreads() would drain the DMAs).
// ---------------------------------------------------------------------------------------------------------------
constexpr int DW_BUF_BYTES = 16 * 1024;

// s_waitcnt takes an immediate: wait until at most 4 * ahead of this wave's copies (the k-blocks behind the current one) are
// still in flight
template <int N>
__device__ __forceinline__ void wait_copies_ahead(int ahead) {
    if constexpr (N > 0) {
        if (ahead >= N) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * N) : "memory");
            return;
        }
        wait_copies_ahead<N - 1>(ahead);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// DW_NBUF LDS buffers = DW_NBUF - 1 k-blocks in flight per workgroup.  The loop is bound by the LATENCY of the copies (the
// operands were written by the previous launch from every XCD: memory-side trips of ~2.4 k cycles): with 2 k-blocks in flight a
// workgroup moved 16 KB per ~1.2 k cycles (13.7 B/clk) against 256 cycles of MFMA work per k-block.
template <int DW_NBUF>
__global__ __launch_bounds__(256, 2) void ppo_step_dw_bf16_lds_kernel(DwArgs a, const __bf16* __restrict__ st_tiles) {
    extern __shared__ __align__(16) char dsm[];
    const rlx_mlp_layout& lay = a.lay;
    const long long M = a.M;
    const int tid = threadIdx.x;
    const int gemm_blocks = round_up(a.gemm_items, 8);
    int b = blockIdx.x;

    if (b >= gemm_blocks) {
        b -= gemm_blocks;
        if (b < a.slabs * 2) {
            head_reduce_block(a, b >> 1, b & 1, tid);
        } else {
            double* s_red = reinterpret_cast<double*>(dsm);
            metric_block(a, s_red, tid, 256);
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
    const size_t mat_elems = (size_t)16 * nrb * 512;
    const int Kin = l == 0 ? lay.obs_dim : HID;
    const __bf16* A = reinterpret_cast<const __bf16*>(a.dz) + (size_t)(y * 3 + l) * mat_elems;
    const __bf16* Bm = l == 0 ? st_tiles : reinterpret_cast<const __bf16*>(a.h) + (size_t)(y * 2 + l - 1) * mat_elems;
    const int ncb_b = l == 0 ? 4 : 16;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    const int wi = wave >> 1, wj = wave & 1;
    const int rb0 = (int)((long long)s * a.rows_per_slab / 32);
    const int rb1 = min(nrb, (int)(((long long)(s + 1) * a.rows_per_slab) / 32));
    const int nkb = max(0, rb1 - rb0);

    // this wave's DMA duty: A tiles 2w, 2w+1 and B tiles 2w, 2w+1 of every k-block (LDS slots t and 8 + t)
    const __bf16* src[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int t = 2 * wave + q;
        src[q] = A + ((size_t)(i0 / 16 + t) * nrb + rb0) * 512 + lane * 8;
        src[2 + q] = Bm + ((size_t)min(j0 / 16 + t, ncb_b - 1) * nrb + rb0) * 512 + lane * 8;
    }
    auto dma = [&](int kb, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = (q < 2 ? 0 : 8) + 2 * wave + (q & 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[q] + (size_t)kb * 512),
                                             (__attribute__((address_space(3))) void*)(dsm + buf * DW_BUF_BYTES + slot * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool live = j0 + wj * 64 < Kin;  // first layers: only the first 64-column block holds inputs (the wave still copies)

    // stamps by the workgroup that owns GEMM item 0
    long long* stp = (a.stamps != nullptr && item == 0 && tid == 0) ? a.stamps : nullptr;
    if (stp) stp[0] = clock64();
    constexpr int AHEAD = DW_NBUF - 1;  // k-blocks in flight
    for (int rep = 0; rep < a.repeat; ++rep) {  // (1 pass; development: a second, L2-warm pass for timing)
    if (rep > 0) __builtin_amdgcn_s_barrier();  // everybody finished reading the last k-blocks before the ring is refilled
#pragma unroll
    for (int d = 0; d < AHEAD; ++d)
        if (d < nkb) dma(d, d);
    for (int kb = 0; kb < nkb; ++kb) {
        // this wave's copies of k-block kb have landed once only those of the k-blocks behind it are still in flight
        wait_copies_ahead<AHEAD - 1>(min(AHEAD - 1, nkb - 1 - kb));
        __builtin_amdgcn_s_barrier();  // everybody's copies of kb landed; everybody finished reading kb - 1
        asm volatile("" ::: "memory");
        if (stp && kb < 12) stp[1 + kb] = clock64();
        if (kb + AHEAD < nkb) dma(kb + AHEAD, (kb + AHEAD) % DW_NBUF);  // into the buffer k-block kb - 1 just vacated
        const char* buf = dsm + (kb % DW_NBUF) * DW_BUF_BYTES + lane * 16;
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[t] = *reinterpret_cast<const bf16x8*>(buf + (wi * 4 + t) * 1024);
            fb[t] = *reinterpret_cast<const bf16x8*>(buf + (8 + wj * 4 + t) * 1024);
        }
        if (live) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ti], fb[tj], acc[ti][tj], 0, 0, 0);
                float t8 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) t8 += (float)fa[ti][e];
                bsum[ti] += t8;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragment reads are done before this wave arrives at the next barrier
    }
    }
    if (stp) stp[14] = clock64();
    if (!live) return;
    float* slab = a.grads + (size_t)s * lay.n_params;
    float* dW = slab + lay.off_w[y][l];
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
    if (j0 == 0 && wj == 0) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            float tot = bsum[ti];
            tot += __shfl_xor(tot, 16, 64);
            tot += __shfl_xor(tot, 32, 64);
            if (kq == 0) slab[lay.off_b[y][l] + i0 + wi * 64 + ti * 16 + r16] = tot;
        }
    }
    if (stp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[15] = clock64();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradients, RING variant (round 4): the same items, tiles and arithmetic as ppo_step_dw_bf16_lds_kernel -- every MFMA sees
// the same operands in the same order: bit-identical slabs -- restructured around three things measured this round:
//   * A compute wave that issues its own LDS-DMAs stalls ~100-185 cycles per instruction while the LDS serves fragment reads
//     (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"): with four copies per wave and k-block that WAS the ~1.2 k cycles per
//     k-block of the kernel above (256 cycles of MFMA work).  Two LOADER waves (one per operand: the 8 dZ^T tiles, the 8 H^T
//     tiles of every k-block) do nothing but copy, DW_NBUF - 1 k-blocks ahead; their memory queues hold only copies, so their
//     counted vmcnt waits are exact; the four compute waves never wait on vmcnt.
//   * MFMAs on BOTH sides of the hand-off: 8 MFMAs of k-block kb | barrier | the 8 fragment reads of k-block kb + 1 | the other 8
//     MFMAs of kb -- their operands were read one k-block earlier, so the matrix pipe restarts the moment the barrier releases.
//   * The 64 x 64 f32 result of a wave left as 64 four-byte stores per lane, 16 rows x 64 B each (7 k cycles of store issue);
//     now it goes through the (idle) ring as two 32-row halves and leaves as 16-byte stores, four full 256-byte rows per instruction.
// ---------------------------------------------------------------------------------------------------------------
constexpr int DWR_THREADS = 384;  // 4 compute waves (2 x 2, 64 x 64 each) + 2 loader waves
template <int N>
__device__ __forceinline__ void dwr_wait_behind(int behind) {  // vmcnt(8 x behind)
    if constexpr (N > 0) {
        if (behind >= N) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * N) : "memory");
            return;
        }
        dwr_wait_behind<N - 1>(behind);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
__device__ __forceinline__ void dwr_barrier() {  // lgkmcnt(0) as the builtin: hipcc then knows the fragment reads have landed
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int DW_NBUF>
__global__ __launch_bounds__(DWR_THREADS) void ppo_step_dw_bf16_ring_kernel(DwArgs a, const __bf16* __restrict__ st_tiles) {
    extern __shared__ __align__(16) char dsm[];
    static_assert(DW_NBUF >= 3 && 8 * (DW_NBUF - 2) < 64, "ring depth");
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
            metric_block(a, s_red, tid, DWR_THREADS);
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
    const size_t mat_elems = (size_t)16 * nrb * 512;
    const int Kin = l == 0 ? lay.obs_dim : HID;
    const __bf16* A = reinterpret_cast<const __bf16*>(a.dz) + (size_t)(y * 3 + l) * mat_elems;
    const __bf16* Bm = l == 0 ? st_tiles : reinterpret_cast<const __bf16*>(a.h) + (size_t)(y * 2 + l - 1) * mat_elems;
    const int ncb_b = l == 0 ? 4 : 16;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, kq = lane >> 4;
    const int rb0 = (int)((long long)s * a.rows_per_slab / 32);
    const int rb1 = min(nrb, (int)(((long long)(s + 1) * a.rows_per_slab) / 32));
    const int nkb = max(0, rb1 - rb0);
    constexpr int AHEAD = DW_NBUF - 1;

    if (wave >= 4) {  // ---- loader waves: wave 4 copies the 8 dZ^T tiles of every k-block (LDS slots 0-7), wave 5 the 8 H^T tiles (8-15)
        const int op = wave - 4;
        const __bf16* src[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
            src[t] = (op == 0 ? A + ((size_t)(i0 / 16 + t) * nrb + rb0) * 512 : Bm + ((size_t)min(j0 / 16 + t, ncb_b - 1) * nrb + rb0) * 512) + lane * 8;
        auto issue = [&](int kb, int buf) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + (size_t)kb * 512),
                                                 (__attribute__((address_space(3))) void*)(dsm + buf * DW_BUF_BYTES + (8 * op + t) * 1024), 16, 0, 0);
        };
        int fill = 0;
        for (int kb = 0; kb < AHEAD && kb < nkb; ++kb) {
            issue(kb, fill);
            fill = fill + 1 == DW_NBUF ? 0 : fill + 1;
        }
        for (int kb = 0; kb < nkb; ++kb) {
            dwr_wait_behind<AHEAD - 1>(min(AHEAD - 1, nkb - 1 - kb));
            __builtin_amdgcn_s_barrier();  // hand-off kb: k-block kb has landed, k-block kb - 1 has been read by everybody
            asm volatile("" ::: "memory");
            if (kb + AHEAD < nkb) {
                issue(kb + AHEAD, fill);
                fill = fill + 1 == DW_NBUF ? 0 : fill + 1;
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
    long long* stp = (a.stamps != nullptr && item == 0 && tid == 0) ? a.stamps : nullptr;
    if (stp) stp[0] = clock64();
    bf16x8 fa[2][4], fb[2][4];
    auto read_frags = [&](int kb, bf16x8 (&xa)[4], bf16x8 (&xb)[4]) {
        const char* buf = dsm + (kb % DW_NBUF) * DW_BUF_BYTES + lane * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xa[t] = *reinterpret_cast<const bf16x8*>(buf + (wi * 4 + t) * 1024);
            xb[t] = *reinterpret_cast<const bf16x8*>(buf + (8 + wj * 4 + t) * 1024);
        }
    };
    auto mfmas = [&](const bf16x8 (&xa)[4], const bf16x8 (&xb)[4], int t0) {  // row tiles t0, t0 + 1 (+ the bias sums riding on the A fragments)
        if (!live) return;
#pragma unroll
        for (int ti = t0; ti < t0 + 2; ++ti) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[ti], xb[tj], acc[ti][tj], 0, 0, 0);
            float t8 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) t8 += (float)xa[ti][e];
            bsum[ti] += t8;
        }
    };
    if (nkb > 0) {
        dwr_barrier();  // hand-off 0
        read_frags(0, fa[0], fb[0]);
    }
    for (int kb0 = 0; kb0 < nkb; kb0 += 2) {  // two k-blocks per trip: the register double buffer alternates without copies
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int kb = kb0 + d;
            if (kb >= nkb) break;
            mfmas(fa[d], fb[d], 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < nkb) {
                dwr_barrier();  // hand-off kb + 1
                if (stp && kb < 12) stp[1 + kb] = clock64();
                read_frags(kb + 1, fa[d ^ 1], fb[d ^ 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(fa[d], fb[d], 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (stp) stp[14] = clock64();
    dwr_barrier();  // every compute wave has read its last fragments (the loaders are gone): the ring becomes the store staging area
    float* slab = a.grads + (size_t)s * lay.n_params;
    float* dW = slab + lay.off_w[y][l];
    const bool vec_ok = (lay.n_params & 3) == 0 && (reinterpret_cast<uintptr_t>(a.grads) & 15) == 0;  // 16-byte aligned rows in every slab
    if (live && l != 0 && vec_ok) {
        // 64 x 64 f32 of this wave -> [32][68] f32 in LDS (accumulator layout in, rows out), twice; 16-byte stores, a lane quad-row per
        // instruction: 16 lanes write one full 256-byte row of the gradient
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
    if (stp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[15] = clock64();
    }
}

template <typename K>
int set_lds_b(K kern, size_t bytes) {
    static thread_local const void* done[16] = {};
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

int pack_tiles_bf16(const float* params, const rlx_mlp_layout& lay, void* tiles, hipStream_t st) {
    hipLaunchKernelGGL(pack_tiles_bf16_kernel, dim3(num_cu() * 4), dim3(256), 0, st, params, lay, static_cast<__bf16*>(tiles));
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int launch_rollout_bf16(const RolloutArgs& a, int blocks, hipStream_t st) {
#ifdef RLX_DEV_VARIANTS
    if (rollout_bm_bf16() == 32) {
        const size_t lds = GeoB<2, 8>::LDS_BYTES;
        if (int rc = set_lds_b(rollout_step_bf16_kernel<2>, lds)) return rc;
        hipLaunchKernelGGL(rollout_step_bf16_kernel<2>, dim3(blocks), dim3(512), lds, st, a);
    } else
#endif
    {
        const size_t lds = GeoB<1, 8>::LDS_BYTES;
        // Half layers requested one phase ahead of their use (ORDER 1345, see the kernel): measured in a replayed graph of 64 steps,
        // 7.84 -> 6.84 us against everything up front (ORDER 0); profiles/r06_rollout_request_order.txt.
#define RLX_RO(O) { if (int rc = set_lds_b(rollout_step_bf16_kernel<1, O>, lds)) return rc; hipLaunchKernelGGL((rollout_step_bf16_kernel<1, O>), dim3(blocks), dim3(512), lds, st, a); }
#ifdef RLX_DEV_VARIANTS
        switch (dev_variant("RLX_ROLLOUT_ORDER", 1345)) {
            case 0: RLX_RO(0) break;
            case 1135: RLX_RO(1135) break;
            case 1134: RLX_RO(1134) break;
            case 33: RLX_RO(33) break;
            case 1355: RLX_RO(1355) break;
            case 2345: RLX_RO(2345) break;
            default: RLX_RO(1345) break;
        }
#else
        RLX_RO(1345)
#endif
#undef RLX_RO
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// workspace carve (bytes) of the bf16 step: tiled activation images instead of the f32 row-major ones
size_t bf16_image_bytes(int64_t m) { return (size_t)16 * ((m + 31) / 32) * 512 * sizeof(__bf16); }

int launch_step_bf16(const StepArgs& a, const DwArgs& d, void* st_tiles, int tiles, int dw_blocks, bool op8, bool rows, hipStream_t st) {
    __bf16* stt = static_cast<__bf16*>(st_tiles);
#define RLX_FUSED_LAUNCH(RTV, PDV, OPV, STV, DECV, LDSB)                                                                              \
    {                                                                                                                              \
        const size_t lb = (LDSB);                                                                                                  \
        if (int rc = set_lds_b(ppo_step_fused_bf16_kernel<RTV, 8, PDV, OPV, STV, DECV>, lb)) return rc;                            \
        hipLaunchKernelGGL((ppo_step_fused_bf16_kernel<RTV, 8, PDV, OPV, STV, DECV>), dim3(tiles, 2), dim3(512), lb, st, a, stt); \
    }
    const size_t lds32 = GeoB<2, 8>::LDS_BYTES;
#ifdef RLX_DEV_VARIANTS
    // development builds: the row-split launch, 64-row tiles (RLX_FUSED_RT=4), other weight-ring depths, phase stamps, LDS padding
    const size_t lds64 = GeoB<4, 8>::LDS_BYTES + GeoB<4, 8>::KEEP_BYTES;
    const int pd = dev_variant("RLX_FUSED_PD", 4);
    if (rows) {  // `tiles` counts 64-row tiles
        if (int rc = launch_fused_rows_bf16(a, st_tiles, tiles, st)) return rc;
    } else if (fused_bm_bf16() == 64) {
        if (a.dec.on) { if (op8) RLX_FUSED_LAUNCH(4, 4, 8, false, true, lds64) else RLX_FUSED_LAUNCH(4, 2, 16, false, true, lds64) }
        else if (op8 && a.stamps != nullptr) RLX_FUSED_LAUNCH(4, 4, 8, true, false, lds64)
        else if (op8 && pd == 8) RLX_FUSED_LAUNCH(4, 8, 8, false, false, lds64)
        else if (op8) RLX_FUSED_LAUNCH(4, 4, 8, false, false, lds64)
        else RLX_FUSED_LAUNCH(4, 2, 16, false, false, lds64)
    } else if (a.dec.on) {
        RLX_FUSED_LAUNCH(2, 4, 8, false, true, lds32)
    } else if (a.stamps != nullptr) {
        RLX_FUSED_LAUNCH(2, 4, 8, true, false, lds32)
    } else if (pd == 2) {
        RLX_FUSED_LAUNCH(2, 2, 8, false, false, lds32)
    } else if (pd == 3) {
        RLX_FUSED_LAUNCH(2, 3, 8, false, false, lds32)
    } else {
        RLX_FUSED_LAUNCH(2, 4, 8, false, false, lds32 + (size_t)dev_variant("RLX_FUSED_LDS_PAD", 0))
    }
#else
    (void)rows;
    (void)op8;
    // the product launch: 32-row tiles (two workgroups per CU), weight ring of 4 stages; its decoupled-loss instantiation
    if (a.dec.on) RLX_FUSED_LAUNCH(2, 4, 8, false, true, lds32)
    else RLX_FUSED_LAUNCH(2, 4, 8, false, false, lds32)
#endif
#undef RLX_FUSED_LAUNCH
    RLX_LAUNCH_CHECK();
#define RLX_DWR_LAUNCH(NB)                                                                                                         \
    {                                                                                                                              \
        const size_t rlds = (size_t)(NB) * DW_BUF_BYTES;                                                                           \
        if (int rc = set_lds_b(ppo_step_dw_bf16_ring_kernel<NB>, rlds)) return rc;                                                 \
        hipLaunchKernelGGL(ppo_step_dw_bf16_ring_kernel<NB>, dim3(dw_blocks), dim3(DWR_THREADS), rlds, st, d, static_cast<const __bf16*>(stt)); \
    }
#define RLX_DW_LAUNCH(NB)                                                                                                       \
    {                                                                                                                           \
        const size_t dlds = (size_t)(NB) * DW_BUF_BYTES;                                                                          \
        if (int rc = set_lds_b(ppo_step_dw_bf16_lds_kernel<NB>, dlds)) return rc;                                               \
        hipLaunchKernelGGL(ppo_step_dw_bf16_lds_kernel<NB>, dim3(dw_blocks), dim3(256), dlds, st, d, static_cast<const __bf16*>(stt)); \
    }
    // The ring kernel pays a longer prologue (the loaders' first hand-off) for its faster k-loop and store phase: it wins from
    // 4096 rows on (8192 rows, one box: 15.6 against 16.8 us, the slab reduce behind it 8.2 against 8.8), at 1024 rows the
    // previous kernel is ahead (9.4 against 10.5 us: profiles/r04_per_rank_share_*.txt).
#ifdef RLX_DEV_VARIANTS
    if (dev_variant("RLX_DW_BF16_REG", 0)) {  // the register-streaming variant (kept for comparison)
        hipLaunchKernelGGL(ppo_step_dw_bf16_kernel<3>, dim3(dw_blocks), dim3(256), 0, st, d, static_cast<const __bf16*>(stt));
    } else if (dev_variant("RLX_DW_RING", d.M >= 4096 ? 1 : 0) != 0) {
        switch (dev_variant("RLX_DW_RING_NBUF", 4)) {
            case 6: RLX_DWR_LAUNCH(6) break;
            case 8: RLX_DWR_LAUNCH(8) break;
            case 9: RLX_DWR_LAUNCH(9) break;
            default: RLX_DWR_LAUNCH(4) break;
        }
    } else {
        switch (dw_nbuf()) {
            case 2: RLX_DW_LAUNCH(2) break;
            case 4: RLX_DW_LAUNCH(4) break;
            case 6: RLX_DW_LAUNCH(6) break;
            case 9: RLX_DW_LAUNCH(9) break;
            case 5: RLX_DW_LAUNCH(5) break;
            default: RLX_DW_LAUNCH(3) break;
        }
    }
#else
    if (d.M >= 4096) RLX_DWR_LAUNCH(4)
    else RLX_DW_LAUNCH(3)
#endif
#undef RLX_DWR_LAUNCH
#undef RLX_DW_LAUNCH
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

}  // namespace step
}  // namespace rlx
