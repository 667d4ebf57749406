// gae_scan.hip -- GAE advantages/returns over the (env x step) trajectory buffer, gfx950.
//
// Replaces rlinf/algorithms/advantages.py:24-86 (compute_gae_advantages_and_returns), its
// [n,B,C]<->[T,B] pre/post-processing (rlinf/algorithms/utils.py:67-131,155-174) and
// safe_normalize (rlinf/algorithms/utils.py:397-404).
//
// Data is time-major ([T(+1), B] for C == 1): the recurrence runs along T, coalescing wants lanes
// along B.  One lane owns VEC consecutive envs (VEC*4-byte vector loads, 64*VEC envs per wave); the
// T axis of one env group is cut into NSEG segments, one wave each:
//
//   pass 1  every wave streams its segment backwards (coalesced rows of r, V, done), forms
//           delta_t and the segment's affine map  g_start = A * g_in + G  (A = prod c_t), and
//           parks delta_t / V_t in an LDS slab [t][lane] (lane-private columns: conflict-free);
//   carry   the NSEG maps are exchanged through LDS and composed from the tail -> g_in per wave;
//   pass 2  every wave replays its segment from LDS with the true g_in using the SAME sequential
//           recurrence, writes adv/ret rows (coalesced) and accumulates the masked moments.
//
// NSEG == 1 degenerates to a single streaming pass (no LDS): each byte is read once and written
// once, and the arithmetic is the reference's op-for-op (explicitly rounded mul/add, no FMA
// contraction), so un-normalised outputs are bit-identical to the CPU loop.
//
// Roofline: HBM-bound, 17 B per env-step (r 4 + V 4 + done 1 read, adv 4 + ret 4 write); the
// normalisation pass re-reads/re-writes adv (+8 B), +1 B with a loss mask.

#include <algorithm>
#include <type_traits>

#include "rlx_common.h"

namespace rlx {
namespace {

#ifdef RLX_DEV_VARIANTS
inline int dev_switch(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

struct GaeArgs {
    const float* r;
    const float* v;      // nullptr => critic-free
    const uint8_t* d;
    const uint8_t* m;    // nullptr => all elements
    float* adv;
    float* ret;
    double* partials;    // [gridDim.x][5] = {count, sum_adv, sumsq_adv, sum_ret, sumsq_ret}
    int T, B, C;
    float gamma, gl;
};

template <int VEC> struct Vf;
template <> struct Vf<1> { typedef float type; typedef uint8_t btype; };
template <> struct Vf<2> { typedef float type __attribute__((ext_vector_type(2))); typedef uint16_t btype; };
template <> struct Vf<4> { typedef float type __attribute__((ext_vector_type(4))); typedef uint32_t btype; };

template <int VEC, bool NT>
__device__ __forceinline__ void ld(const float* p, float (&out)[VEC]) {
    typedef typename Vf<VEC>::type vt;
    vt x;
    if constexpr (NT) x = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
    else x = *reinterpret_cast<const vt*>(p);
    if constexpr (VEC == 1) out[0] = x;
    else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[k] = x[k];
    }
}
template <int VEC, bool NT>
__device__ __forceinline__ void st(float* p, const float (&in)[VEC]) {
    typedef typename Vf<VEC>::type vt;
    vt x;
    if constexpr (VEC == 1) x = in[0];
    else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) x[k] = in[k];
    }
    if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<vt*>(p));
    else *reinterpret_cast<vt*>(p) = x;
}
template <int VEC, bool NT>
__device__ __forceinline__ uint32_t ldb(const uint8_t* p) {  // VEC bools packed little-endian
    typedef typename Vf<VEC>::btype bt;
    if constexpr (NT) return (uint32_t)__builtin_nontemporal_load(reinterpret_cast<const bt*>(p));
    else return (uint32_t)*reinterpret_cast<const bt*>(p);
}

struct Moments {
    double n = 0.0, sa = 0.0, qa = 0.0, sr = 0.0, qr = 0.0;
    __device__ __forceinline__ void add(float adv, float ret, bool on) {
        if (on) {
            n += 1.0;
            sa += (double)adv;
            qa += (double)adv * (double)adv;
            sr += (double)ret;
            qr += (double)ret * (double)ret;
        }
    }
};

// In FRONT of the partials (the workspace starts with them): the finished moments of the two arrays a normalisation can follow
// (`which` = 0 advantages, 1 returns) as published words valid << 32 | payload -- (mean, denominator, skip) -- in kNormReplicas
// copies, a 128-byte line each (see standardize_kernel).  The launch that writes the partials clears them.
constexpr int kNormReplicas = 16;
constexpr int kNormWords = 2 * kNormReplicas * 16;
__device__ __forceinline__ unsigned long long* norm_words(double* partials, int which, int replica) {
    return reinterpret_cast<unsigned long long*>(partials) - kNormWords + (which * kNormReplicas + replica) * 16;
}

__device__ __forceinline__ void flush_moments(const Moments& mo, double* partials, double* scratch, long long slot = -1) {
    double v[5] = {mo.n, mo.sa, mo.qa, mo.sr, mo.qr};
    block_sum<5>(v, scratch);
    const size_t at = slot >= 0 ? (size_t)slot : (size_t)blockIdx.x;  // the env group's slot: the final sum keeps its order
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) partials[at * 5 + k] = v[k];
    }
    if (at == 0)  // ONE block per launch takes back the words of the previous normalisation: the standardize launches behind this
        //           one (stream order) read "published" as "published from THESE partials", whatever the workspace held before
        for (int t = threadIdx.x; t < 2 * kNormReplicas * 3; t += blockDim.x) norm_words(partials, 0, t / 3)[t % 3] = 0ull;
}

// One backward step of the reference loop (advantages.py:66-77), op for op.
__device__ __forceinline__ void gae_step(float r, float v, float v_next, float alive, bool critic,
                                         float gamma, float gl, float& g, float& adv, float& ret) {
    if (critic) {
        const float delta = fsub(fadd(r, fmul(fmul(gamma, v_next), alive)), v);
        g = fadd(delta, fmul(fmul(gl, alive), g));
        ret = fadd(g, v);
        adv = fsub(ret, v);
    } else {
        g = fadd(r, fmul(alive, g));  // gamma = lambda = 1, delta = r
        ret = g;
        adv = g;
    }
}

// ------------------------------------------------------------------------------------------
// C == 1 fast path.  grid = ceil(B / (64*VEC)), block = 64*NSEG threads.
// dynamic LDS (NSEG > 1): delta[T][64*VEC] f32, v[T][64*VEC] f32, carryA/carryG[NSEG][64*VEC] f32,
//                         + 5*NSEG doubles of reduction scratch.
// U  = rows per load batch (all loads of a batch are issued before the first dependent op)
// PF (NSEG == 1) = register double-buffering: the next batch's loads are issued before the current batch is
//      consumed, so a wave always has a batch in flight (counted vmcnt waits, in-order returns)
// NT = nontemporal loads/stores (streamed once: keep them out of the way in L2)
// ------------------------------------------------------------------------------------------
template <int VEC, int UU>
struct Batch {
    float r[UU][VEC], v[UU][VEC];
    uint32_t dn[UU], mk[UU];
};

// PAIR (streaming variant, one wave per workgroup): a wave's `done` row is 64 bytes -- half a 128-byte line, and the
// neighbouring wave's half is fetched AGAIN when that wave runs on another XCD (workgroup i goes to XCD i % 8): measured
// reads 1.117 x algorithmic (profiles/r01_*pmc*).  With PAIR the env groups 2k and 2k + 1 are given to workgroups b and
// b + 8 -- the same XCD, dispatched together -- and the `done` loads go through the caches normally, so the second half
// of the line is an L2 hit.
template <int VEC, int NSEG, int U, bool NT, bool CRITIC, bool MASK, bool PAIR = false>
__global__ __launch_bounds__(64 * NSEG) void gae_scan_c1(GaeArgs a) {
    // CRITIC / MASK are compile-time: a runtime "load or constant" select makes hipcc branch around
    // every load and drain vmcnt(0) per element (measured 27 us -> see DESIGN.md).
    // streaming scan: always keep one batch in flight (U >= 128: the whole trajectory is one register batch --
    // every load of the wave is issued before the first dependent op, 384 VGPRs at one wave per SIMD)
    constexpr bool PF = NSEG == 1 && U < 128;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int W = 64 * VEC;  // envs per block
    const int lane = threadIdx.x & 63;
    const int seg = NSEG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    long long grp = blockIdx.x;  // env group (64 * VEC envs) of this workgroup
    if constexpr (PAIR) {
        const unsigned b = blockIdx.x, paired = gridDim.x & ~15u;
        if (b < paired) grp = (long long)(((b >> 4) * 8 + (b & 7)) * 2 + ((b >> 3) & 1));
    }
    const long long e0 = (grp * 64 + lane) * VEC;
    const bool active = e0 < a.B;
    constexpr bool critic = CRITIC;
    const int T = a.T;
    const size_t B = (size_t)a.B;
    const int seg_len = (T + NSEG - 1) / NSEG;
    const int t_lo = seg * seg_len;
    const int t_hi = min(T, t_lo + seg_len);

    float* s_delta = reinterpret_cast<float*>(smem);
    float* s_v = s_delta + (NSEG > 1 ? (size_t)T * W : 0);
    float* s_ca = s_v + (NSEG > 1 ? (size_t)T * W : 0);
    float* s_cg = s_ca + (NSEG > 1 ? NSEG * W : 0);
    double* s_red = reinterpret_cast<double*>(s_cg + (NSEG > 1 ? NSEG * W : 0));

    Moments mo;
    float g[VEC], A[VEC], vnext[VEC];
    unsigned long long alive_bits[VEC], mask_bits[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { g[k] = 0.f; A[k] = 1.f; vnext[k] = 0.f; alive_bits[k] = 0; mask_bits[k] = ~0ull; }

    if (active && t_hi > t_lo) {
        if constexpr (critic) ld<VEC, false>(a.v + (size_t)t_hi * B + e0, vnext);

        // ---- pass 1 (the only pass when NSEG == 1) -------------------------------------------
        auto load = [&](auto& bt, int t_top) {  // rows t_top-1 ... t_top-UU
            constexpr int UU = sizeof(bt.dn) / sizeof(uint32_t);
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                const size_t t = (size_t)(t_top - 1 - u);
                ld<VEC, NT>(a.r + t * B + e0, bt.r[u]);
                if constexpr (critic) ld<VEC, NT>(a.v + t * B + e0, bt.v[u]);
                bt.dn[u] = ldb<VEC, NT && !PAIR>(a.d + (t + 1) * B + e0);
                if constexpr (MASK) bt.mk[u] = ldb<VEC, NT && !PAIR>(a.m + t * B + e0);  // (a 64-byte row like `done`: see PAIR)
                else bt.mk[u] = 0x01010101u;
            }
        };
        auto consume = [&](auto& bt, int t_top) {
            constexpr int UU = sizeof(bt.dn) / sizeof(uint32_t);
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                const int t = t_top - 1 - u;
                float adv[VEC], ret[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const bool done = (bt.dn[u] >> (8 * k)) & 0xffu;
                    const bool on = (bt.mk[u] >> (8 * k)) & 0xffu;
                    const float alive = done ? 0.f : 1.f;
                    const float vk = critic ? bt.v[u][k] : 0.f;
                    if constexpr (NSEG == 1) {
                        gae_step(bt.r[u][k], vk, vnext[k], alive, critic, a.gamma, a.gl, g[k], adv[k], ret[k]);
                        mo.add(adv[k], ret[k], on);
                    } else {
                        // local scan with g_in = 0; remember delta, V and the alive/mask bits
                        const float delta = critic ? fsub(fadd(bt.r[u][k], fmul(fmul(a.gamma, vnext[k]), alive)), vk)
                                                   : bt.r[u][k];
                        const float c = critic ? fmul(a.gl, alive) : alive;
                        g[k] = fadd(delta, fmul(c, g[k]));
                        A[k] = A[k] * c;
                        adv[k] = delta;  // staged below
                        const unsigned long long bit = 1ull << (t - t_lo);
                        if (!done) alive_bits[k] |= bit;
                        if (!on) mask_bits[k] &= ~bit;
                    }
                    vnext[k] = vk;
                }
                if constexpr (NSEG == 1) {
                    st<VEC, NT>(a.adv + (size_t)t * B + e0, adv);
                    st<VEC, NT>(a.ret + (size_t)t * B + e0, ret);
                } else {
                    st<VEC, false>(s_delta + ((size_t)t * 64 + lane) * VEC, adv);
                    if constexpr (critic) st<VEC, false>(s_v + ((size_t)t * 64 + lane) * VEC, bt.v[u]);
                }
            }
        };
        int t = t_hi;
        if constexpr (PF) {
            if (t - U >= t_lo) {
                Batch<VEC, U> b0, b1;
                load(b0, t);
                while (true) {
                    const bool more1 = t - 2 * U >= t_lo;
                    if (more1) load(b1, t - U);
                    consume(b0, t);
                    t -= U;
                    if (!more1) break;
                    const bool more0 = t - 2 * U >= t_lo;
                    if (more0) load(b0, t - U);
                    consume(b1, t);
                    t -= U;
                    if (!more0) break;
                }
            }
        } else {
            for (; t - U >= t_lo; t -= U) {
                Batch<VEC, U> b;
                load(b, t);
                consume(b, t);
            }
        }
        for (; t > t_lo; --t) {
            Batch<VEC, 1> b;
            load(b, t);
            consume(b, t);
        }
    }

    if constexpr (NSEG > 1) {
        // ---- carry exchange: compose the later segments' maps, tail first -----------------------
        if (active) {
            st<VEC, false>(s_ca + ((size_t)seg * 64 + lane) * VEC, A);
            st<VEC, false>(s_cg + ((size_t)seg * 64 + lane) * VEC, g);
        }
        __syncthreads();
        float gin[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) gin[k] = 0.f;
        if (active) {
            for (int s = NSEG - 1; s > seg; --s) {
                float As[VEC], Gs[VEC];
                ld<VEC, false>(s_ca + ((size_t)s * 64 + lane) * VEC, As);
                ld<VEC, false>(s_cg + ((size_t)s * 64 + lane) * VEC, Gs);
#pragma unroll
                for (int k = 0; k < VEC; ++k) gin[k] = fadd(Gs[k], fmul(As[k], gin[k]));
            }
            // ---- pass 2: replay the segment from LDS with the true incoming accumulator --------
#pragma unroll 4
            for (int t = t_hi - 1; t >= t_lo; --t) {
                float dl[VEC], vv[VEC], adv[VEC], ret[VEC];
                ld<VEC, false>(s_delta + ((size_t)t * 64 + lane) * VEC, dl);
                if constexpr (critic) ld<VEC, false>(s_v + ((size_t)t * 64 + lane) * VEC, vv);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float alive = ((alive_bits[k] >> (t - t_lo)) & 1ull) ? 1.f : 0.f;
                    const bool on = (mask_bits[k] >> (t - t_lo)) & 1ull;
                    if (critic) {
                        gin[k] = fadd(dl[k], fmul(fmul(a.gl, alive), gin[k]));
                        ret[k] = fadd(gin[k], vv[k]);
                        adv[k] = fsub(ret[k], vv[k]);
                    } else {
                        gin[k] = fadd(dl[k], fmul(alive, gin[k]));
                        ret[k] = gin[k];
                        adv[k] = gin[k];
                    }
                    mo.add(adv[k], ret[k], on);
                }
                st<VEC, NT>(a.adv + (size_t)t * B + e0, adv);
                st<VEC, NT>(a.ret + (size_t)t * B + e0, ret);
            }
        }
    }
    flush_moments(mo, a.partials, s_red, grp);
}

#ifdef RLX_DEV_VARIANTS  // register-resident segment scans: bit-exact, measured slower (32-44 us against 28 us): development builds only
// ------------------------------------------------------------------------------------------
// Register-resident segmented scan (C == 1).  grid = ceil(B / (64*VEC)), block = 64*NSEG threads.
//   Wave s of a block owns time segment [s*SEG, (s+1)*SEG) of 64*VEC envs and issues EVERY load of that segment
//   up front (SEG rows x (r, V, done[, mask]) = 9 B per element live in VGPRs: SEG = 32, VEC = 4 -> 288 registers at
//   one wave per SIMD, ~72 KB in flight per wave, 1 KiB per load instruction).  The recurrence then runs as a
//   serial chain over the waves, latest segment first: wave s waits at a workgroup barrier for the accumulator
//   g and hands it on through LDS (64*VEC floats -- the only LDS traffic), so every wave executes the reference's
//   sequential recurrence with the true incoming g: results are bit-identical to the CPU loop, every byte is
//   read once and written once, and while one wave computes and stores, the later waves' loads are still landing.
// ------------------------------------------------------------------------------------------
// Buffer (SRSRC) addressing: descriptor built from kernargs only (provably wave-uniform: no waterfall loops), the
// per-lane part is one 32-bit voffset shared by every access, the row is a scalar soffset -- no per-lane 64-bit
// address exists anywhere, which is what lets SEG x VEC x 9 bytes of segment data own the register file.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int VEC, bool NT>
__device__ __forceinline__ void bld(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float (&out)[VEC]) {
    constexpr int aux = NT ? 2 : 0;
    if constexpr (VEC == 1) {
        out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, aux));
    } else if constexpr (VEC == 2) {
        const u32x2 x = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, aux);
        out[0] = __uint_as_float(x[0]); out[1] = __uint_as_float(x[1]);
    } else {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = __uint_as_float(x[k]);
    }
}
template <int VEC, bool NT>
__device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const float (&in)[VEC]) {
    constexpr int aux = NT ? 2 : 0;
    if constexpr (VEC == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(in[0]), rs, voff, soff, aux);
    } else if constexpr (VEC == 2) {
        u32x2 x; x[0] = __float_as_uint(in[0]); x[1] = __float_as_uint(in[1]);
        __builtin_amdgcn_raw_buffer_store_b64(x, rs, voff, soff, aux);
    } else {
        u32x4 x;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = __float_as_uint(in[k]);
        __builtin_amdgcn_raw_buffer_store_b128(x, rs, voff, soff, aux);
    }
}
template <int VEC, bool NT>
__device__ __forceinline__ uint32_t bldb(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    constexpr int aux = NT ? 2 : 0;
    if constexpr (VEC == 1) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, aux);
    else if constexpr (VEC == 2) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, aux);
    else return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, aux);
}

template <int VEC, int NSEG, int SEG, bool NT, bool CRITIC, bool MASK>
__global__ __launch_bounds__(64 * NSEG) void gae_scan_regseg(GaeArgs a) {
    __shared__ float s_g[64 * VEC];
    __shared__ double s_red[5 * NSEG];
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned e0 = ((unsigned)blockIdx.x * 64u + (unsigned)lane) * VEC;
    const bool active = e0 < (unsigned)a.B;  // B % VEC == 0 (host-checked): a lane is all in or all out
    const int T = a.T;
    const unsigned B = (unsigned)a.B;
    const int t_lo = seg * SEG;
    const int t_hi = min(T, t_lo + SEG);
    const unsigned nf = (unsigned)T * B * 4u, nf1 = (unsigned)(T + 1) * B * 4u;  // < 2^31 (host-checked)
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.r), 0, (int)nf, 0x00020000);
    const auto rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v), 0, CRITIC ? (int)nf1 : 0, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.d), 0, (int)(nf1 / 4), 0x00020000);
    const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.m), 0, MASK ? (int)(nf / 4) : 0, 0x00020000);
    const auto rs_adv = __builtin_amdgcn_make_buffer_rsrc(a.adv, 0, (int)nf, 0x00020000);
    const auto rs_ret = __builtin_amdgcn_make_buffer_rsrc(a.ret, 0, (int)nf, 0x00020000);
    // inactive lanes (e0 >= B) park their offset past every buffer: hardware bounds-checking returns 0 / drops stores
    const unsigned vo4 = active ? e0 * 4u : 0x7fffffffu, vo1 = active ? e0 : 0x7fffffffu;

    float r[SEG][VEC], v[SEG][VEC], vtop[VEC];
    uint32_t dn[SEG], mk[SEG];
    // rows past the end of a ragged last segment re-load row T-1 (valid memory, value unused): a clamped
    // address keeps the loads branch-free, a per-row predicate would make hipcc drain vmcnt(0) per row
#pragma unroll
    for (int u = SEG - 1; u >= 0; --u) {  // latest row first: that is the order the recurrence consumes them
        const unsigned t = (unsigned)min(t_lo + u, T - 1);
        bld<VEC, NT>(rs_r, vo4, t * B * 4u, r[u]);
        if constexpr (CRITIC) bld<VEC, NT>(rs_v, vo4, t * B * 4u, v[u]);
        dn[u] = bldb<VEC, NT>(rs_d, vo1, (t + 1) * B);
        if constexpr (MASK) mk[u] = bldb<VEC, NT>(rs_m, vo1, t * B);
        else mk[u] = 0x01010101u;
    }
    if constexpr (CRITIC) bld<VEC, false>(rs_v, vo4, (unsigned)min(t_hi, T) * B * 4u, vtop);

    Moments mo;
    for (int s = NSEG - 1; s >= 0; --s) {
        if (s == seg && active && t_hi > t_lo) {
            float g[VEC], vnext[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                g[k] = (t_hi == T) ? 0.f : s_g[lane * VEC + k];
                vnext[k] = CRITIC ? vtop[k] : 0.f;
            }
#pragma unroll
            for (int u = SEG - 1; u >= 0; --u) {
                if (t_lo + u < t_hi) {  // wave-uniform
                    float adv[VEC], ret[VEC];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const bool done = (dn[u] >> (8 * k)) & 0xffu;
                        const bool on = (mk[u] >> (8 * k)) & 0xffu;
                        const float vk = CRITIC ? v[u][k] : 0.f;
                        gae_step(r[u][k], vk, vnext[k], done ? 0.f : 1.f, CRITIC, a.gamma, a.gl, g[k], adv[k], ret[k]);
                        mo.add(adv[k], ret[k], on);
                        vnext[k] = vk;
                    }
                    const unsigned row = (unsigned)(t_lo + u) * B * 4u;
                    bst<VEC, NT>(rs_adv, vo4, row, adv);
                    bst<VEC, NT>(rs_ret, vo4, row, ret);
                }
            }
            if (s > 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) s_g[lane * VEC + k] = g[k];
            }
        }
        if (NSEG > 1 && s > 0) __syncthreads();
    }
    flush_moments(mo, a.partials, s_red);
}

// ------------------------------------------------------------------------------------------
// Register-resident segments with an EARLY hand-off (C == 1, one env per lane).  grid = ceil(B / 64), block = 64 * NSEG threads.
//   Why: a wave can have at most 64 vector-memory operations outstanding (vmcnt is 6 bits), so the one-wave-per-env-group
//   streaming scan caps the bytes in flight at B x 256 B whatever its register batches hold -- 16.8 MB at 65536 envs, about what
//   5.3 TB/s x 3 us needs and no more.  More waves per env group is the only way to put more requests in flight.
//   How: wave s owns the time segment [s * SEG, (s + 1) * SEG) and requests ALL its rows up front (r, V, done: 3 * SEG loads).
//   The recurrence needs the accumulator g from the later segment; gae_scan_regseg passes it down a serial chain in which
//   every wave computes AND stores before the next one starts.  Here the chain carries ONLY the recurrence -- SEG fused
//   multiply-add-class steps from registers, no stores -- then every wave runs its full pass (recurrence again, moments, stores)
//   CONCURRENTLY with its true incoming g.  Same operations in the same order as the sequential loop: bit-identical results.
// ------------------------------------------------------------------------------------------
template <int NSEG, int SEG, bool NT, bool CRITIC, bool MASK>
__global__ __launch_bounds__(64 * NSEG) void gae_scan_handoff(GaeArgs a) {
    __shared__ float s_gin[NSEG][64];
    __shared__ double s_red[5 * NSEG];
    const int lane = threadIdx.x & 63;
    const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned e0 = (unsigned)blockIdx.x * 64u + (unsigned)lane;
    const bool active = e0 < (unsigned)a.B;
    const int T = a.T;
    const unsigned B = (unsigned)a.B;
    const int t_lo = seg * SEG;
    const int t_hi = min(T, t_lo + SEG);
    const unsigned nf = (unsigned)T * B * 4u, nf1 = (unsigned)(T + 1) * B * 4u;  // < 2^31 (host-checked)
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.r), 0, (int)nf, 0x00020000);
    const auto rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.v), 0, CRITIC ? (int)nf1 : 0, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.d), 0, (int)(nf1 / 4), 0x00020000);
    const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.m), 0, MASK ? (int)(nf / 4) : 0, 0x00020000);
    const auto rs_adv = __builtin_amdgcn_make_buffer_rsrc(a.adv, 0, (int)nf, 0x00020000);
    const auto rs_ret = __builtin_amdgcn_make_buffer_rsrc(a.ret, 0, (int)nf, 0x00020000);
    const unsigned vo4 = active ? e0 * 4u : 0x7fffffffu, vo1 = active ? e0 : 0x7fffffffu;

    float r[SEG][1], v[SEG][1], vtop[1] = {0.f};
    uint32_t dn[SEG], mk[SEG];
#pragma unroll
    for (int u = SEG - 1; u >= 0; --u) {  // latest row first: the order the recurrence consumes them
        const unsigned t = (unsigned)min(t_lo + u, T - 1);
        bld<1, NT>(rs_r, vo4, t * B * 4u, r[u]);
        if constexpr (CRITIC) bld<1, NT>(rs_v, vo4, t * B * 4u, v[u]);
        dn[u] = bldb<1, NT>(rs_d, vo1, (t + 1) * B);
        if constexpr (MASK) mk[u] = bldb<1, NT>(rs_m, vo1, t * B);
        else mk[u] = 1u;
    }
    if constexpr (CRITIC) bld<1, false>(rs_v, vo4, (unsigned)min(t_hi, T) * B * 4u, vtop);

    // ---- the chain: recurrence only, latest segment first; s_gin[s] = the accumulator segment s hands to segment s - 1 ----
    for (int s = NSEG - 1; s >= 1; --s) {
        if (s == seg) {
            float g = (s == NSEG - 1 || t_hi >= T) ? 0.f : s_gin[s + 1 < NSEG ? s + 1 : s][lane];
            float vnext = CRITIC ? vtop[0] : 0.f, adv, ret;
#pragma unroll
            for (int u = SEG - 1; u >= 0; --u) {
                if (t_lo + u < t_hi) {
                    const float vk = CRITIC ? v[u][0] : 0.f;
                    gae_step(r[u][0], vk, vnext, (dn[u] & 0xffu) ? 0.f : 1.f, CRITIC, a.gamma, a.gl, g, adv, ret);
                    vnext = vk;
                }
            }
            s_gin[s][lane] = g;
        }
        __syncthreads();
    }
    // ---- the full pass, all segments at once -------------------------------------------------------------------------------
    Moments mo;
    if (active && t_hi > t_lo) {
        float g = (seg == NSEG - 1 || t_hi >= T) ? 0.f : s_gin[seg + 1 < NSEG ? seg + 1 : seg][lane];
        float vnext = CRITIC ? vtop[0] : 0.f;
#pragma unroll
        for (int u = SEG - 1; u >= 0; --u) {
            if (t_lo + u < t_hi) {  // wave-uniform
                float adv[1], ret[1];
                const float vk = CRITIC ? v[u][0] : 0.f;
                gae_step(r[u][0], vk, vnext, (dn[u] & 0xffu) ? 0.f : 1.f, CRITIC, a.gamma, a.gl, g, adv[0], ret[0]);
                mo.add(adv[0], ret[0], (mk[u] & 0xffu) != 0);
                vnext = vk;
                const unsigned row = (unsigned)(t_lo + u) * B * 4u;
                bst<1, NT>(rs_adv, vo4, row, adv);
                bst<1, NT>(rs_ret, vo4, row, ret);
            }
        }
    }
    flush_moments(mo, a.partials, s_red);
}

#endif  // RLX_DEV_VARIANTS
// ------------------------------------------------------------------------------------------
// Generic time-chunk layout (C > 1): one lane per env, sequential, strided addressing.
//   time step t = k*C + c  ->  element ((k*B + b)*C + c); dones use flat rows shifted by C-1
//   (the reference keeps the LAST T+1 rows of the (n+1)*C flattened done rows, utils.py:111-114),
//   values the FIRST T+1 rows (utils.py:116-120).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t tc_index(int f, size_t b, size_t B, int C) {
    return ((size_t)(f / C) * B + b) * C + (f % C);
}

__global__ __launch_bounds__(64) void gae_scan_chunked(GaeArgs a) {
    __shared__ double s_red[5];
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    const bool critic = a.v != nullptr;
    Moments mo;
    if (b < (size_t)a.B) {
        float g = 0.f;
        float vnext = critic ? a.v[tc_index(a.T, b, a.B, a.C)] : 0.f;
        for (int t = a.T - 1; t >= 0; --t) {
            const size_t i = tc_index(t, b, a.B, a.C);
            const float r = a.r[i];
            const float v = critic ? a.v[i] : 0.f;
            const bool done = a.d[tc_index(t + 1 + (a.C - 1), b, a.B, a.C)] != 0;
            const bool on = a.m ? a.m[i] != 0 : true;
            float adv, ret;
            gae_step(r, v, vnext, done ? 0.f : 1.f, critic, a.gamma, a.gl, g, adv, ret);
            vnext = v;
            a.adv[i] = adv;
            a.ret[i] = ret;
            mo.add(adv, ret, on);
        }
    }
    flush_moments(mo, a.partials, s_red);
}

// ------------------------------------------------------------------------------------------
// Moments of an arbitrary masked array (stand-alone safe_normalize) and the normalising pass.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moments_kernel(const float* x, const uint8_t* m, size_t n, double* partials) {
    __shared__ double s_red[5 * 4];
    Moments mo;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float xv = x[i];
        mo.add(xv, xv, m ? m[i] != 0 : true);
    }
    flush_moments(mo, partials, s_red);
}

// which = 0: use {sum_adv, sumsq_adv}; which = 1: {sum_ret, sumsq_ret}.
// A read-modify-write stream behind a small reduction.  A thread owns kStdU float4s per sweep and ALL of a sweep's loads are issued
// before anything else -- in front of the reduction of the moment partials, whose latency (a few L2 round trips and two barriers)
// then passes while the first sweep's data is in flight; round 5's form (load -> divide -> store, one float4 at a time, behind
// the reduction) left every wave with one request in flight: 3.8 TB/s.  NT: the array is streamed once and far larger than the
// caches (the 65 536-env shape) -> non-temporal both ways; the contract shape lives in L2 and uses ordinary accesses.
template <bool NT, int kStdU>
__global__ __launch_bounds__(256) void standardize_kernel(float* x, size_t n, const double* partials, int nparts,
                                                          int which, float eps) {
    __shared__ double s_red[5 * 4];
    __shared__ unsigned s_pub[6];
    typedef float v4f __attribute__((ext_vector_type(4)));
    const size_t n4 = (reinterpret_cast<uintptr_t>(x) % 16 == 0) ? n / 4 : 0;
    v4f* x4 = reinterpret_cast<v4f*>(x);
    const size_t sweep = (size_t)gridDim.x * 256 * kStdU;
    size_t bb = (size_t)blockIdx.x * 256 * kStdU;  // a block's sweep is one contiguous 16 KiB run; thread t owns float4s bb + t + 256 u
    v4f q[kStdU];
    auto load_sweep = [&](size_t b) {
#pragma unroll
        for (int u = 0; u < kStdU; ++u) {
            const size_t i = b + threadIdx.x + (size_t)u * 256;
            if (i < n4) {
                if constexpr (NT) q[u] = __builtin_nontemporal_load(x4 + i);
                else q[u] = x4[i];
            }
        }
    };
    // The moments are FINISHED ONCE: block 0 reduces the partials (fixed order), forms mean and denominator of BOTH arrays (the
    // partial records hold them side by side, so the returns' launch finds its words ready) and publishes them as 64-bit words
    // valid << 32 | f32 bits (relaxed agent-scope stores: single-copy atomic, no fence); every other block requests its sweep
    // first and has three lanes poll one of kNormReplicas copies while that data is in flight.  Round 6's first form let all 1024
    // blocks reduce the 1024 partials themselves: 24 MB of reads on the same few hundred L2 lines at the same moment -- 2.9 of the
    // launch's 13.8 us (with 8 partials: 10.9), and no ordering of the loads hid it (partials first: 15.7); a single published
    // copy polled by 1024 blocks still queued 3072 requests on one line (12.6).
    double* parts = const_cast<double*>(partials);
    if (blockIdx.x == 0) {
        double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        constexpr int PB = 4;
        double pa[PB][5];
#pragma unroll
        for (int u = 0; u < PB; ++u) {  // (one batch of loads: 4 x 5 doubles per lane cover 1024 partials)
            const size_t pi = (size_t)min((int)threadIdx.x + 256 * u, nparts - 1);
#pragma unroll
            for (int k = 0; k < 5; ++k) pa[u][k] = partials[pi * 5 + k];
        }
#pragma unroll
        for (int u = 0; u < PB; ++u)
            if ((int)threadIdx.x + 256 * u < nparts) {
#pragma unroll
                for (int k = 0; k < 5; ++k) acc[k] += pa[u][k];
            }
        for (int p = threadIdx.x + 256 * PB; p < nparts; p += blockDim.x) {
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] += partials[(size_t)p * 5 + k];
        }
        block_sum<5>(acc, s_red);
        if (threadIdx.x == 0) {
            const double cnt = acc[0];
            const unsigned skip = cnt <= 0.0;  // "if len(valid_array) > 0" (utils.py:399)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const double sum = acc[1 + 2 * w], sq = acc[2 + 2 * w];
                const double mean = sum / cnt;
                const double var = (sq - sum * sum / cnt) / (cnt - 1.0);  // unbiased; NaN for cnt == 1, as torch
                const float mf = (float)mean, df = fadd((float)sqrt(var > 0.0 || var != var ? var : 0.0), eps);
                s_pub[w * 3 + 0] = __float_as_uint(mf), s_pub[w * 3 + 1] = __float_as_uint(df), s_pub[w * 3 + 2] = skip;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * kNormReplicas * 3) {  // (in the returns' launch this republishes the same words; nobody waits for it)
            const int rep = threadIdx.x / 3, k = threadIdx.x % 3;  // rep counts through both arrays' copies
            __hip_atomic_store(norm_words(parts, 0, rep) + k, 1ull << 32 | s_pub[(rep / kNormReplicas) * 3 + k], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        load_sweep(bb);  // (behind the partials: a wave's loads return in order, and the sweep queues with the whole chip's)
    } else {
        load_sweep(bb);
        if (threadIdx.x < 64) {  // waits for a block with a LOWER index (dispatched first: forward progress), bounded all the same
            const unsigned long long* mine = norm_words(parts, which, blockIdx.x % kNormReplicas) + (threadIdx.x < 3 ? threadIdx.x : 0);
            unsigned long long w = 0;
            const long long t0 = wall_clock64();
            bool late = false;
            for (int spins = 0;; ++spins) {
                if (threadIdx.x < 3) w = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((__ballot((w >> 32) != 0ull) & 7ull) == 7ull) break;
                if (spins > 16) {
                    if (wall_clock64() - t0 > 200000000ll) {  // 2 s: block 0 never ran -- never hang the GPU over it; the array
                        late = true;                           // comes back NaN (loud in every metric), not silently unscaled
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (threadIdx.x < 3) s_pub[which * 3 + threadIdx.x] = late ? (threadIdx.x < 2 ? 0x7FC00000u : 0u) : (unsigned)w;
        }
    }
    __syncthreads();
    __syncthreads();
    if (s_pub[which * 3 + 2]) return;
    const float mean = __uint_as_float(s_pub[which * 3 + 0]), den = __uint_as_float(s_pub[which * 3 + 1]);
    // a / den with the reciprocal hoisted out of the stream: y = RN(1 / den), q0 = RN(a y), one residual step q = fma(fma(-q0,
    // den, a), y, q0) -- Markstein's correction: the correctly rounded quotient unless den's significand is all ones (then at most
    // one ulp off); 4 VALU operations per element instead of the division sequence's ~12, which at 16-32 elements per lane was
    // a measurable share of the launch (14.4 -> 13.6 us at 65 536 x 128).
    const float rden = 1.0f / den;
    auto quot = [&](float a) -> float {
        const float q0 = a * rden;
        return __builtin_fmaf(__builtin_fmaf(-q0, den, a), rden, q0);
    };
    for (; bb < n4; bb += sweep) {
        v4f r[kStdU];
#pragma unroll
        for (int u = 0; u < kStdU; ++u) {
            r[u].x = quot(fsub(q[u].x, mean));
            r[u].y = quot(fsub(q[u].y, mean));
            r[u].z = quot(fsub(q[u].z, mean));
            r[u].w = quot(fsub(q[u].w, mean));
        }
        if (bb + sweep < n4) load_sweep(bb + sweep);  // the next sweep's loads in front of this sweep's stores
#pragma unroll
        for (int u = 0; u < kStdU; ++u) {
            const size_t i = bb + threadIdx.x + (size_t)u * 256;
            if (i < n4) {
                if constexpr (NT) __builtin_nontemporal_store(r[u], x4 + i);
                else x4[i] = r[u];
            }
        }
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        x[i] = quot(fsub(x[i], mean));
}

constexpr int kMaxParts = 4096;

inline size_t lds_bytes(int vec, int nseg, int T) {
    if (nseg == 1) return 0;
    const size_t W = 64 * (size_t)vec;
    return (size_t)T * W * 4 * 2 + (size_t)nseg * W * 4 * 2 + 5 * nseg * sizeof(double);
}

template <int VEC, int NSEG, int U, bool NT, bool CRITIC, bool MASK, bool PAIR = false>
int launch_c1(const GaeArgs& a, hipStream_t s, int nblk) {
    const size_t lds = lds_bytes(VEC, NSEG, a.T);
    auto kern = gae_scan_c1<VEC, NSEG, U, NT, CRITIC, MASK, PAIR>;
    if (lds > 48 * 1024) {
        RLX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * NSEG), lds, s, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// Only VEC = 1, 8 rows per batch, plain (non-nontemporal) accesses are instantiated: the MI355X sweep
// (profiles/r01_gae_variant_sweep.txt) showed wider vectors, deeper batches and nt accesses never win
// (fewer waves in flight outweighs the wider loads), so they are compiled out rather than shipped.
template <int NSEG>
int dispatch_c1(const GaeArgs& a, hipStream_t s, int nblk, int rows = 8, bool nt = false) {
    const bool critic = a.v != nullptr, mask = a.m != nullptr;
    if constexpr (NSEG == 1) {  // streaming-scan tuning variants (critic, no mask: the bench / sweep configuration)
        if (critic && !mask) {
#ifdef RLX_DEV_VARIANTS
            if (rows == 16) return nt ? launch_c1<1, 1, 16, true, true, false>(a, s, nblk) : launch_c1<1, 1, 16, false, true, false>(a, s, nblk);
            if (rows == 32) return nt ? launch_c1<1, 1, 32, true, true, false>(a, s, nblk) : launch_c1<1, 1, 32, false, true, false>(a, s, nblk);
            if (rows == 64 && nt && dev_switch("RLX_GAE_PAIR", 1) == 0) return launch_c1<1, 1, 64, true, true, false>(a, s, nblk);
            if (rows == 64 && !nt) return launch_c1<1, 1, 64, false, true, false>(a, s, nblk);
            if (rows == 128) return nt ? launch_c1<1, 1, 128, true, true, false>(a, s, nblk) : launch_c1<1, 1, 128, false, true, false>(a, s, nblk);
            if (nt && rows != 64) return launch_c1<1, 1, 8, true, true, false>(a, s, nblk);
#else
            if (rows != 64 && (rows != 8 || nt)) {
                set_error("rlx_gae_scan: this streaming tuning variant (rows %d, nt %d) is compiled into development builds only", rows, (int)nt);
                return RLX_ENOSYS;
            }
            if (rows == 64 && !nt) {
                set_error("rlx_gae_scan: the 64-row streaming scan is compiled with non-temporal accesses only (development builds hold the other)");
                return RLX_ENOSYS;
            }
#endif
            if (rows == 64) return launch_c1<1, 1, 64, true, true, false, true>(a, s, nblk);  // the measured-best streaming scan (auto at HBM sizes)
        }
        // ... and with a loss mask (round 6): the same streaming form at 32 rows per register batch -- the mask byte is a fourth
        // register per row, and two 64-row batches of four would not fit the register file
        if (critic && mask && rows == 64 && nt) return launch_c1<1, 1, 32, true, true, true, true>(a, s, nblk);
    }
    if (critic && !mask) return launch_c1<1, NSEG, 8, false, true, false>(a, s, nblk);
    if (critic && mask) return launch_c1<1, NSEG, 8, false, true, true>(a, s, nblk);
    if (!critic && mask) return launch_c1<1, NSEG, 8, false, false, true>(a, s, nblk);
    return launch_c1<1, NSEG, 8, false, false, false>(a, s, nblk);
}

#ifdef RLX_DEV_VARIANTS
template <int VEC, int NSEG, int SEG, bool NT>
int launch_regseg(const GaeArgs& a, hipStream_t s) {
    const int nblk = ceil_div(a.B, 64 * VEC);
    const bool critic = a.v != nullptr, mask = a.m != nullptr;
    const dim3 g(nblk), b(64 * NSEG);
    if (critic && !mask) hipLaunchKernelGGL((gae_scan_regseg<VEC, NSEG, SEG, NT, true, false>), g, b, 0, s, a);
    else if (critic && mask) hipLaunchKernelGGL((gae_scan_regseg<VEC, NSEG, SEG, NT, true, true>), g, b, 0, s, a);
    else if (!critic && mask) hipLaunchKernelGGL((gae_scan_regseg<VEC, NSEG, SEG, NT, false, true>), g, b, 0, s, a);
    else hipLaunchKernelGGL((gae_scan_regseg<VEC, NSEG, SEG, NT, false, false>), g, b, 0, s, a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// register-resident segments of SEG steps: NSEG = ceil(T / SEG) waves per block (at most 16 waves)
template <int VEC, int SEG, bool NT>
int dispatch_regseg(const GaeArgs& a, hipStream_t s) {
    const int nseg = ceil_div(a.T, SEG);
    if (nseg <= 1) return launch_regseg<VEC, 1, SEG, NT>(a, s);
    if (nseg <= 2) return launch_regseg<VEC, 2, SEG, NT>(a, s);
    if (nseg <= 4) return launch_regseg<VEC, 4, SEG, NT>(a, s);
    if (nseg <= 8) return launch_regseg<VEC, 8, SEG, NT>(a, s);
    if (nseg <= 16) return launch_regseg<VEC, 16, SEG, NT>(a, s);
    set_error("rlx_gae_scan: the register-resident variant needs T <= %d (T=%d)", 16 * SEG, a.T);
    return RLX_EINVAL;
}

#endif  // RLX_DEV_VARIANTS

int standardize_grid(size_t n, int u, int per_cu) {
    const long long want = (long long)((n / 4 + 256 * u - 1) / (256 * u));
    const long long cap = (long long)num_cu() * per_cu;
    return (int)std::max<long long>(1, std::min(want, cap));
}
// (x - mean) / (std + eps) in place from the moment partials.  Arrays beyond the caches are streamed non-temporally, eight float4s
// per lane and sweep on four blocks per CU (rocprofv3 medians at 65 536 x 128, profiles/r06_standardize_sweep.txt: 13.6 us against
// 14.4-14.8 for four float4s on eight blocks, 16.3 for two; round 5's unbatched loop: 17.5).
void launch_standardize(float* x, size_t n, const double* partials, int nparts, int which, float eps, hipStream_t s) {
    if (n * sizeof(float) > ((size_t)16 << 20))
        hipLaunchKernelGGL((standardize_kernel<true, 8>), dim3(standardize_grid(n, 8, 4)), dim3(256), 0, s, x, n, partials, nparts, which, eps);
    else
        hipLaunchKernelGGL((standardize_kernel<false, 4>), dim3(standardize_grid(n, 4, 8)), dim3(256), 0, s, x, n, partials, nparts, which, eps);
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_gae_workspace_bytes(int n_chunk, int batch, int chunk) {
    (void)n_chunk;
    (void)chunk;
    if (batch <= 0) return 0;
    return (size_t)kNormWords * 8 + (size_t)(ceil_div(batch, 64)) * 5 * sizeof(double);
}

extern "C" size_t rlx_standardize_workspace_bytes(size_t n) {
    (void)n;
    return (size_t)kNormWords * 8 + (size_t)kMaxParts * 5 * sizeof(double);
}

extern "C" int rlx_gae_scan(const float* rewards, const float* values, const uint8_t* dones,
                            const uint8_t* loss_mask, float* advantages, float* returns, void* workspace,
                            size_t workspace_bytes, int n_chunk, int batch, int chunk,
                            const rlx_gae_params* p, rlx_stream_t stream) {
    RLX_REQUIRE(p != nullptr, "rlx_gae_scan: NULL params");
    RLX_REQUIRE(n_chunk >= 0 && batch >= 0 && chunk >= 1, "rlx_gae_scan: bad sizes n_chunk=%d batch=%d chunk=%d",
                n_chunk, batch, chunk);
    if (n_chunk == 0 || batch == 0) return RLX_OK;  // empty buffer: nothing to do (pointers may be NULL)
    RLX_REQUIRE(rewards && dones && advantages && returns, "rlx_gae_scan: NULL argument");
    RLX_REQUIRE(workspace != nullptr, "rlx_gae_scan: NULL workspace");
    if (workspace_bytes < rlx_gae_workspace_bytes(n_chunk, batch, chunk)) {
        set_error("rlx_gae_scan: workspace %zu < %zu bytes", workspace_bytes,
                  rlx_gae_workspace_bytes(n_chunk, batch, chunk));
        return RLX_ENOSPC;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    GaeArgs a;
    a.r = rewards; a.v = values; a.d = dones; a.m = loss_mask; a.adv = advantages; a.ret = returns;
    a.partials = reinterpret_cast<double*>(static_cast<char*>(workspace) + kNormWords * 8);  // (the normalisation words come first)
    a.T = n_chunk * chunk; a.B = batch; a.C = chunk;
    a.gamma = p->gamma; a.gl = p->gamma_lambda;
    const size_t n = (size_t)a.T * batch;

    int nblk;
    if (chunk != 1) {
        nblk = ceil_div(batch, 64);
        hipLaunchKernelGGL(gae_scan_chunked, dim3(nblk), dim3(64), 0, s, a);
        RLX_LAUNCH_CHECK();
    } else {
        // variant word: bits 0-7 envs per lane (only 1 is compiled in), bits 8-15 time segments
        int vec = p->variant & 0xff, nseg = (p->variant >> 8) & 0xff;
        const int rows = (p->variant >> 16) & 0xff;   // rows per register batch (0 = 8)
        const bool nt = (p->variant >> 25) & 1;        // nontemporal loads/stores
        if (p->variant == 0) {
            // auto, from the MI355X sweep: with >= 4 env groups per CU the pure streaming scan is the
            // fastest (and bit-exact); below that, split time over just enough waves to reach that
            // occupancy, keeping segments of 8..64 steps and the LDS slab under 150 KB.
            const int cus = num_cu();
            const int groups = ceil_div(batch, 64);
            vec = 1;
            nseg = 1;
            while (nseg < 8 && groups * nseg < 4 * cus && a.T / (nseg * 2) >= 8 &&
                   lds_bytes(1, nseg * 2, a.T) <= 150 * 1024)
                nseg *= 2;
            if (nseg > 1 && ceil_div(a.T, nseg) > 64) nseg = 1;
        }
        if (vec == 0) vec = 1;
        if (nseg == 0) nseg = 1;
        int rows_sel = rows;
        bool nt_sel = nt;
        if (p->variant == 0 && nseg == 1 && a.T >= 128 && a.T % 64 == 0 && ceil_div(batch, 64) >= 4 * num_cu()) {
            // HBM-regime sweep (profiles/r01_gae_hbm_sweep_stream.txt): with the buffer far larger than the caches, 64-row
            // register batches (two of them = a 128-step trajectory entirely in flight) + nontemporal accesses is the
            // fastest streaming variant (60.5 % of 8 TB/s vs 58 %); critic only (with a loss mask: 32-row batches, dispatch_c1)
            rows_sel = 64;
            nt_sel = true;
        }
        const bool regseg = (p->variant >> 26) & 1;  // register-resident 32-step segments, serial carry chain
        const bool handoff = (p->variant >> 28) & 1;  // register-resident segments, early hand-off (gae_scan_handoff)
#ifndef RLX_DEV_VARIANTS
        if (handoff || regseg) {
            set_error("rlx_gae_scan: the register-resident segment variants are compiled into development builds only (-DRLX_DEV_VARIANTS)");
            return RLX_ENOSYS;
        }
        {
#else
        if (handoff) {
            RLX_REQUIRE((unsigned long long)(a.T + 1) * (unsigned long long)batch * 4ull < (1ull << 31),
                        "rlx_gae_scan: the hand-off variant addresses rows through 32-bit buffer offsets: (T+1)*B*4 must stay below 2 GiB");
            const int seg = 16 << ((p->variant >> 29) & 3);  // 16, 32, 64
            RLX_REQUIRE(seg <= 64, "rlx_gae_scan: hand-off segment of %d steps", seg);
            const int ns = ceil_div(a.T, seg);
            RLX_REQUIRE(ns >= 1 && ns <= 8, "rlx_gae_scan: hand-off variant with %d-step segments needs T <= %d (T=%d)", seg, 8 * seg, a.T);
            const bool critic = a.v != nullptr, mask = a.m != nullptr;
            nblk = ceil_div(batch, 64);
#define RLX_HANDOFF(NS, SG)                                                                                                        \
            do {                                                                                                                       \
                if (critic && !mask) { if (nt) hipLaunchKernelGGL((gae_scan_handoff<NS, SG, true, true, false>), dim3(nblk), dim3(64 * NS), 0, s, a);   \
                                       else hipLaunchKernelGGL((gae_scan_handoff<NS, SG, false, true, false>), dim3(nblk), dim3(64 * NS), 0, s, a); }    \
                else if (critic) hipLaunchKernelGGL((gae_scan_handoff<NS, SG, false, true, true>), dim3(nblk), dim3(64 * NS), 0, s, a);                \
                else if (mask) hipLaunchKernelGGL((gae_scan_handoff<NS, SG, false, false, true>), dim3(nblk), dim3(64 * NS), 0, s, a);                 \
                else hipLaunchKernelGGL((gae_scan_handoff<NS, SG, false, false, false>), dim3(nblk), dim3(64 * NS), 0, s, a);                          \
            } while (0)
            const int nsp = ns <= 1 ? 1 : ns <= 2 ? 2 : ns <= 4 ? 4 : 8;
            if (seg == 64) { if (nsp == 1) RLX_HANDOFF(1, 64); else if (nsp == 2) RLX_HANDOFF(2, 64); else if (nsp == 4) RLX_HANDOFF(4, 64); else RLX_HANDOFF(8, 64); }
            else if (seg == 32) { if (nsp == 1) RLX_HANDOFF(1, 32); else if (nsp == 2) RLX_HANDOFF(2, 32); else if (nsp == 4) RLX_HANDOFF(4, 32); else RLX_HANDOFF(8, 32); }
            else { if (nsp == 1) RLX_HANDOFF(1, 16); else if (nsp == 2) RLX_HANDOFF(2, 16); else if (nsp == 4) RLX_HANDOFF(4, 16); else RLX_HANDOFF(8, 16); }
#undef RLX_HANDOFF
            RLX_LAUNCH_CHECK();
        } else if (regseg) {
            RLX_REQUIRE(vec == 1 || vec == 2 || vec == 4, "rlx_gae_scan: regseg vec=%d", vec);
            RLX_REQUIRE(batch % vec == 0, "rlx_gae_scan: regseg vec=%d needs batch %% vec == 0 (batch=%d)", vec, batch);
            RLX_REQUIRE((unsigned long long)(a.T + 1) * (unsigned long long)batch * 4ull < (1ull << 31),
                        "rlx_gae_scan: regseg addresses rows through 32-bit buffer offsets: (T+1)*B*4 must stay below 2 GiB");
            const uintptr_t al = (uintptr_t)rewards | (uintptr_t)values | (uintptr_t)advantages | (uintptr_t)returns;
            RLX_REQUIRE(vec == 1 || al % (4 * vec) == 0, "rlx_gae_scan: regseg vec=%d needs %d-byte aligned buffers", vec, 4 * vec);
            nblk = ceil_div(batch, 64 * vec);
            const bool seg16 = (p->variant >> 27) & 1;  // 16-step segments (twice the waves, half the registers)
            int rc;
            // VEC = 4 keeps 16-step segments only: 32 x 4 x 9 B = 288 data registers makes hipcc spill
            if (vec == 4) rc = nt ? dispatch_regseg<4, 16, true>(a, s) : dispatch_regseg<4, 16, false>(a, s);
            else if (vec == 2 && seg16) rc = nt ? dispatch_regseg<2, 16, true>(a, s) : dispatch_regseg<2, 16, false>(a, s);
            else if (vec == 2) rc = nt ? dispatch_regseg<2, 32, true>(a, s) : dispatch_regseg<2, 32, false>(a, s);
            else if (seg16) rc = nt ? dispatch_regseg<1, 16, true>(a, s) : dispatch_regseg<1, 16, false>(a, s);
            else rc = nt ? dispatch_regseg<1, 32, true>(a, s) : dispatch_regseg<1, 32, false>(a, s);
            if (rc != RLX_OK) return rc;
        } else {
#endif
        if (vec != 1) {
            set_error("rlx_gae_scan: vec=%d is compiled out (no gain on MI355X, see profiles/r01_gae_variant_sweep.txt)", vec);
            return RLX_ENOSYS;
        }
        RLX_REQUIRE(nseg == 1 || nseg == 2 || nseg == 4 || nseg == 8 || nseg == 16, "rlx_gae_scan: variant nseg=%d", nseg);
        RLX_REQUIRE(nseg == 1 || ceil_div(a.T, nseg) <= 64, "rlx_gae_scan: nseg=%d leaves segments > 64 steps (T=%d)",
                    nseg, a.T);
        RLX_REQUIRE(lds_bytes(vec, nseg, a.T) <= 160 * 1024, "rlx_gae_scan: nseg=%d T=%d exceeds the 160 KB LDS", nseg, a.T);
        nblk = ceil_div(batch, 64);
        int rc = RLX_ENOSYS;
        switch (nseg) {
            case 1: rc = dispatch_c1<1>(a, s, nblk, rows_sel ? rows_sel : 8, nt_sel); break;
            case 2: rc = dispatch_c1<2>(a, s, nblk); break;
            case 4: rc = dispatch_c1<4>(a, s, nblk); break;
            case 8: rc = dispatch_c1<8>(a, s, nblk); break;
            case 16: rc = dispatch_c1<16>(a, s, nblk); break;
        }
        if (rc != RLX_OK) return rc;
        }
    }
    if (p->normalize_advantages) {
        launch_standardize(advantages, n, a.partials, nblk, 0, p->norm_eps, s);
        RLX_LAUNCH_CHECK();
    }
    if (p->normalize_returns) {
        launch_standardize(returns, n, a.partials, nblk, 1, p->norm_eps, s);
        RLX_LAUNCH_CHECK();
    }
    return RLX_OK;
}

extern "C" int rlx_masked_standardize(float* x, const uint8_t* mask, size_t n, float eps, void* workspace,
                                      size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(x != nullptr || n == 0, "rlx_masked_standardize: NULL x");
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(workspace != nullptr, "rlx_masked_standardize: NULL workspace");
    if (workspace_bytes < rlx_standardize_workspace_bytes(n)) {
        set_error("rlx_masked_standardize: workspace too small");
        return RLX_ENOSPC;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nblk = (int)std::max<long long>(1, std::min<long long>((long long)((n + 255) / 256), std::min(kMaxParts, num_cu() * 8)));
    double* partials = reinterpret_cast<double*>(static_cast<char*>(workspace) + kNormWords * 8);
    hipLaunchKernelGGL(moments_kernel, dim3(nblk), dim3(256), 0, s, x, mask, n, partials);
    RLX_LAUNCH_CHECK();
    launch_standardize(x, n, partials, nblk, 0, eps, s);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
