// adamw_clip.hip -- global-norm gradient clipping + two-group AdamW on flat f32 buffers, gfx950.
//
// Replaces FSDPModelManager.optimizer_step (rlinf/hybrid_engines/fsdp/fsdp_model_manager.py:429-463):
//   torch.nn.utils.clip_grad_norm_ (strategy/fsdp.py:363-369, the all-NO_SHARD branch)
//   + torch.optim.AdamW.step with the actor / value_head parameter groups (fsdp_model_manager.py:501-590),
//   including "skip the step when the norm is non-finite".
// Two launches per optimizer step: (1) sum the split-K gradient slabs, scale, write the reduced gradient
// and per-block sum-of-squares partials (one thread also forms the step's scalars in double); (2) every block
// re-reduces the (<= 1024) partials, derives the clip coefficient and applies clip + AdamW to its float4 slice.  HBM-bound: 28 B per parameter (+4 B per extra
// gradient slab); at 287 504 parameters it is launch-latency bound, which is why it is only two launches.
// (A single persistent launch around a device-wide barrier was measured: 33-49 us against 7 + 11 us here -- on an 8-XCD part
// the arrivals are serialised same-address device-scope atomics and every poll crosses to the memory-side coherence point.)

#include <algorithm>

#include "opt_common.h"

namespace rlx {
namespace {

using namespace opt;

// state (device, int32[2]): [0] = optimizer steps applied so far, [1] = "the previous call applied a step"
// (folded into [0] here, i.e. strictly after that call's update kernel and before this call's), so that a
// captured hipGraph can be replayed without host-side step bookkeeping.
__device__ __forceinline__ void form_scalars(const rlx_adamw_params& a, int step, AdamScalars* sc);

// publish + wait (see opt_common.h, PeerWait).  Called by every block: thread r < world handles rank r.
__device__ __forceinline__ void peer_poll(const PeerWait& w, int r, unsigned s, int sleep) {
    const unsigned* flag = w.flags_mine + w.phase * kMaxRanks + r;
    // an earlier wait of this run already expired (the status word stays set until the host reads it): the exchange is dead,
    // every further wait would cost its full bound again -- fall through, the AdamW launches skip, the host raises
    if (__hip_atomic_load(w.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - s) < 0) {
        if (wall_clock64() - t0 > w.timeout_ticks) {
            *w.status = 1;
            break;
        }
        if (sleep > 64) __builtin_amdgcn_s_sleep(127);
        else __builtin_amdgcn_s_sleep(8);
    }
}

__device__ __forceinline__ void peer_handshake(const PeerWait& w) {
    if (w.world <= 1) {
        if (w.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the wait was the previous launch; this is its acquire
        return;
    }
    const unsigned s = *w.seq + 1u;
    if ((int)threadIdx.x < w.world) {
        const int r = threadIdx.x;
        if (blockIdx.x == 0)  // this rank's buffer s is complete (the launch that wrote it ended before this one began)
            __hip_atomic_store(w.flags_peer[r] + w.phase * kMaxRanks + w.rank, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        peer_poll(w, r, s, 8);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: nothing of the peers' buffers may be served from a stale line
}

// The same hand-shake as a launch of its own: one wave, thread r handles rank r, ~5 us between polls.  The launch behind it
// starts when every peer has published; nothing else of this rank occupies the device while it waits -- which is what lets
// ranks that SHARE a device (the one-GPU test set-up, an oversubscribed node) make progress instead of starving each other.
__global__ __launch_bounds__(64) void peer_wait_kernel(PeerWait w) {
    const unsigned s = *w.seq + 1u;
    if ((int)threadIdx.x < w.world) {
        const int r = threadIdx.x;
        __hip_atomic_store(w.flags_peer[r] + w.phase * kMaxRanks + w.rank, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        peer_poll(w, r, s, 127);
    }
}

// DEFER (one base): the slabs are summed group by group with the device-side scales of opt_common.h's DeferredScale.
template <bool ADAM, bool DEFER = false>
__global__ __launch_bounds__(256) void grad_reduce_sqnorm(ReduceSrc src, float* __restrict__ out, long long n, float scale,
                                                          double* __restrict__ partials, int* __restrict__ state,
                                                          rlx_adamw_params a, AdamScalars* __restrict__ scalars, PeerWait wait,
                                                          DeferredScale dfr) {
    __shared__ double s_red[4];
    if (ADAM && state != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && state[1] != 0) {
        state[0] += 1;  // the previous call applied its step: fold it in strictly before this call's AdamW launch reads it
        state[1] = 0;
    }
    (void)a;
    (void)scalars;
    peer_handshake(wait);
    if (src.seq != nullptr) {
        const size_t off = (size_t)((*src.seq + 1u) & 1u) * (size_t)src.slot_stride;
#pragma unroll
        for (int b = 0; b < kMaxRanks; ++b) src.base[b] += off;
    }
    double acc[1] = {0.0};
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nslab = src.nslab, nbase = src.nbase;
    const bool inplace_single = !DEFER && nbase == 1 && nslab == 1 && scale == 1.f && out == src.base[0];
    // float4 body (slab stride n*4 bytes keeps 16-byte alignment when n % 4 == 0), slabs unrolled four at a time so that
    // a lane has up to 64 bytes in flight; scalar tail / fallback below
    bool vec = (n % 4 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    for (int b = 0; b < nbase; ++b) vec = vec && (reinterpret_cast<uintptr_t>(src.base[b]) % 16 == 0);
    const long long n4 = vec ? n / 4 : 0;
    for (long long i = tid0; i < n4; i += stride) {
        float4 g;
        if constexpr (DEFER) {
            g = sum_slab_groups_f4(reinterpret_cast<const float4*>(src.base[0]), i, n4, nslab, dfr);
        } else if (nbase == 1) {
            g = sum_slabs_f4(reinterpret_cast<const float4*>(src.base[0]) + i, n4, nslab);  // opt_common.h
        } else {  // one staged gradient per rank, all peer loads in flight before the first add, fixed rank order
            float4 x[kMaxRanks];
#pragma unroll
            for (int b = 0; b < kMaxRanks; ++b)
                if (b < nbase) x[b] = reinterpret_cast<const float4*>(src.base[b])[i];
            g = x[0];
#pragma unroll
            for (int b = 1; b < kMaxRanks; ++b)
                if (b < nbase) { g.x += x[b].x; g.y += x[b].y; g.z += x[b].z; g.w += x[b].w; }
        }
        g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
        if (!inplace_single) reinterpret_cast<float4*>(out)[i] = g;
        acc[0] += (double)g.x * (double)g.x + (double)g.y * (double)g.y + (double)g.z * (double)g.z + (double)g.w * (double)g.w;
    }
    for (long long i = n4 * 4 + tid0; i < n; i += stride) {
        float g;
        if constexpr (DEFER) {
            g = sum_slab_groups(src.base[0], i, n, nslab, dfr);
        } else {
            g = src.base[0][i];
            for (int k = 1; k < nslab; ++k) g += src.base[0][(long long)k * n + i];
        }
        for (int b = 1; b < nbase; ++b) g += src.base[b][i];
        g *= scale;
        if (!inplace_single) out[i] = g;
        acc[0] += (double)g * (double)g;
    }
    block_sum<1>(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}

// Reduce-scatter half of the RS + AG all-reduce: this rank sums ITS shard (float4 elements [lo4, lo4 + n4)) of every rank's
// staged gradient -- W peer reads of n / W floats each instead of n -- scales it, writes it to its own fine-grained shard area
// (read by every peer in the gather) and leaves one squared-norm partial per block next to it.
__global__ __launch_bounds__(256) void reduce_scatter_kernel(ReduceSrc src, float* __restrict__ shard0, long long shard_stride,
                                                             double* __restrict__ parts0, int nparts, long long lo4, long long n4,
                                                             float scale, int* __restrict__ state, unsigned* __restrict__ seq_snapshot,
                                                             PeerWait wait) {
    __shared__ double s_red[4];
    if (state != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && state[1] != 0) {
        state[0] += 1;  // the previous optimizer call applied its step: folded in strictly before this call's AdamW launch
        state[1] = 0;
    }
    // The gather + AdamW launch behind this one INCREMENTS the sequence word at its end, so its own blocks must not read it (a
    // block that runs late would see seq + 1: the other slot, a flag value nobody publishes).  They read this copy instead, which
    // nothing writes while that launch runs.
    if (seq_snapshot != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *seq_snapshot = *src.seq;
    peer_handshake(wait);
    const unsigned slot = (*src.seq + 1u) & 1u;
    const size_t off = (size_t)slot * (size_t)src.slot_stride;
    float* __restrict__ shard_out = shard0 + (size_t)slot * (size_t)shard_stride;
    double* __restrict__ parts_out = parts0 + (size_t)slot * (size_t)nparts;
    const int nbase = src.nbase;
    double acc[1] = {0.0};
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 x[kMaxRanks];
#pragma unroll
        for (int b = 0; b < kMaxRanks; ++b)
            if (b < nbase) x[b] = reinterpret_cast<const float4*>(src.base[b] + off)[lo4 + i];
        float4 g = x[0];
#pragma unroll
        for (int b = 1; b < kMaxRanks; ++b)
            if (b < nbase) { g.x += x[b].x; g.y += x[b].y; g.z += x[b].z; g.w += x[b].w; }
        g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
        reinterpret_cast<float4*>(shard_out)[i] = g;
        acc[0] += (double)g.x * (double)g.x + (double)g.y * (double)g.y + (double)g.z * (double)g.z + (double)g.w * (double)g.w;
    }
    block_sum<1>(acc, s_red);
    if (threadIdx.x == 0) parts_out[blockIdx.x] = acc[0];
}

__device__ __forceinline__ float4 gather_f4(const GatherSrc& src, size_t off, long long i) {
    const int owner = (int)(i / src.shard4);
    const float* base = src.shard[0];
#pragma unroll
    for (int r = 1; r < kMaxRanks; ++r)
        if (r == owner) base = src.shard[r];  // scalar selects: an indexed read of the argument block would be a vector load
    return reinterpret_cast<const float4*>(base + off)[i - (long long)owner * src.shard4];
}

// All-gather half on its own (validation / the plain all-reduce entry point): out = every rank's reduced shard, in place order.
__global__ __launch_bounds__(256) void gather_kernel(GatherSrc src, float* __restrict__ out, long long n4, PeerWait wait) {
    peer_handshake(wait);
    const size_t off = (size_t)((*src.seq + 1u) & 1u) * (size_t)src.slot_stride;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<float4*>(out)[i] = gather_f4(src, off, i);
}

// Fragment-tile weight image (ppo_step.hip, struct Tiles) kept in step with the flat parameters: the optimizer writes
// every updated weight to its tile slot(s) as well, so no separate re-pack launch is needed before the next forward.
__device__ __forceinline__ size_t tile_slot(size_t mat_off, int nit, int n, int k) {  // f32 tiles: 16 (n) x 16 (k), two per 32 k
    return mat_off + ((((size_t)(n >> 4) * nit + (k >> 5)) * 2 + ((k >> 4) & 1)) * 64 + ((k >> 2) & 3) * 16 + (n & 15)) * 4 + (k & 3);
}
__device__ __forceinline__ size_t tile_slot_bf16(size_t mat_off, int nit, int n, int k) {  // bf16 tiles: 16 (n) x 32 (k), k-step major
    (void)nit;
    return mat_off + (((size_t)(k >> 5) * 16 + (n >> 4)) * 64 + ((k >> 3) & 3) * 16 + (n & 15)) * 8 + (k & 7);
}
// FMT: 0 = f32 tiles (exact-f32 MFMA launches), 1 = bf16 tiles, 2 = three bf16 planes hi | mid | lo of the f32 weight (the default
// f32 image: ppo_step_f32x.hip; plane stride = both networks' tiles)
template <int FMT>
__device__ __forceinline__ void tile_store(float* __restrict__ tiles, size_t mat_off, int nit, int n, int k, float val) {
    if constexpr (FMT == 1) {
        reinterpret_cast<__bf16*>(tiles)[tile_slot_bf16(mat_off, nit, n, k)] = (__bf16)val;
    } else if constexpr (FMT == 2) {
        constexpr size_t plane = 2 * ((size_t)256 * 64 + 4 * (size_t)256 * 256);
        __bf16* t = reinterpret_cast<__bf16*>(tiles) + tile_slot_bf16(mat_off, nit, n, k);
        const __bf16 h = (__bf16)val;
        const float r1 = fsub(val, (float)h);
        const __bf16 md = (__bf16)r1;
        t[0] = h;
        t[plane] = md;
        t[2 * plane] = (__bf16)fsub(r1, (float)md);
    } else {
        tiles[tile_slot(mat_off, nit, n, k)] = val;
    }
}
template <int BF16>
__device__ __forceinline__ void tile_scatter(const rlx_mlp_layout& lay, float* __restrict__ tiles, long long i, float val) {
    constexpr int HIDW = 256, K1P = 64;
    const size_t per_net = (size_t)HIDW * K1P + 4 * (size_t)HIDW * HIDW;
#pragma unroll
    for (int y = 0; y < 2; ++y) {
        const long long r0 = i - lay.off_w[y][0];
        if (r0 >= 0 && r0 < (long long)HIDW * lay.obs_dim) {
            const int n = (int)(r0 / lay.obs_dim), k = (int)(r0 % lay.obs_dim);
            tile_store<BF16>(tiles, y * per_net, K1P / 32, n, k, val);
            return;
        }
#pragma unroll
        for (int l = 1; l <= 2; ++l) {
            const long long r = i - lay.off_w[y][l];
            if (r >= 0 && r < (long long)HIDW * HIDW) {
                const int o = (int)(r >> 8), in = (int)(r & 255);
                const size_t base = y * per_net + (size_t)HIDW * K1P;
                tile_store<BF16>(tiles, base + (size_t)(l - 1) * HIDW * HIDW, HIDW / 32, o, in, val);   // W_l
                tile_store<BF16>(tiles, base + (size_t)(l + 1) * HIDW * HIDW, HIDW / 32, in, o, val);   // W_l^T
                return;
            }
        }
    }
}

// beta ** step for an integer step by repeated squaring: <= 2 log2(step) double multiplications (each within half an ulp; the
// result is narrowed to f32 right after) instead of libm's pow(), which costs ~3 us on the single lane that forms the scalars
__device__ __forceinline__ double ipow(double b, int e) {
    double r = 1.0;
    while (e > 0) {
        if (e & 1) r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}

__device__ __forceinline__ void form_scalars(const rlx_adamw_params& a, int step, AdamScalars* sc) {
    const double bc1 = 1.0 - ipow(a.beta1, step);
    const double bc2 = 1.0 - ipow(a.beta2, step);
    sc->bc2_sqrt = (float)sqrt(bc2);                 // bias_correction2 ** 0.5
    sc->one_m_b1 = (float)(1.0 - a.beta1);           // lerp weight
    sc->one_m_b2 = (float)(1.0 - a.beta2);           // addcmul value
    sc->beta2 = (float)a.beta2;
    sc->eps = (float)a.eps;
    for (int k = 0; k < RLX_ADAMW_MAX_GROUPS; ++k) {
        if (k < a.n_groups) {  // one double division per LIVE group: this lane is the launch's critical path
            const double lr = a.groups[k].lr;
            sc->step_size[k] = (float)(lr / bc1);
            sc->decay[k] = (float)(1.0 - lr * a.weight_decay);
        } else {
            sc->step_size[k] = 0.f;
            sc->decay[k] = 1.f;
        }
    }
}

__device__ __forceinline__ int group_of(const rlx_adamw_params& a, long long i) {
    int grp = -1;
#pragma unroll
    for (int k = 0; k < RLX_ADAMW_MAX_GROUPS; ++k)
        if (k < a.n_groups && i >= a.groups[k].begin && i < a.groups[k].end) grp = k;
    return grp;
}

__device__ __forceinline__ void adamw_elem(float& pi, float& gi, float& mi, float& vi, float coef, int grp, bool skip,
                                           const AdamScalars& sc) {
    gi = gi * coef;  // grads.mul_(clip_coef_clamped): always applied
    if (skip || grp < 0) return;
    pi = pi * sc.decay[grp];                                  // param.mul_(1 - lr*wd)
    mi = mi + (gi - mi) * sc.one_m_b1;                        // exp_avg.lerp_(grad, 1-b1)
    vi = vi * sc.beta2 + sc.one_m_b2 * gi * gi;               // mul_(b2).addcmul_(g,g,1-b2)
    const float denom = sqrtf(vi) / sc.bc2_sqrt + sc.eps;
    pi = pi - sc.step_size[grp] * (mi / denom);               // addcdiv_(m, denom, -step_size)
}

// Clip + AdamW on one float4 of parameters per lane (element 4 * (iw + lane) .. + 3), the clipped gradient written back, and the
// fragment-tile weight image kept in step.  `iw` is wave-uniform: the tile scatter shuffles across lanes.
__device__ __forceinline__ void update_values_f4(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, long long i, bool live, float4 p4, float4 g4, float4 m4, float4 v4,
                                                 float coef, bool skip, const AdamScalars& sc, const rlx_adamw_params& a,
                                                 float (&pe)[4], int (&grp)[4]) {
    float ge[4] = {g4.x, g4.y, g4.z, g4.w}, me[4] = {m4.x, m4.y, m4.z, m4.w}, ve[4] = {v4.x, v4.y, v4.z, v4.w};
    pe[0] = p4.x, pe[1] = p4.y, pe[2] = p4.z, pe[3] = p4.w;
    grp[0] = grp[1] = grp[2] = grp[3] = -1;
    if (live) {
        const int g_first = group_of(a, 4 * i), g_last = group_of(a, 4 * i + 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            grp[e] = (e == 0) ? g_first : (e == 3 ? g_last : (g_first == g_last ? g_first : group_of(a, 4 * i + e)));
            adamw_elem(pe[e], ge[e], me[e], ve[e], coef, grp[e], skip, sc);
        }
        reinterpret_cast<float4*>(g)[i] = float4{ge[0], ge[1], ge[2], ge[3]};
        if (!skip && (grp[0] >= 0 || grp[1] >= 0 || grp[2] >= 0 || grp[3] >= 0)) {
            reinterpret_cast<float4*>(p)[i] = float4{pe[0], pe[1], pe[2], pe[3]};
            reinterpret_cast<float4*>(m)[i] = float4{me[0], me[1], me[2], me[3]};
            reinterpret_cast<float4*>(v)[i] = float4{ve[0], ve[1], ve[2], ve[3]};
        }
    }
}
// The weight image's copy of the float4 a lane just updated (any parameter: the slot search of tile_scatter).
__device__ __forceinline__ void scatter_tiles_f4(long long iw, int lane, const float (&pe)[4], const int (&grp)[4], bool skip,
                                                 const rlx_adamw_params& a, const rlx_mlp_layout& lay, float* __restrict__ tiles) {
    if (tiles != nullptr) {
        // The scatter wants CONSECUTIVE parameters in consecutive lanes (a weight row's 64 neighbours land in a handful of
        // full tile lines); a lane that scatters its own four values writes 4 bytes of every 16 per instruction instead --
        // measured on the f32 image: AdamW 12 -> 16.5 us and the next fused launch 102 -> 120 us behind the partial-line
        // writes.  So the wave transposes: instruction e covers elements 4 * iw + 64 e .. + 63, lane L takes component L % 4
        // of lane 16 e + L / 4.
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int srcl = 16 * e + (lane >> 2), c = lane & 3;
            const float v0 = __shfl(pe[0], srcl, 64), v1 = __shfl(pe[1], srcl, 64), v2 = __shfl(pe[2], srcl, 64),
                        v3 = __shfl(pe[3], srcl, 64);
            const int q0 = __shfl(grp[0], srcl, 64), q1 = __shfl(grp[1], srcl, 64), q2 = __shfl(grp[2], srcl, 64),
                      q3 = __shfl(grp[3], srcl, 64);
            const float val = c == 0 ? v0 : (c == 1 ? v1 : (c == 2 ? v2 : v3));
            const int gq = c == 0 ? q0 : (c == 1 ? q1 : (c == 2 ? q2 : q3));
            if (skip || gq < 0) continue;
            const long long idx = 4 * iw + 64 * e + lane;
            if (a.tiles_bf16 == 1) tile_scatter<1>(lay, tiles, idx, val);
            else if (a.tiles_bf16 == 2) tile_scatter<2>(lay, tiles, idx, val);
            else tile_scatter<0>(lay, tiles, idx, val);
        }
    }
}
__device__ __forceinline__ void update_f4(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                          long long iw, int lane, bool live, float4 p4, float4 g4, float4 m4, float4 v4, float coef,
                                          bool skip, const AdamScalars& sc, const rlx_adamw_params& a, const rlx_mlp_layout& lay,
                                          float* __restrict__ tiles) {
    float pe[4];
    int grp[4];
    update_values_f4(p, g, m, v, iw + lane, live, p4, g4, m4, v4, coef, skip, sc, a, pe, grp);
    scatter_tiles_f4(iw, lane, pe, grp, skip, a, lay, tiles);
}

constexpr int kSyncSlot0 = 2, kSlotStride = 2;  // sync[0] = epoch, sync[1] = "a poll expired" (sticky), then one slot of two words per block
constexpr int kOneThreads = 512, kMaxSegs = 10, kOneRows = kOneThreads / 64;  // rows of a hidden matrix per block
constexpr int kMaxOneBlocks = 512;  // slots a polling wave covers (8 per lane)
static_assert(kMaxOneBlocks == kExchangeSlots, "the exchange's partial slots (opt_common.h) are the one-launch kernel's");
struct SegPlan {
    int nseg, nblk;
    int blk0[kMaxSegs];                          // first block of segment k
    long long start4[kMaxSegs], end4[kMaxSegs];  // its float4 range
    int mat[kMaxSegs];                           // -1, or y * 2 + (l - 1): the segment IS hidden matrix l of network y
};
// A block's place under a SegPlan: its first float4, this thread's float4, its segment's first block and matrix id.
struct SegBlock {
    long long blk4, i;
    int b0, mat;
    bool live;
};
__device__ __forceinline__ SegBlock seg_block(const SegPlan& plan, int bidx = -1) {  // (scalar selects over the argument block)
    if (bidx < 0) bidx = (int)blockIdx.x;
    long long s4 = plan.start4[0], e4 = plan.end4[0];
    int b0 = 0, mat = plan.mat[0];
#pragma unroll
    for (int q = 1; q < kMaxSegs; ++q)
        if (q < plan.nseg && bidx >= plan.blk0[q]) s4 = plan.start4[q], e4 = plan.end4[q], b0 = plan.blk0[q], mat = plan.mat[q];
    SegBlock sb;
    sb.blk4 = s4 + (long long)(bidx - b0) * kOneThreads;
    sb.i = sb.blk4 + threadIdx.x;
    sb.b0 = b0, sb.mat = mat, sb.live = sb.i < e4;
    return sb;
}
typedef __bf16 (*ImageStage)[kOneRows][256 + 8];  // [plane][row of the block][input]: the transposed image's staging (12.4 KiB)
// Clip + AdamW on this thread's float4 and the weight image's copy of it.  LAST thing a kernel does (it returns early per block).
__device__ __forceinline__ void apply_seg_block(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                const SegBlock& sb, float4 p4, float4 g4, float4 m4, float4 v4, float coef, bool skip,
                                                const AdamScalars& sc, const rlx_adamw_params& a, const rlx_mlp_layout& lay,
                                                float* __restrict__ tiles, ImageStage s_t) {
    const int lane = threadIdx.x & 63;
    float pe[4];
    int grp[4];
    update_values_f4(p, g, m, v, sb.i, sb.live, p4, g4, m4, v4, coef, skip, sc, a, pe, grp);
    if (tiles == nullptr) return;
    if (sb.mat < 0 || a.tiles_bf16 == 0) {  // (block-uniform)
        scatter_tiles_f4(sb.i - lane, lane, pe, grp, skip, a, lay, tiles);
        return;
    }
    // eight whole rows of hidden matrix l of network y: rows row0 .. row0 + 7, this thread's four inputs in0 .. in0 + 3.  Does any
    // optimizer group reach into this block's 2048 parameters (scalar)?  If not (critic warm-up), or the step is skipped, the image
    // stands; otherwise every value is written -- an unchanged parameter rewrites what the image already holds.
    bool touched = false;
#pragma unroll
    for (int q = 0; q < RLX_ADAMW_MAX_GROUPS; ++q)
        touched = touched || (q < a.n_groups && a.groups[q].begin < 4 * sb.blk4 + 4 * kOneThreads && a.groups[q].end > 4 * sb.blk4);
    if (skip || !touched) return;
    constexpr size_t per_net = (size_t)256 * 64 + 4 * (size_t)256 * 256, plane = 2 * per_net;
    const int y = sb.mat >> 1, l = (sb.mat & 1) + 1, nplanes = a.tiles_bf16 == 2 ? 3 : 1;
    const size_t hid = (size_t)y * per_net + (size_t)256 * 64;
    const int row0 = ((int)blockIdx.x - sb.b0) * kOneRows, rr = threadIdx.x >> 6, in0 = 4 * lane;
    __bf16 pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)pe[e];
        const float r1 = fsub(pe[e], (float)h);
        const __bf16 md = (__bf16)r1;
        pl[0][e] = h, pl[1][e] = md, pl[2][e] = (__bf16)fsub(r1, (float)md);
    }
    __bf16* img = reinterpret_cast<__bf16*>(tiles);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q < nplanes) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 w = {pl[q][0], pl[q][1], pl[q][2], pl[q][3]};
            *reinterpret_cast<bf16x4*>(img + q * plane + tile_slot_bf16(hid + (size_t)(l - 1) * 65536, 8, row0 + rr, in0)) = w;  // W_l
            *reinterpret_cast<bf16x4*>(&s_t[q][rr][in0]) = w;
        }
    }
    __syncthreads();
    constexpr int RG = kOneRows < 8 ? kOneRows : 8, NG = kOneRows / RG;  // outputs per store (a 16-byte slot holds 8), stores per input
    for (int idx = threadIdx.x; idx < nplanes * NG * 256; idx += kOneThreads) {
        const int q = idx / (NG * 256), gq = (idx / 256) % NG, in = idx & 255;
        typedef __bf16 bf16xr __attribute__((ext_vector_type(RG)));
        bf16xr w;
#pragma unroll
        for (int r = 0; r < RG; ++r) w[r] = s_t[q][gq * RG + r][in];
        *reinterpret_cast<bf16xr*>(img + q * plane + tile_slot_bf16(hid + (size_t)(l + 1) * 65536, 8, in, row0 + gq * RG)) = w;  // W_l^T
    }
}

// One float4 of parameters per thread (p, g, m, v: four 16-byte loads in flight per lane); the (<= 1024) norm partials are
// re-reduced by every block (a device-wide "last block" finalisation would serialise one memory-side atomic per block).
// GATHER (the RS + AG all-reduce): the reduced gradient and its norm partials are read from every rank's shard area (peer reads
// over xGMI behind the phase-1 hand-shake) instead of from `g` / `partials`; `g` still receives the clipped gradient.
template <bool GATHER>
__global__ __launch_bounds__(256) void clip_adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, rlx_adamw_params a,
                                                         const double* __restrict__ partials, int nparts,
                                                         const AdamScalars* __restrict__ scalars,
                                                         float* __restrict__ stats, int* __restrict__ state,
                                                         rlx_mlp_layout lay, float* __restrict__ tiles, unsigned* __restrict__ seq_inc,
                                                         const int* status, GatherSrc gsrc, PeerWait wait) {
    __shared__ double s_red[4];
    __shared__ float s_coef;
    __shared__ int s_skip;
    __shared__ AdamScalars s_sc;
    size_t goff = 0;
    if constexpr (GATHER) {
        peer_handshake(wait);
        goff = (size_t)((*gsrc.seq + 1u) & 1u) * (size_t)gsrc.slot_stride;
    }
    // issue this thread's loads before the norm is known: they do not depend on it
    const bool vec = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                                       reinterpret_cast<uintptr_t>(v)) % 16 == 0);
    const long long n4 = vec ? n / 4 : 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float4 p4 = {0, 0, 0, 0}, g4 = p4, m4 = p4, v4 = p4;
    if (i0 < n4) {
        p4 = reinterpret_cast<const float4*>(p)[i0];
        if constexpr (GATHER) g4 = gather_f4(gsrc, goff, i0);
        else g4 = reinterpret_cast<const float4*>(g)[i0];
        m4 = reinterpret_cast<const float4*>(m)[i0];
        v4 = reinterpret_cast<const float4*>(v)[i0];
    }
    // the norm partials (<= kMaxParts = 4 x 256): all of a lane's loads in flight together, added in ascending order
    double acc[1] = {0.0};
    if constexpr (GATHER) {  // world x nparts partials, rank-major: the same tree on every rank -> the same norm on every rank
        const int total = gsrc.world * gsrc.nparts;
        const size_t poff = (size_t)((*gsrc.seq + 1u) & 1u) * (size_t)gsrc.nparts;
        double pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = min((int)threadIdx.x + u * 256, total - 1), owner = idx / gsrc.nparts;
            const double* base = gsrc.parts[0];
#pragma unroll
            for (int r = 1; r < kMaxRanks; ++r)
                if (r == owner) base = gsrc.parts[r];
            pv[u] = base[poff + (idx - owner * gsrc.nparts)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((int)threadIdx.x + u * 256 < total) acc[0] += pv[u];
    } else {
        double pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = partials[min((int)threadIdx.x + u * 256, nparts - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((int)threadIdx.x + u * 256 < nparts) acc[0] += pv[u];
    }
    // the step's scalars: one lane per block forms them (in parallel across blocks, hidden behind the partial-norm reduction;
    // formed once in the reduce launch they sat on ITS critical path: +2.7 us measured)
    if (threadIdx.x == 64) form_scalars(a, state != nullptr ? state[0] + 1 : a.step, &s_sc);  // state[0] is stable for the whole launch
    (void)scalars;
    block_sum<1>(acc, s_red);
    if (threadIdx.x == 0) {
        const float total_norm = (float)sqrt(acc[0]);
        float coef = 1.f;
        if (a.max_grad_norm > 0.f) coef = fminf(a.max_grad_norm / (total_norm + 1e-6f), 1.0f);  // clip_grad_norm_
        s_coef = coef;
        // a peer wait of this step's all-reduce timed out: the sums are garbage -- never apply them (the host raises)
        // (`status` is the word peer_poll sets on a timeout, possibly in THIS launch: no restrict / readonly promise, a fresh load)
        s_skip = !isfinite(total_norm) || (status != nullptr && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        if (blockIdx.x == 0) {
            stats[0] = total_norm;
            stats[1] = s_skip ? 0.f : 1.f;
            if (state != nullptr) state[1] = s_skip ? 0 : 1;
            if (seq_inc != nullptr) *seq_inc += 1u;  // this all-reduce is consumed: the next staging launch uses the other buffer
        }
    }
    __syncthreads();
    const float coef = s_coef;
    const bool skip = s_skip != 0;
    const int lane = threadIdx.x & 63;
    for (long long iw = i0 - lane; iw < n4; iw += stride) {  // wave-uniform trip count: the tile scatter below shuffles across lanes
        const long long i = iw + lane;
        const bool live = i < n4;
        if (live && i != i0) {
            p4 = reinterpret_cast<const float4*>(p)[i];
            if constexpr (GATHER) g4 = gather_f4(gsrc, goff, i);
            else g4 = reinterpret_cast<const float4*>(g)[i];
            m4 = reinterpret_cast<const float4*>(m)[i];
            v4 = reinterpret_cast<const float4*>(v)[i];
        }
        update_f4(p, g, m, v, iw, lane, live, p4, g4, m4, v4, coef, skip, s_sc, a, lay, tiles);
    }
    for (long long i = n4 * 4 + i0; i < n; i += stride) {  // scalar tail / unaligned fallback
        const int grp = group_of(a, i);
        float pi = p[i], gi = g[i], mi = m[i], vi = v[i];
        adamw_elem(pi, gi, mi, vi, coef, grp, skip, s_sc);
        g[i] = gi;
        if (skip || grp < 0) continue;
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
        if (tiles != nullptr) {
            if (a.tiles_bf16 == 1) tile_scatter<1>(lay, tiles, i, pi);
            else if (a.tiles_bf16 == 2) tile_scatter<2>(lay, tiles, i, pi);
            else tile_scatter<0>(lay, tiles, i, pi);
        }
    }
}

// The whole optimizer step as ONE launch (rlx_adamw_params.sync_words): slab sum + norm partial, a device-wide exchange of the
// partials, clip + AdamW on the gradient still in registers.  What it saves against the two launches above: a launch boundary
// (drain, cache write-back, ramp), the reduced gradient's trip through memory, and p / m / v are in flight while the norm forms.
// The exchange is NOT a counter: a block publishes its f64 partial as two 64-bit words (epoch << 32 | half) in its own slot and
// every block polls all slots (one per thread) -- no same-address atomics (the device-wide-barrier attempt of round 2 serialised
// 280 of them: 33-49 us), one store and one load round trip to the memory-side cache (4.0 us measured), and the words carry
// their own validity (64-bit accesses are single-copy atomic), so no fence orders data against a flag.  The epoch lives in
// sync[0]; block 0 advances it after ITS poll completed, i.e. after every block has read it.  Needs every block resident
// (launch_reduce_clip_adamw checks the occupancy once) -- a poll that expires anyway (2 s) reports a non-finite norm and skips.
//
// Blocks are 512 threads = 2048 consecutive parameters, dealt out segment by segment (SegPlan): a 256 x 256 hidden matrix of the
// MLP is a segment of its own, so that a block holds EIGHT WHOLE ROWS of it.  That is what the weight image wants: the forward
// image takes a lane's four consecutive inputs as one 8-byte store, and the transposed image -- eight consecutive OUTPUTS per
// 16-byte fragment slot, i.e. one value from each of eight rows -- is assembled through LDS and written as full 16-byte stores.
// (The generic scatter writes the transposed image as 2-byte stores 16 bytes apart: 3.1 of the first version's 17.0 us.)
// The gradient exchange INSIDE the one-launch optimizer step (world_size > 1, one rank per GPU; rlx_xgmi_clip_adamw_step with
// rlx_adamw_params.sync_words).  The four-launch chain (stage -> hand-shake -> reduce(-scatter) -> gather + clip + AdamW) moved the
// DATA by peer reads behind flag hand-shakes; here the data travels the way round 5's norm partials do: every value is PUSHED into
// the consumer's fine-grained buffer as a self-validating 64-bit word, epoch << 32 | float bits -- a single-copy-atomic store that
// carries its own validity -- and the consumer polls the payload itself in its LOCAL memory.  No stage launch, no hand-shake
// launch, no flag, no fence, no remote read; one kernel per rank and step:
//
//   every block   sums its 2048 parameters' slabs (the gradient stays in registers)
//   block b is OWNED by rank b * W / nblk (contiguous blocks = contiguous parameters: the reduce-scatter's shards)
//     non-owner    pushes its float4s into the owner's inbox row [its rank]
//     owner        polls the W - 1 inbox rows of its own buffer, adds the W gradients in RANK order (one sum per element,
//                  formed once, on the owner: every replica receives the same bits), scales by 1 / W, publishes the block's
//                  squared-norm partial to every rank's slot b and pushes the reduced float4s into every other rank's gather area
//     non-owner    polls its float4s of its own gather area
//   every block   polls all nblk partial slots of its own buffer (wave 0, fixed order: the same norm in every block of every
//                  rank), then clip + AdamW + weight-image refresh as on one GPU.
//
// Two one-way trips over xGMI on the critical path (contribution in, reduced value out) and (W - 1) / W x 2 x 8 bytes per parameter
// and direction on a rank's links.  Buffers are reused every step with the next epoch: a rank can only push step s + 1 after its
// step-s kernel ended, i.e. after it polled every step-s partial, i.e. after every owner consumed every step-s inbox row; a
// gather word of step s + 1 needs all ranks' step-s + 1 contributions, which follow their step-s kernels in stream order.
// Protocol model with torn-word and stale-epoch negative controls: tests/test_xgmi_protocol_model.py.
struct XchgPeers {
    unsigned long long* inbox[kMaxRanks];   // rank q's inbox: [world rows][n_cap words]; this rank writes row [rank]
    unsigned long long* gather[kMaxRanks];  // rank q's gather area: [n_cap words]
    unsigned long long* parts[kMaxRanks];   // rank q's norm-partial slots: [kMaxOneBlocks][2 words]
    long long n_cap;                        // words per inbox row (the communicator's n_max)
    int rank, world;                        // world <= 1: no exchange (the single-GPU launch)
    int self_alias;                         // timing / single-device emulation: every peer is this rank's own buffer (see below)
    int* status;                            // the communicator's time-out word (host: rlx_xgmi_status)
};
struct OneLaunchArgs {
    ReduceSrc src;
    float *p, *g, *m, *v;
    long long n4;
    float scale;
    rlx_adamw_params a;
    float* stats;
    int* state;
    rlx_mlp_layout lay;
    float* tiles;
    const int* status;
    unsigned long long* sync;
    DeferredScale dfr;
    SegPlan plan;
    long long poll_ticks;  // bound of the exchange's poll in 100 MHz ticks (2 s; RLX_ONE_LAUNCH_POLL_MS shortens it for the expiry tests)
    XchgPeers x;
};
// One tagged float4: four 64-bit words, each a single-copy-atomic store / load of its own.  Word e of float4 i lives at
// xword(i) + 64 e: the four components of 64 consecutive float4s are four runs of 64 words, so that ONE store / load instruction of
// a wave covers 512 contiguous bytes (component-major inside a 64-float4 group; lane-major put 8 bytes into every 32: sixteen
// partially written lines per instruction on memory that no cache merges).
__device__ __forceinline__ size_t xword(long long i) { return (size_t)(i & ~63ll) * 4 + (size_t)(i & 63ll); }
template <int SCOPE>
__device__ __forceinline__ void push_f4(unsigned long long* dst, float4 v, unsigned e1) {
    const unsigned long long t = (unsigned long long)e1 << 32;
    __hip_atomic_store(dst + 0, t | (unsigned long long)__float_as_uint(v.x), __ATOMIC_RELAXED, SCOPE);
    __hip_atomic_store(dst + 64, t | (unsigned long long)__float_as_uint(v.y), __ATOMIC_RELAXED, SCOPE);
    __hip_atomic_store(dst + 128, t | (unsigned long long)__float_as_uint(v.z), __ATOMIC_RELAXED, SCOPE);
    __hip_atomic_store(dst + 192, t | (unsigned long long)__float_as_uint(v.w), __ATOMIC_RELAXED, SCOPE);
}
// Poll `nsrc` tagged float4s (this lane's, one per source) until every word carries epoch e1; false: the bound expired or the sticky
// word was set meanwhile.  All loads of a round are issued before the first check.
template <int SCOPE, int MAXSRC>
__device__ __forceinline__ bool poll_f4(unsigned long long* const (&src)[MAXSRC], unsigned pending, unsigned e1, const unsigned long long* sticky,
                                        long long ticks, float4 (&out)[MAXSRC]) {  // pending: bit q set = src[q] is to be polled
    const long long t0 = wall_clock64();
    for (int spins = 0; pending != 0; ++spins) {
        unsigned long long w[MAXSRC][4];
#pragma unroll
        for (int q = 0; q < MAXSRC; ++q)
            if (pending & (1u << q)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) w[q][e] = __hip_atomic_load(src[q] + 64 * e, __ATOMIC_RELAXED, SCOPE);
            }
        const unsigned long long stick = __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int q = 0; q < MAXSRC; ++q)
            if ((pending & (1u << q)) && (unsigned)(w[q][0] >> 32) == e1 && (unsigned)(w[q][1] >> 32) == e1 &&
                (unsigned)(w[q][2] >> 32) == e1 && (unsigned)(w[q][3] >> 32) == e1) {
                out[q] = float4{__uint_as_float((unsigned)w[q][0]), __uint_as_float((unsigned)w[q][1]), __uint_as_float((unsigned)w[q][2]),
                                __uint_as_float((unsigned)w[q][3])};
                pending &= ~(1u << q);
            }
        if (pending == 0) break;
        if (stick != 0ull) return false;
        if (spins > 8) {
            if (wall_clock64() - t0 > ticks) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return true;
}
template <bool DEFER, int SB, bool XCHG = false>
__global__ __launch_bounds__(kOneThreads, 4) void reduce_clip_adamw_one_launch(OneLaunchArgs k) {  // (4 waves per SIMD = two blocks per CU: <= 128 VGPRs)
    touch_kernargs<(int)sizeof(OneLaunchArgs)>();
    __shared__ double s_red[kOneThreads / 64];
    __shared__ AdamScalars s_sc;
    __shared__ double s_tot;
    __shared__ int s_exp;
    __shared__ int s_xexp;  // XCHG: a data poll of this block expired
    // partial slots and payload words live in the communicator's fine-grained buffers and are written by OTHER devices: system scope
    constexpr int SCOPE = XCHG ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT;
    __shared__ __bf16 s_t[3][kOneRows][256 + 8];  // [plane][row of the block][input]: the transposed image's staging (12.4 KiB)
    const rlx_adamw_params& a = k.a;
    float* __restrict__ p = k.p;
    float* __restrict__ g = k.g;
    float* __restrict__ m = k.m;
    float* __restrict__ v = k.v;
    unsigned long long* sync = k.sync;
#ifdef RLX_ONE_LAUNCH_STAMPS  /* timing experiment: wall-clock (100 MHz) stamps of every block behind the slots */
#define RLX_OL_STAMP(q) do { if (threadIdx.x == 0) sync[kSyncSlot0 + kSlotStride * kMaxOneBlocks + 8 * blockIdx.x + (q)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define RLX_OL_STAMP(q) do { } while (0)
#endif
    RLX_OL_STAMP(0);
    const unsigned e1 = (unsigned)__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    // sync[1], sticky: the exchange of an EARLIER launch on these words expired.  Every launch since then skips AS A WHOLE -- each
    // block reads the word here, before anything else, and it was set before this launch started -- so slots left behind by
    // blocks that arrived late (with tags of an epoch they should not have) are never taken for partials, and the parameters
    // stay exactly as the last complete step left them until the host has seen the word (ops.check_adamw_sync: two launches from
    // then on).
    const bool poisoned = __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    // step count: the two-launch form folds the previous call's "applied" flag in its first launch; here every wave folds it
    // for itself (both words are stable until block 0 writes them back, which it does behind the exchange)
    int st0 = 0, st1 = 0;
    if (k.state != nullptr) st0 = k.state[0], st1 = k.state[1];
    const int steps_done = st0 + (st1 != 0 ? 1 : 0);
    const SegBlock sb = seg_block(k.plan);
    const int lane = threadIdx.x & 63;
    const long long i = sb.i;
    const bool live = sb.live;
    float4 p4 = {0, 0, 0, 0}, g4 = p4, m4 = p4, v4 = p4;
    if (live) {
        if constexpr (DEFER) g4 = sum_slab_groups_f4<SB>(reinterpret_cast<const float4*>(k.src.base[0]), i, k.n4, k.src.nslab, k.dfr);
        else g4 = sum_slabs_f4<SB>(reinterpret_cast<const float4*>(k.src.base[0]) + i, k.n4, k.src.nslab);
        if constexpr (!XCHG) {  // (under the exchange they are requested behind the data polls: registers)
            p4 = reinterpret_cast<const float4*>(p)[i];
            m4 = reinterpret_cast<const float4*>(m)[i];
            v4 = reinterpret_cast<const float4*>(v)[i];
            g4.x *= k.scale; g4.y *= k.scale; g4.z *= k.scale; g4.w *= k.scale;
        }
    }
    // where this launch's partial slots are: the caller's sync words (one GPU), or this rank's slots in its exchange buffer
    unsigned long long* slots_mine = XCHG ? k.x.parts[k.x.rank] : sync + kSyncSlot0;
    bool is_owner = true;
    if constexpr (XCHG) {
        const XchgPeers& x = k.x;
        const int W = x.world, me = x.rank;
        const int owner = (int)(((long long)blockIdx.x * W) / gridDim.x);
        is_owner = owner == me;
        if (threadIdx.x == 0) s_xexp = 0;
        __syncthreads();
        bool ok = true;
        const size_t w4 = xword(i);  // this lane's first word in a row
        // self_alias 2 (timing): this rank's blocks b, b + per, b + 2 per, ... play ranks 0, 1, 2, ...'s copies of owned block b -- the
        // W - 1 contributions an owner waits for come from OTHER workgroups running concurrently, and its answers go to theirs,
        // like over the links (values are then sums of different blocks' gradients: scratch buffers only)
        const int per = (int)gridDim.x / W, vb = (int)blockIdx.x % per, vp = (int)blockIdx.x / per;
        const bool timing = x.self_alias == 2;
        if (timing && !poisoned) {
            if (!is_owner) {
                const SegBlock ob = seg_block(k.plan, vb);  // the owned block this workgroup is rank vp's copy of
                if (ob.live) push_f4<SCOPE>(x.inbox[me] + (size_t)vp * (size_t)x.n_cap + xword(ob.i), g4, e1);
            }
        }
        if (live && !poisoned) {
            if (!is_owner) {
                if (!timing) push_f4<SCOPE>(x.inbox[owner] + (size_t)me * (size_t)x.n_cap + w4, g4, e1);  // contribution -> the owner's inbox row [me]
            } else {
                if (x.self_alias == 1)  // single-device emulation, exact: this block also plays the W - 1 peers that contribute to it
                    for (int p = 0; p < W; ++p)
                        if (p != me) push_f4<SCOPE>(x.inbox[me] + (size_t)p * (size_t)x.n_cap + w4, g4, e1);
                // rank order: ranks 0 .. me - 1 from the inbox, this rank's own, then me + 1 .. W - 1 -- four ranks' rows per
                // round of loads (the rows arrive concurrently; one row per round made W - 1 dependent trips to memory: 14 us of
                // the W = 8 step against 9 this way; all seven at once does not fit 128 registers)
                float4 sum{0.f, 0.f, 0.f, 0.f};
                for (int base = 0; base < W; base += 4) {
                    unsigned long long* src[4];
                    float4 got[4];
                    unsigned mask = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = base + u;
                        src[u] = x.inbox[me] + (size_t)(r < W ? r : 0) * (size_t)x.n_cap + w4;
                        got[u] = float4{0.f, 0.f, 0.f, 0.f};
                        if (r < W && r != me) mask |= 1u << u;
                    }
                    if (mask != 0) ok = poll_f4<SCOPE, 4>(src, mask, e1, &sync[1], k.poll_ticks, got) && ok;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = base + u;
                        if (r < W) {
                            const float4 t = r == me ? g4 : got[u];
                            if (r == 0) sum = t;
                            else { sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w; }
                        }
                    }
                }
                g4 = sum;
            }
            if (!is_owner && x.self_alias == 1) {  // ... and the owner this block's contribution went to: W copies of its own gradient
                float4 sum = g4;
                for (int p = 1; p < W; ++p) { sum.x += g4.x; sum.y += g4.y; sum.z += g4.z; sum.w += g4.w; }
                g4 = sum;
            }
            if (is_owner || x.self_alias == 1) {
                g4.x *= k.scale; g4.y *= k.scale; g4.z *= k.scale; g4.w *= k.scale;
                if (is_owner && x.self_alias == 0) {
                    for (int p = 0; p < W; ++p)
                        if (p != me) push_f4<SCOPE>(x.gather[p] + w4, g4, e1);  // reduced value -> every other rank's gather area
                } else if (!is_owner) {
                    push_f4<SCOPE>(x.gather[me] + w4, g4, e1);  // (exact emulation: the answer the remote owner would have pushed)
                }
            }
            if (is_owner && timing) {  // the answers to ranks 1 .. W - 1's copies of this block: their own gather words
                for (int p = 1; p < W; ++p) {
                    const SegBlock cb = seg_block(k.plan, vb + p * per);
                    if (cb.live) push_f4<SCOPE>(x.gather[me] + xword(cb.i), g4, e1);
                }
            }
            if (!is_owner) {
                unsigned long long* src[1] = {x.gather[me] + w4};
                float4 got[1] = {float4{0.f, 0.f, 0.f, 0.f}};
                ok = poll_f4<SCOPE, 1>(src, 1u, e1, &sync[1], k.poll_ticks, got) && ok;
                g4 = got[0];
            }
        }
        if (!ok) atomicOr(&s_xexp, 1);
        if (live) {  // in flight while the norm forms
            p4 = reinterpret_cast<const float4*>(p)[i];
            m4 = reinterpret_cast<const float4*>(m)[i];
            v4 = reinterpret_cast<const float4*>(v)[i];
        }
    }
    // the squared-norm partial of this block's 2048 REDUCED, scaled values: every block on one GPU, the owner (once per block
    // across the job) under the exchange -- which publishes it to every rank's slot
    const bool publishes = !XCHG || is_owner || k.x.self_alias;
    double acc[1] = {(double)g4.x * (double)g4.x + (double)g4.y * (double)g4.y + (double)g4.z * (double)g4.z + (double)g4.w * (double)g4.w};
    if (!live) acc[0] = 0.0;
    RLX_OL_STAMP(1);
    if (publishes) block_sum<1>(acc, s_red);
    else __syncthreads();
    if (threadIdx.x == 0 && publishes) {
        double part = acc[0];
        if constexpr (XCHG) {
            if (s_xexp != 0) part = __longlong_as_double(0x7ff8000000000000ll);  // an expired data poll: a NaN partial -> every rank skips
        }
        const unsigned long long bits = (unsigned long long)__double_as_longlong(part), tag = (unsigned long long)e1 << 32;
        if constexpr (XCHG) {
            for (int p = 0; p < k.x.world; ++p) {
                if (k.x.self_alias && p != k.x.rank) continue;
                unsigned long long* slot = k.x.parts[p] + kSlotStride * (size_t)blockIdx.x;
                __hip_atomic_store(slot, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, SCOPE);
                __hip_atomic_store(slot + 1, tag | (bits >> 32), __ATOMIC_RELAXED, SCOPE);
            }
        } else {
            unsigned long long* slot = slots_mine + kSlotStride * (size_t)blockIdx.x;
            __hip_atomic_store(slot, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, SCOPE);
            __hip_atomic_store(slot + 1, tag | (bits >> 32), __ATOMIC_RELAXED, SCOPE);
        }
    }
    RLX_OL_STAMP(2);
    // Wave 0 polls for the block: lane L takes blocks L, L + 64, ... (up to 8), adds them in a fixed order (the same in every block ->
    // the same norm everywhere) and forms the step's scalars while its first polls are in flight (they cost ~1 us of f64 math on
    // one lane: in front of the publication they delayed every block's partial).  Measured alternatives, alternating in one
    // process (tools/ab_opt_libs.py, profiles/r05_one_launch_optimizer_step_ab.txt): every wave polling for itself (no barrier behind
    // the exchange, 8 x the poll traffic) +0.6 us; slots 256 bytes apart: no difference; one 16-byte system-scope load per slot
    // instead of two 8-byte agent-scope ones +1.2 us; 256-thread blocks (all CUs loading) +0.5 us, 1024-thread blocks +2.3 us.
    const int nparts = gridDim.x;
    double pv[8];
    unsigned pending = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        pv[u] = 0.0;
        if (lane + 64 * u < nparts) pending |= 1u << u;
    }
    AdamScalars& sc = s_sc;
    bool expired = poisoned;
    if (poisoned) pending = 0;
    if (threadIdx.x < 64) {
        const long long t0 = wall_clock64();
        for (int spins = 0; pending != 0; ++spins) {
            unsigned long long lo[8], hi[8];
            // (every polling lane reads the sticky word with its slots -- one address, one request per wave, no latency of its own:
            // a block whose poll expires sets it, and a block still polling then gives up with it instead of completing on partials
            // it already holds and applying an update the expired block skipped)
            const unsigned long long stick = __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (pending & (1u << u)) {
                    const unsigned long long* slot = slots_mine + kSlotStride * (size_t)(lane + 64 * u);
                    lo[u] = __hip_atomic_load(slot, __ATOMIC_RELAXED, SCOPE);
                    hi[u] = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, SCOPE);
                }
            if (spins == 0 && lane == 0) form_scalars(a, k.state != nullptr ? steps_done + 1 : a.step, &sc);  // (behind the first polls' latency)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if ((pending & (1u << u)) && (unsigned)(lo[u] >> 32) == e1 && (unsigned)(hi[u] >> 32) == e1) {
                    pv[u] = __longlong_as_double((long long)((hi[u] << 32) | (lo[u] & 0xffffffffull)));
                    pending &= ~(1u << u);
                }
            if (stick != 0ull) {
                expired = true;
                break;
            }
            if (pending != 0 && spins > 16) {
                if (wall_clock64() - t0 > k.poll_ticks) {  // (2 s at 100 MHz: a block of this launch never became resident)
                    expired = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    RLX_OL_STAMP(3);
    double tot = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) tot += pv[u];
    tot = wave_sum(tot);
    const int wave_expired = __any(expired ? 1 : 0);
    if (threadIdx.x == 0) s_tot = tot, s_exp = (wave_expired != 0 || (XCHG && s_xexp != 0)) ? 1 : 0;  // (wave 0 polled for the block)
    __syncthreads();
    tot = s_tot;
    const bool any_expired = s_exp != 0;
    const float total_norm = any_expired ? __builtin_nanf("") : (float)sqrt(tot);
    float coef = 1.f;
    if (a.max_grad_norm > 0.f) coef = fminf(a.max_grad_norm / (total_norm + 1e-6f), 1.0f);  // clip_grad_norm_
    const bool skip = !isfinite(total_norm) ||
                      (k.status != nullptr && __hip_atomic_load(k.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
    if (threadIdx.x == 0 && any_expired && !poisoned) {
        // the sticky word first (pollers watch it), then this block's own slot again with a NaN payload: a block that becomes resident
        // only now and finds every slot published still forms a non-finite norm and skips with the rest
        __hip_atomic_store(&sync[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long nan_bits = 0x7ff8000000000000ull, tag = (unsigned long long)e1 << 32;
        unsigned long long* slot = slots_mine + kSlotStride * (size_t)blockIdx.x;
        __hip_atomic_store(slot, tag | (nan_bits & 0xffffffffull), __ATOMIC_RELAXED, SCOPE);
        __hip_atomic_store(slot + 1, tag | (nan_bits >> 32), __ATOMIC_RELAXED, SCOPE);
        if constexpr (XCHG) {  // a peer never delivered: the communicator's time-out word (the host raises: rlx_xgmi_status)
            if (k.x.status != nullptr) __hip_atomic_store(k.x.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) {  // (ONE writer of the step's stats: an expiry elsewhere is the sticky word's to report)
        k.stats[0] = total_norm;
        k.stats[1] = skip ? 0.f : 1.f;
        if (k.state != nullptr) k.state[0] = steps_done, k.state[1] = skip ? 0 : 1;
        __hip_atomic_store(&sync[0], (unsigned long long)e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RLX_OL_STAMP(4);
    apply_seg_block(p, g, m, v, sb, p4, g4, m4, v4, coef, skip, sc, a, k.lay, k.tiles, s_t);
    RLX_OL_STAMP(6);
#undef RLX_OL_STAMP
}

// clip_adamw_kernel on the SegPlan's blocks (2048 parameters, hidden matrices as segments of their own -> the weight image's fast
// stores, see reduce_clip_adamw_one_launch): the second launch of the two-launch form and of the xGMI exchange whenever the
// buffers are float4-shaped.  The norm partials come from the launch in front, so the block shape here is free.
struct ClipSegArgs {
    float *p, *g, *m, *v;
    rlx_adamw_params a;
    const double* partials;
    int nparts;
    float* stats;
    int* state;
    rlx_mlp_layout lay;
    float* tiles;
    unsigned* seq_inc;
    const int* status;
    GatherSrc gsrc;
    PeerWait wait;
    SegPlan plan;
};
template <bool GATHER>
__global__ __launch_bounds__(kOneThreads) void clip_adamw_seg_kernel(ClipSegArgs k) {
    touch_kernargs<(int)sizeof(ClipSegArgs)>();
    __shared__ double s_red[kOneThreads / 64];
    __shared__ float s_coef;
    __shared__ int s_skip;
    __shared__ AdamScalars s_sc;
    __shared__ __bf16 s_t[3][kOneRows][256 + 8];
    const rlx_adamw_params& a = k.a;
    size_t goff = 0;
    if constexpr (GATHER) {
        peer_handshake(k.wait);
        goff = (size_t)((*k.gsrc.seq + 1u) & 1u) * (size_t)k.gsrc.slot_stride;
    }
    const SegBlock sb = seg_block(k.plan);
    float4 p4 = {0, 0, 0, 0}, g4 = p4, m4 = p4, v4 = p4;
    if (sb.live) {  // issued before the norm is known: they do not depend on it
        p4 = reinterpret_cast<const float4*>(k.p)[sb.i];
        if constexpr (GATHER) g4 = gather_f4(k.gsrc, goff, sb.i);
        else g4 = reinterpret_cast<const float4*>(k.g)[sb.i];
        m4 = reinterpret_cast<const float4*>(k.m)[sb.i];
        v4 = reinterpret_cast<const float4*>(k.v)[sb.i];
    }
    // the norm partials (<= kMaxParts = 2 x 512): all of a lane's loads in flight together, added in ascending order
    double acc[1] = {0.0};
    {
        const int total = GATHER ? k.gsrc.world * k.gsrc.nparts : k.nparts;
        const size_t poff = GATHER ? (size_t)((*k.gsrc.seq + 1u) & 1u) * (size_t)k.gsrc.nparts : 0;
        double pv[kMaxParts / kOneThreads];
#pragma unroll
        for (int u = 0; u < kMaxParts / kOneThreads; ++u) {
            const int idx = min((int)threadIdx.x + u * kOneThreads, total - 1);
            if constexpr (GATHER) {  // world x nparts partials, rank-major: the same tree on every rank -> the same norm on every rank
                const int owner = idx / k.gsrc.nparts;
                const double* base = k.gsrc.parts[0];
#pragma unroll
                for (int r = 1; r < kMaxRanks; ++r)
                    if (r == owner) base = k.gsrc.parts[r];
                pv[u] = base[poff + (idx - owner * k.gsrc.nparts)];
            } else {
                pv[u] = k.partials[idx];
            }
        }
#pragma unroll
        for (int u = 0; u < kMaxParts / kOneThreads; ++u)
            if ((int)threadIdx.x + u * kOneThreads < total) acc[0] += pv[u];
    }
    if (threadIdx.x == 64) form_scalars(a, k.state != nullptr ? k.state[0] + 1 : a.step, &s_sc);  // state[0] is stable for the whole launch
    block_sum<1>(acc, s_red);
    if (threadIdx.x == 0) {
        const float total_norm = (float)sqrt(acc[0]);
        float coef = 1.f;
        if (a.max_grad_norm > 0.f) coef = fminf(a.max_grad_norm / (total_norm + 1e-6f), 1.0f);  // clip_grad_norm_
        s_coef = coef;
        s_skip = !isfinite(total_norm) || (k.status != nullptr && __hip_atomic_load(k.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
        if (blockIdx.x == 0) {
            k.stats[0] = total_norm;
            k.stats[1] = s_skip ? 0.f : 1.f;
            if (k.state != nullptr) k.state[1] = s_skip ? 0 : 1;
            if (k.seq_inc != nullptr) *k.seq_inc += 1u;  // this all-reduce is consumed: the next staging launch uses the other buffer
        }
    }
    __syncthreads();
    apply_seg_block(k.p, k.g, k.m, k.v, sb, p4, g4, m4, v4, s_coef, s_skip != 0, s_sc, a, k.lay, k.tiles, s_t);
}

template <bool DEFER>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ g, long long n, int nslab, float* __restrict__ out,
                                                        DeferredScale dfr) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s;
        if constexpr (DEFER) {
            s = sum_slab_groups(g, i, n, nslab, dfr);
        } else {
            s = g[i];
            for (int k = 1; k < nslab; ++k) s += g[(long long)k * n + i];
        }
        out[i] = s;
    }
}

__global__ void seq_inc_kernel(unsigned* seq) { *seq += 1u; }

}  // namespace

namespace step {
bool f32_split();  // ppo_step_f32x.hip
}

namespace opt {

int check_deferred(const rlx_adamw_params* p, int nslab, int64_t n, const char* who) {
    RLX_REQUIRE(p->deferred_groups >= 1 && nslab % p->deferred_groups == 0 && p->deferred_stride >= 0,
                "%s: %d slabs do not split into deferred_groups=%d", who, nslab, p->deferred_groups);
    for (int k = 0; k < 2; ++k)
        RLX_REQUIRE(p->deferred_range[k][0] >= 0 && p->deferred_range[k][0] <= p->deferred_range[k][1] && p->deferred_range[k][1] <= n,
                    "%s: deferred_range %d out of bounds", who, k);
    return RLX_OK;
}

int grid_for(long long n) {  // one float4 per thread up to kMaxParts blocks, grid-stride beyond
    return (int)std::max<long long>(1, std::min<long long>((n / 4 + 255) / 256 + 1, (long long)kMaxParts));
}

namespace {
// The kernel's copy of the parameters with the tile image's format resolved: an f32 image (tiles_bf16 == 0) is the three-plane one
// unless the process runs the exact-f32-MFMA launches (RLX_F32_EXACT_MFMA=1) -- the same switch rlx_mlp_pack_tiles follows.
rlx_adamw_params tile_format_resolved(const rlx_adamw_params* p, const float* tiles) {
    rlx_adamw_params k = *p;
    if (tiles != nullptr && k.tiles_bf16 == 0 && ::rlx::step::f32_split()) k.tiles_bf16 = 2;
    return k;
}
// Compute units a stream may dispatch to: hipExtStreamCreateWithCUMask streams (utils/streams.py) see a subset, and the
// one-launch kernel's residency bound has to be taken against THAT.  Not asked while the stream is capturing (a query on a
// capturing stream may invalidate the capture; graphs are captured on ordinary streams).
int stream_cus(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s != nullptr && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) {
        (void)hipGetLastError();
        return num_cu();
    }
    if (s == nullptr) return num_cu();
    uint32_t mask[32] = {0};
    if (hipExtStreamGetCUMask(s, 32, mask) != hipSuccess) {
        (void)hipGetLastError();
        return num_cu();
    }
    int n = 0;
    for (uint32_t w : mask) n += __builtin_popcount(w);
    return n > 0 ? std::min(n, num_cu()) : num_cu();
}
// Test hooks of the expiry path (tests/test_gpu_losses.py::test_one_launch_expiry_*), read at every launch: the poll's bound in ms
// (default 2000), and "launch even though the stream cannot hold every block" -- which is how a test gets a poll to expire.
long long one_launch_poll_ticks() {  // 100 MHz ticks
    const char* e = getenv("RLX_ONE_LAUNCH_POLL_MS");
    const long long ms = (e != nullptr && atoll(e) > 0) ? atoll(e) : 2000ll;
    return ms * 100000ll;
}
bool one_launch_ignores_capacity() {
    const char* e = getenv("RLX_ONE_LAUNCH_TEST_OVERSUBSCRIBE");
    return e != nullptr && e[0] == '1';
}
// How many blocks of the one-launch kernel can be resident together on ONE compute unit?  (Asked once per process.)
int one_launch_blocks_per_cu() {
    static const int capacity = [] {
        int best = 1 << 30;
        const void* kernels[4] = {(const void*)reduce_clip_adamw_one_launch<false, 9>, (const void*)reduce_clip_adamw_one_launch<false, 24>,
                                  (const void*)reduce_clip_adamw_one_launch<true, 9>, (const void*)reduce_clip_adamw_one_launch<true, 24>};
        for (const void* k : kernels) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kOneThreads, 0) != hipSuccess) return 0;
            best = std::min(best, per_cu);
        }
        return best;
    }();
    return capacity;
}
int xchg_blocks_per_cu() {
    static const int capacity = [] {
        int best = 1 << 30;
        const void* kernels[4] = {(const void*)reduce_clip_adamw_one_launch<false, 9, true>, (const void*)reduce_clip_adamw_one_launch<false, 24, true>,
                                  (const void*)reduce_clip_adamw_one_launch<true, 9, true>, (const void*)reduce_clip_adamw_one_launch<true, 24, true>};
        for (const void* k : kernels) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kOneThreads, 0) != hipSuccess) return 0;
            best = std::min(best, per_cu);
        }
        return best;
    }();
    return capacity;
}
// ... and on the compute units THIS stream may use (an unmasked stream: the whole device).  The bound is against an otherwise idle
// device: what another process or another stream holds is not visible from here -- for that case the kernel's poll is bounded
// and its expiry is sticky (see the kernel), and ranks of one job that share a GPU do not use the form at all.
int one_launch_capacity(hipStream_t s) { return std::min(one_launch_blocks_per_cu() * stream_cus(s), kMaxOneBlocks); }
// Blocks of 2048 parameters dealt out segment by segment; with a bf16 / three-plane weight image to keep, every 256 x 256 hidden
// matrix of the layout is a segment of its own (see the kernel).  nblk == 0: the layout is not one this plan understands.
SegPlan plan_segments(long long n4, const rlx_mlp_layout* lay, bool image) {
    SegPlan sp{};
    struct Mat { long long at4; int id; } mats[4];
    int nm = 0;
    if (image && lay != nullptr) {
        for (int y = 0; y < 2; ++y)
            for (int l = 1; l <= 2; ++l) {
                const long long off = lay->off_w[y][l];
                if (off < 0 || off % 4 != 0 || off / 4 + 16384 > n4) return sp;
                mats[nm++] = Mat{off / 4, y * 2 + (l - 1)};
            }
        std::sort(mats, mats + nm, [](const Mat& u, const Mat& w) { return u.at4 < w.at4; });
        for (int k = 1; k < nm; ++k)
            if (mats[k].at4 < mats[k - 1].at4 + 16384) return sp;  // overlapping matrices: not a layout of ours
    }
    long long cur = 0;
    auto push = [&](long long lo, long long hi, int id) {
        if (hi <= lo) return;
        sp.blk0[sp.nseg] = sp.nblk, sp.start4[sp.nseg] = lo, sp.end4[sp.nseg] = hi, sp.mat[sp.nseg] = id;
        sp.nblk += (int)((hi - lo + kOneThreads - 1) / kOneThreads);
        ++sp.nseg;
    };
    for (int k = 0; k < nm; ++k) {
        push(cur, mats[k].at4, -1);
        push(mats[k].at4, mats[k].at4 + 16384, mats[k].id);
        cur = mats[k].at4 + 16384;
    }
    push(cur, n4, -1);
    return sp;
}
int check_adamw_args(float* params, float* out, float* exp_avg, float* exp_avg_sq, int64_t n, const rlx_adamw_params* p, float* stats,
                     int32_t* step_state, rlx_mlp_layout& lay, float*& tiles) {
    RLX_REQUIRE(p != nullptr, "rlx_clip_adamw_step: NULL params struct");
    RLX_REQUIRE(n >= 0 && (p->step >= 1 || step_state != nullptr) && p->n_groups >= 0 && p->n_groups <= RLX_ADAMW_MAX_GROUPS,
                "rlx_clip_adamw_step: bad sizes (n=%lld step=%d groups=%d)", (long long)n, p->step, p->n_groups);
    RLX_REQUIRE(params && out && exp_avg && exp_avg_sq && stats, "rlx_clip_adamw_step: NULL argument");
    for (int k = 0; k < p->n_groups; ++k)
        RLX_REQUIRE(p->groups[k].begin >= 0 && p->groups[k].end <= n && p->groups[k].begin <= p->groups[k].end,
                    "rlx_clip_adamw_step: group %d range out of bounds", k);
    tiles = nullptr;
    if (p->tile_layout != nullptr && p->tiles != nullptr) {
        lay = *p->tile_layout;
        tiles = p->tiles;
        RLX_REQUIRE(lay.hidden == 256 && lay.obs_dim >= 1 && lay.obs_dim <= 64 && lay.n_params == n,
                    "rlx_clip_adamw_step: tile_layout does not describe these %lld parameters", (long long)n);
    }
    return RLX_OK;
}
}  // namespace

// The clip + AdamW launch behind a reduce (GATHER false: gradient and norm partials local) or behind a reduce-scatter (true: read
// from every rank's shard area): SegPlan blocks with the weight image's fast stores when the buffers are float4-shaped, else the
// general grid-stride kernel.
static int launch_clip_adamw(bool gather, float* params, float* out, float* exp_avg, float* exp_avg_sq, int64_t n, const rlx_adamw_params* p,
                             const double* partials, int nparts, const AdamScalars* scalars, float* stats, int32_t* step_state,
                             const rlx_mlp_layout& lay, float* tiles, unsigned* seq_inc, const int* status, const GatherSrc& gsrc,
                             const PeerWait& w, hipStream_t s) {
    const rlx_adamw_params k = tile_format_resolved(p, tiles);
    const bool vec = n % 4 == 0 && (reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(exp_avg) |
                                    reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(tiles)) % 16 == 0;
    if (vec && n / 4 / kOneThreads < (1ll << 30)) {
        const SegPlan sp = plan_segments(n / 4, &lay, tiles != nullptr && k.tiles_bf16 != 0);
        if (sp.nblk >= 1) {
            ClipSegArgs ca{};
            ca.p = params, ca.g = out, ca.m = exp_avg, ca.v = exp_avg_sq, ca.a = k, ca.partials = partials, ca.nparts = nparts, ca.stats = stats;
            ca.state = step_state, ca.lay = lay, ca.tiles = tiles, ca.seq_inc = seq_inc, ca.status = status, ca.gsrc = gsrc, ca.wait = w, ca.plan = sp;
            if (gather) hipLaunchKernelGGL(clip_adamw_seg_kernel<true>, dim3(sp.nblk), dim3(kOneThreads), 0, s, ca);
            else hipLaunchKernelGGL(clip_adamw_seg_kernel<false>, dim3(sp.nblk), dim3(kOneThreads), 0, s, ca);
            RLX_LAUNCH_CHECK();
            return RLX_OK;
        }
    }
    const int nblk = grid_for(n);
    if (gather)
        hipLaunchKernelGGL(clip_adamw_kernel<true>, dim3(nblk), dim3(256), 0, s, params, out, exp_avg, exp_avg_sq, (long long)n, k, partials, nparts,
                           scalars, stats, step_state, lay, tiles, seq_inc, status, gsrc, w);
    else
        hipLaunchKernelGGL(clip_adamw_kernel<false>, dim3(nblk), dim3(256), 0, s, params, out, exp_avg, exp_avg_sq, (long long)n, k, partials, nparts,
                           scalars, stats, step_state, lay, tiles, seq_inc, status, gsrc, w);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

// The optimizer step WITH the gradient exchange as one launch per rank (see XchgPeers).  *used = false: the plan does not allow it
// (buffers not float4-shaped, too many blocks for the stream / for `co_resident` ranks sharing this device) -- the caller falls
// back to its launch chain.  `grads`: this rank's split-K slabs; `out`: receives the clipped mean gradient.
int launch_exchange_clip_adamw_one_launch(float* params, const float* grads, int nslab, float* out, float* exp_avg, float* exp_avg_sq,
                                          int64_t n, const rlx_adamw_params* p, float* stats, int32_t* step_state, const ExchangeBuffers& xb,
                                          unsigned long long* xsync, int co_resident, long long timeout_ticks, int* status, hipStream_t s, bool* used) {
    *used = false;
    if (n == 0 || p->sync_words == nullptr) return RLX_OK;
    rlx_mlp_layout lay{};
    float* tiles = nullptr;
    if (int rc = check_adamw_args(params, out, exp_avg, exp_avg_sq, n, p, stats, step_state, lay, tiles)) return rc;
    const bool defer = p->deferred_scale != nullptr;
    if (defer)
        if (int rc = check_deferred(p, nslab, n, "rlx_xgmi_clip_adamw_step")) return rc;
    if (n % 4 != 0 || (n + 255) / 256 * 256 > xb.n_cap || n / 4 > (long long)kOneThreads * kMaxOneBlocks ||
        (reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(p->sync_words) |
         reinterpret_cast<uintptr_t>(tiles)) % 16 != 0)
        return RLX_OK;
    const rlx_adamw_params k = tile_format_resolved(p, tiles);
    const SegPlan sp = plan_segments(n / 4, &lay, tiles != nullptr && k.tiles_bf16 != 0);
    const int capacity = std::min(xchg_blocks_per_cu() * stream_cus(s), kMaxOneBlocks);
    if (sp.nblk < 1 || sp.nblk < xb.world || ((long long)sp.nblk * std::max(1, co_resident) > capacity && !one_launch_ignores_capacity()))
        return RLX_OK;
    OneLaunchArgs oa{};
    oa.src.nbase = 1, oa.src.nslab = nslab, oa.src.base[0] = grads;
    oa.p = params, oa.g = out, oa.m = exp_avg, oa.v = exp_avg_sq, oa.n4 = n / 4, oa.scale = p->grad_scale, oa.a = k;
    oa.stats = stats, oa.state = step_state, oa.lay = lay, oa.tiles = tiles, oa.status = status;
    oa.sync = xsync;  // (the communicator's epoch / sticky words; p->sync_words only says that the caller allows the form)
    oa.dfr = defer ? deferred_of(p) : DeferredScale{};
    oa.plan = sp;
    oa.poll_ticks = timeout_ticks;  // (a peer that died, not a block that is not resident: the communicator's bound)
    XchgPeers& x = oa.x;
    for (int r = 0; r < xb.world; ++r) x.inbox[r] = xb.inbox[r], x.gather[r] = xb.gather[r], x.parts[r] = xb.parts[r];
    x.n_cap = xb.n_cap, x.rank = xb.rank, x.world = xb.world, x.self_alias = (xb.self_alias == 2 && (sp.nblk % xb.world != 0 || xb.rank != 0)) ? 1 : xb.self_alias, x.status = status;
    const int per_group = defer ? nslab / p->deferred_groups : nslab;
#define RLX_XCHG(DEFER_, SB_) \
    hipLaunchKernelGGL((reduce_clip_adamw_one_launch<DEFER_, SB_, true>), dim3(sp.nblk), dim3(kOneThreads), 0, s, oa)
    if (defer) {
        if (per_group <= 10) RLX_XCHG(true, 9);
        else RLX_XCHG(true, 24);
    } else {
        if (per_group <= 10) RLX_XCHG(false, 9);
        else RLX_XCHG(false, 24);
    }
#undef RLX_XCHG
    RLX_LAUNCH_CHECK();
    *used = true;
    return RLX_OK;
}

int launch_reduce_clip_adamw(float* params, const ReduceSrc& src, float* out, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const rlx_adamw_params* p, float* stats, int32_t* step_state, void* workspace, size_t workspace_bytes,
                             const PeerWait* wait, unsigned* seq_inc, const int* status, hipStream_t s) {
    if (n == 0) return RLX_OK;
    rlx_mlp_layout lay{};
    float* tiles = nullptr;
    if (int rc = check_adamw_args(params, out, exp_avg, exp_avg_sq, n, p, stats, step_state, lay, tiles)) return rc;
    RLX_REQUIRE(workspace != nullptr, "rlx_clip_adamw_step: NULL workspace");
    if (workspace_bytes < rlx_adamw_workspace_bytes(n)) {
        set_error("rlx_clip_adamw_step: workspace too small");
        return RLX_ENOSPC;
    }
    double* partials = static_cast<double*>(workspace);
    const int nblk = grid_for(n);
    AdamScalars* scalars = reinterpret_cast<AdamScalars*>(static_cast<char*>(workspace) + scalars_offset());
    PeerWait w{};
    if (wait != nullptr) w = *wait;
    const bool defer = p->deferred_scale != nullptr && src.nbase == 1;  // (with peers the staging launch has applied it already)
    if (defer)
        if (int rc = check_deferred(p, src.nslab, n, "rlx_clip_adamw_step")) return rc;
    // one launch instead of two (see reduce_clip_adamw_one_launch): this rank's own slabs, no peer hand-shake, every block resident
    if (p->sync_words != nullptr && src.nbase == 1 && src.seq == nullptr && seq_inc == nullptr && w.world <= 1 && !w.fence && n % 4 == 0 &&
        n / 4 <= (long long)kOneThreads * kMaxOneBlocks &&
        (reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(src.base[0]) | reinterpret_cast<uintptr_t>(p->sync_words) |
         reinterpret_cast<uintptr_t>(tiles)) % 16 == 0) {
        const rlx_adamw_params k = tile_format_resolved(p, tiles);
        const SegPlan sp = plan_segments(n / 4, &lay, tiles != nullptr && k.tiles_bf16 != 0);
        if (sp.nblk >= 1 && (sp.nblk <= one_launch_capacity(s) || one_launch_ignores_capacity())) {
            unsigned long long* sync = reinterpret_cast<unsigned long long*>(p->sync_words);
            const int per_group = defer ? src.nslab / p->deferred_groups : src.nslab;
            const DeferredScale dfr = defer ? deferred_of(p) : DeferredScale{};
            OneLaunchArgs oa{};
            oa.src = src, oa.p = params, oa.g = out, oa.m = exp_avg, oa.v = exp_avg_sq, oa.n4 = n / 4, oa.scale = p->grad_scale, oa.a = k;
            oa.stats = stats, oa.state = step_state, oa.lay = lay, oa.tiles = tiles, oa.status = status, oa.sync = sync, oa.dfr = dfr, oa.plan = sp, oa.poll_ticks = one_launch_poll_ticks();
#define RLX_ONE_LAUNCH(DEFER_, SB_) \
    hipLaunchKernelGGL((reduce_clip_adamw_one_launch<DEFER_, SB_>), dim3(sp.nblk), dim3(kOneThreads), 0, s, oa)
            if (defer) {
                if (per_group <= 10) RLX_ONE_LAUNCH(true, 9);
                else RLX_ONE_LAUNCH(true, 24);
            } else {
                if (per_group <= 10) RLX_ONE_LAUNCH(false, 9);
                else RLX_ONE_LAUNCH(false, 24);
            }
#undef RLX_ONE_LAUNCH
            RLX_LAUNCH_CHECK();
            return RLX_OK;
        }
    }
    if (defer) {
        hipLaunchKernelGGL((grad_reduce_sqnorm<true, true>), dim3(nblk), dim3(256), 0, s, src, out, (long long)n, p->grad_scale, partials,
                           step_state, *p, scalars, w, deferred_of(p));
    } else {
        hipLaunchKernelGGL(grad_reduce_sqnorm<true>, dim3(nblk), dim3(256), 0, s, src, out, (long long)n, p->grad_scale, partials,
                           step_state, *p, scalars, w, DeferredScale{});
    }
    RLX_LAUNCH_CHECK();
    if (int rc = launch_clip_adamw(false, params, out, exp_avg, exp_avg_sq, n, p, partials, nblk, scalars, stats, step_state, lay, tiles, seq_inc,
                                   status, GatherSrc{}, PeerWait{}, s))
        return rc;
    return RLX_OK;
}

int launch_peer_wait(const PeerWait& wait, hipStream_t s) {
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, wait);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int rsag_parts(long long shard_n4) {
    return (int)std::max<long long>(1, std::min<long long>((shard_n4 + 255) / 256, (long long)(kMaxParts / kMaxRanks)));
}

int launch_reduce_scatter(const ReduceSrc& src, float* shard0, long long shard_stride, double* parts0, long long shard_lo4,
                          long long shard_n4, int nparts, float scale, int32_t* step_state, unsigned* seq_snapshot, const PeerWait* wait,
                          hipStream_t s) {
    PeerWait w{};
    if (wait != nullptr) w = *wait;
    hipLaunchKernelGGL(reduce_scatter_kernel, dim3(nparts), dim3(256), 0, s, src, shard0, shard_stride, parts0, nparts, shard_lo4,
                       shard_n4, scale, step_state, seq_snapshot, w);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

int launch_gather_clip_adamw(float* params, const GatherSrc& src, float* out, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const rlx_adamw_params* p, float* stats, int32_t* step_state, const PeerWait* wait, unsigned* seq_inc,
                             const int* status, hipStream_t s) {
    if (n == 0) return RLX_OK;
    rlx_mlp_layout lay{};
    float* tiles = nullptr;
    if (int rc = check_adamw_args(params, out, exp_avg, exp_avg_sq, n, p, stats, step_state, lay, tiles)) return rc;
    RLX_REQUIRE(n % 4 == 0 && src.world >= 2 && src.world * src.nparts <= kMaxParts, "gather clip_adamw: n %% 4 != 0 or bad shard plan");
    RLX_REQUIRE((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(exp_avg) |
                 reinterpret_cast<uintptr_t>(exp_avg_sq)) % 16 == 0, "gather clip_adamw: buffers must be 16-byte aligned");
    PeerWait w{};
    if (wait != nullptr) w = *wait;
    return launch_clip_adamw(true, params, out, exp_avg, exp_avg_sq, n, p, nullptr, 0, nullptr, stats, step_state, lay, tiles, seq_inc, status, src, w, s);
}

int launch_gather_only(const GatherSrc& src, float* out, int64_t n, const PeerWait* wait, unsigned* seq_inc, hipStream_t s) {
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(out != nullptr && n % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0, "gather: NULL / unaligned output");
    PeerWait w{};
    if (wait != nullptr) w = *wait;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, out, (long long)(n / 4), w);
    RLX_LAUNCH_CHECK();
    if (seq_inc != nullptr) {
        hipLaunchKernelGGL(seq_inc_kernel, dim3(1), dim3(1), 0, s, seq_inc);
        RLX_LAUNCH_CHECK();
    }
    return RLX_OK;
}

int launch_reduce_only(const ReduceSrc& src, float* out, int64_t n, float scale, void* workspace, size_t workspace_bytes,
                       const PeerWait* wait, unsigned* seq_inc, hipStream_t s) {
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(out && workspace && workspace_bytes >= rlx_adamw_workspace_bytes(n), "reduce: NULL or short workspace");
    PeerWait w{};
    if (wait != nullptr) w = *wait;
    rlx_adamw_params none{};
    hipLaunchKernelGGL(grad_reduce_sqnorm<false>, dim3(grid_for(n)), dim3(256), 0, s, src, out, (long long)n, scale,
                       static_cast<double*>(workspace), (int*)nullptr, none, (AdamScalars*)nullptr, w, DeferredScale{});
    RLX_LAUNCH_CHECK();
    if (seq_inc != nullptr) {
        hipLaunchKernelGGL(seq_inc_kernel, dim3(1), dim3(1), 0, s, seq_inc);
        RLX_LAUNCH_CHECK();
    }
    return RLX_OK;
}

}  // namespace opt
}  // namespace rlx

using namespace rlx;
using namespace rlx::opt;

extern "C" size_t rlx_adamw_sync_words(int64_t n) {
    (void)n;
    return (size_t)kSyncSlot0 + (size_t)kSlotStride * (size_t)kMaxOneBlocks + 8 * (size_t)kMaxOneBlocks;  // (+ room for the development stamps)
}

extern "C" size_t rlx_adamw_workspace_bytes(int64_t n) {
    (void)n;
    return (size_t)kMaxParts * sizeof(double) + 256;  // norm partials | AdamScalars
}

extern "C" int rlx_sum_slabs(const float* grads, int64_t n, int slabs, float* out, rlx_stream_t stream) {
    RLX_REQUIRE(n >= 0 && slabs >= 1, "rlx_sum_slabs: bad sizes");
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(grads && out, "rlx_sum_slabs: NULL argument");
    hipLaunchKernelGGL(sum_slabs_kernel<false>, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), grads, (long long)n,
                       slabs, out, DeferredScale{});
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_sum_slabs_deferred(const float* grads, int64_t n, int slabs, float* out, const rlx_adamw_params* p,
                                      rlx_stream_t stream) {
    RLX_REQUIRE(n >= 0 && slabs >= 1 && p != nullptr && p->deferred_scale != nullptr, "rlx_sum_slabs_deferred: bad sizes / no deferred_scale");
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(grads && out, "rlx_sum_slabs_deferred: NULL argument");
    if (int rc = check_deferred(p, slabs, n, "rlx_sum_slabs_deferred")) return rc;
    hipLaunchKernelGGL(sum_slabs_kernel<true>, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), grads, (long long)n,
                       slabs, out, deferred_of(p));
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_clip_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const rlx_adamw_params* p, float* stats, int32_t* step_state, void* workspace,
                                   size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(p != nullptr && p->grad_partials >= 1, "rlx_clip_adamw_step: NULL params struct or grad_partials < 1");
    ReduceSrc src{};
    src.base[0] = grads;
    src.nbase = 1;
    src.nslab = p->grad_partials;
    return launch_reduce_clip_adamw(params, src, grads, exp_avg, exp_avg_sq, n, p, stats, step_state, workspace, workspace_bytes,
                                    nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream));
}
