// opt_common.h -- pieces of the optimizer-step launches shared by adamw_clip.hip (reduce + clip + AdamW) and
// xgmi_allreduce.hip (the peer-read gradient all-reduce that feeds the same kernels when world_size > 1).
#pragma once

#include "rlx_common.h"

namespace rlx {
namespace opt {

constexpr int kMaxParts = 1024;
constexpr int kMaxRanks = RLX_XGMI_MAX_RANKS;

// Where the reduce kernel reads its addends: `nbase` base pointers, `nslab` slabs of n floats behind each -- the split-K
// slabs of this rank (nbase = 1) or one staged gradient per rank, read over xGMI (nslab = 1).  Summed in this order on every
// rank, so data-parallel replicas stay bit-identical.
struct ReduceSrc {
    const float* base[kMaxRanks];
    int nbase, nslab;
    // staged gradients live in one of two slots `slot_stride` floats apart, chosen by a device-side sequence number (the
    // all-reduce being served is *seq + 1): base[] then points at slot 0 of every rank.  seq == nullptr: no offset.
    long long slot_stride;
    const unsigned* seq;
};

// One float4 of the sum of `nslab` slabs (slab k at gp + k * n4).  The slabs were written by other XCDs' workgroups: every
// dependent batch of loads is one more trip to the memory-side cache (~1 us), so ALL slabs of this float4 (up to 1 + SB) are
// requested before the first add -- clamped, unconditional loads; the adds keep the ascending-slab order and skip the slots
// past nslab.  (Batches of 8 / 4 / 1 made 24 slabs seven dependent trips.)
#ifdef __HIPCC__
template <int SB = 24>  // slabs requested per batch beyond the first (24: the exact-f32 plan; 9: the 10-slab bf16 / three-plane plans)
__device__ __forceinline__ float4 sum_slabs_f4(const float4* __restrict__ gp, long long n4, int nslab) {
    float4 x[SB];
    float4 g = gp[0];
#pragma unroll
    for (int u = 0; u < SB; ++u) x[u] = gp[(long long)min(1 + u, nslab - 1) * n4];
#pragma unroll
    for (int u = 0; u < SB; ++u)
        if (1 + u < nslab) { g.x += x[u].x; g.y += x[u].y; g.z += x[u].z; g.w += x[u].w; }
    for (int k = 1 + SB; k < nslab; k += SB) {
#pragma unroll
        for (int u = 0; u < SB; ++u) x[u] = gp[(long long)min(k + u, nslab - 1) * n4];
#pragma unroll
        for (int u = 0; u < SB; ++u)
            if (k + u < nslab) { g.x += x[u].x; g.y += x[u].y; g.z += x[u].z; g.w += x[u].w; }
    }
    return g;
}
#endif

// Slab groups with a device-side scale per group (rlx_adamw_params.deferred_*: the decoupled rlx_ppo_step leaves the actor network's
// gradients in sum form and 1 / their denominator in its metric row).  The nslab slabs are `groups` consecutive groups; an element
// inside one of the two ranges is multiplied by scale[g * stride] as group g is added; outside them by 1.f (bit-identical to the
// plain sum for one group).  Shared by every kernel that collapses split-K slabs: the single-rank reduce, the xGMI staging launch
// and rlx_sum_slabs in front of an RCCL all-reduce.
struct DeferredScale {
    const float* scale;  // nullptr: none
    int stride, groups;
    long long range[2][2];
};
inline DeferredScale deferred_of(const rlx_adamw_params* p) {
    DeferredScale d{};
    if (p != nullptr && p->deferred_scale != nullptr) {
        d.scale = p->deferred_scale, d.stride = p->deferred_stride, d.groups = p->deferred_groups;
        for (int k = 0; k < 2; ++k) d.range[k][0] = p->deferred_range[k][0], d.range[k][1] = p->deferred_range[k][1];
    }
    return d;
}
int check_deferred(const rlx_adamw_params* p, int nslab, int64_t n, const char* who);  // adamw_clip.hip
#ifdef __HIPCC__
__device__ __forceinline__ float deferred_factor(const DeferredScale& d, long long idx, float sc) {
    const bool in = (idx >= d.range[0][0] && idx < d.range[0][1]) || (idx >= d.range[1][0] && idx < d.range[1][1]);
    return in ? sc : 1.f;
}
template <int SB = 24>
__device__ __forceinline__ float4 sum_slab_groups_f4(const float4* __restrict__ gp, long long i, long long n4, int nslab,
                                                     const DeferredScale& d) {
    const int per = nslab / d.groups;
    float4 g{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < d.groups; ++q) {
        const float4 x = sum_slabs_f4<SB>(gp + (long long)q * per * n4 + i, n4, per);
        const float sc = d.scale[(long long)q * d.stride];
        const float4 f{deferred_factor(d, 4 * i, sc), deferred_factor(d, 4 * i + 1, sc), deferred_factor(d, 4 * i + 2, sc),
                       deferred_factor(d, 4 * i + 3, sc)};
        if (q == 0) { g.x = x.x * f.x; g.y = x.y * f.y; g.z = x.z * f.z; g.w = x.w * f.w; }
        else { g.x += x.x * f.x; g.y += x.y * f.y; g.z += x.z * f.z; g.w += x.w * f.w; }
    }
    return g;
}
__device__ __forceinline__ float sum_slab_groups(const float* __restrict__ g0, long long i, long long n, int nslab, const DeferredScale& d) {
    const int per = nslab / d.groups;
    float g = 0.f;
    for (int q = 0; q < d.groups; ++q) {
        float x = g0[(long long)q * per * n + i];
        for (int k = 1; k < per; ++k) x += g0[((long long)q * per + k) * n + i];
        const float t = x * deferred_factor(d, i, d.scale[(long long)q * d.stride]);
        g = q == 0 ? t : g + t;
    }
    return g;
}
#endif

// Cross-GPU hand-shake in front of a peer read (xgmi_allreduce.hip): publish "my buffer number s of phase p is complete" to
// every peer's flag array, then wait until every peer has published s.  world == 0: nothing to wait for (single GPU, or the
// wait ran as its own one-wave launch in front of this one: `fence` then asks for the acquire that goes with it).
// Flag layout of a rank's buffer: flags[phase][kMaxRanks] u32; phase 0 = "staged gradient complete", phase 1 = "reduced shard
// complete" (the reduce-scatter + all-gather form).
struct PeerWait {
    unsigned* flags_mine;             // [phases][kMaxRanks], fine-grained LOCAL memory, slot r written remotely by rank r
    unsigned* flags_peer[kMaxRanks];  // peer r's flag array (mapped over IPC); this rank writes slot [rank]
    const unsigned* seq;              // local: number of all-reduces completed so far (this one is seq + 1)
    int* status;                      // local: set to 1 when a wait timed out (the AdamW launch then skips; the host raises)
    long long timeout_ticks;          // wall_clock64 ticks (100 MHz)
    int rank, world;
    int phase;                        // which flag set
    int fence;                        // world == 0 only: system-scope acquire at kernel start (the wait was a launch of its own)
};

// Reduce-scatter + all-gather form: where the AdamW launch finds the reduced gradient -- shard r (float4 elements
// [r * shard4, (r + 1) * shard4)) lives in rank r's buffer, next to the squared-norm partials of that shard.
struct GatherSrc {
    const float* shard[kMaxRanks];    // rank r's reduced-shard area, slot 0
    const double* parts[kMaxRanks];   // rank r's norm partials, slot 0 (nparts doubles per slot)
    long long shard4;                 // float4 elements per shard
    long long slot_stride;            // floats between slot 0 and slot 1 of the shard area
    int nparts;                       // partials per rank
    int world;                        // 0: not used (the gradient is local)
    const unsigned* seq;
};

// Per-step scalars of the update, formed ONCE per step in double like torch's python scalars -- bias corrections from
// beta ** step with the betas as DOUBLES (a float 0.999 is 0.99900001...: 1 - beta2 ** t would be off by 1e-5 relative at
// early steps) -- then narrowed to f32 exactly where torch narrows them (a python scalar meeting a float tensor).
struct AdamScalars {
    float bc2_sqrt, one_m_b1, one_m_b2, beta2, eps;
    float step_size[RLX_ADAMW_MAX_GROUPS], decay[RLX_ADAMW_MAX_GROUPS];
};

inline size_t scalars_offset() { return (size_t)kMaxParts * sizeof(double); }

int grid_for(long long n);

// grad_reduce_sqnorm + clip_adamw on `stream`.  out: reduced (scaled) gradient; wait: peer hand-shake or nullptr;
// seq_inc: device word incremented by the AdamW launch (the all-reduce sequence number), or nullptr; status: device word that
// makes the AdamW launch skip its update when set (a peer wait timed out: the sums are garbage), or nullptr.
int launch_reduce_clip_adamw(float* params, const ReduceSrc& src, float* out, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const rlx_adamw_params* p, float* stats, int32_t* step_state, void* workspace, size_t workspace_bytes,
                             const PeerWait* wait, unsigned* seq_inc, const int* status, hipStream_t stream);
// the reduce alone (validation / plain all-reduce): out = scale * sum; no optimizer state touched
int launch_reduce_only(const ReduceSrc& src, float* out, int64_t n, float scale, void* workspace, size_t workspace_bytes,
                       const PeerWait* wait, unsigned* seq_inc, hipStream_t stream);
// publish + wait as a launch of its own: ONE wave, long sleeps between polls (ranks that share a device, or any set-up where a
// device-wide spin must not hold the GPU)
int launch_peer_wait(const PeerWait& wait, hipStream_t stream);
// reduce-scatter: this rank's shard of the sum of every rank's staged gradient -> its shard area + norm partials
// (shard0 / parts0: slot 0 of the areas; the kernel picks the slot from the device-side sequence number)
int launch_reduce_scatter(const ReduceSrc& src, float* shard0, long long shard_stride, double* parts0, long long shard_lo4,
                          long long shard_n4, int nparts, float scale, int32_t* step_state, unsigned* seq_snapshot,
                          const PeerWait* wait, hipStream_t stream);  // seq_snapshot: device word that receives *src.seq (see the kernel), or nullptr
// clip + AdamW with the reduced gradient gathered from every rank's shard area (all-gather fused into the update); `out`
// receives the clipped gradient like the local form.  The launch increments *seq_inc, so src.seq / wait->seq must NOT point at that
// word: pass the snapshot the reduce-scatter launch left.
int launch_gather_clip_adamw(float* params, const GatherSrc& src, float* out, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const rlx_adamw_params* p, float* stats, int32_t* step_state, const PeerWait* wait, unsigned* seq_inc,
                             const int* status, hipStream_t stream);
// plain gather (validation / plain all-reduce): out = the reduced gradient, no optimizer state touched
int launch_gather_only(const GatherSrc& src, float* out, int64_t n, const PeerWait* wait, unsigned* seq_inc, hipStream_t stream);
int rsag_parts(long long shard_n4);

// The exchange areas of every rank's communicator buffer, as rlx_xgmi_clip_adamw_step hands them to the one-launch form
// (adamw_clip.hip: XchgPeers).  All areas hold self-validating 64-bit words (epoch << 32 | payload).
struct ExchangeBuffers {
    unsigned long long* inbox[kMaxRanks];   // rank q's inbox: [world rows][n_cap words]
    unsigned long long* gather[kMaxRanks];  // rank q's gather area: [n_cap words]
    unsigned long long* parts[kMaxRanks];   // rank q's norm-partial slots: [512][2 words]
    long long n_cap;
    int rank, world;
    int self_alias;  // 0 real peers; 1 self-connected, exact emulation; 2 self-connected, timing emulation (rlx_xgmi_self_timing)
};
constexpr int kExchangeSlots = 512;  // (= kMaxOneBlocks of adamw_clip.hip)
int launch_exchange_clip_adamw_one_launch(float* params, const float* grads, int nslab, float* out, float* exp_avg, float* exp_avg_sq,
                                          int64_t n, const rlx_adamw_params* p, float* stats, int32_t* step_state, const ExchangeBuffers& xb,
                                          unsigned long long* xsync, int co_resident, long long timeout_ticks, int* status, hipStream_t stream, bool* used);  // blocks (= norm partials) of the reduce-scatter launch for a shard of this many float4

}  // namespace opt
}  // namespace rlx
