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
__device__ __forceinline__ float4 sum_slabs_f4(const float4* __restrict__ gp, long long n4, int nslab) {
    constexpr int SB = 24;
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

// Cross-GPU hand-shake in front of a peer read (xgmi_allreduce.hip): publish "my staged gradient number s is complete" to
// every peer's flag array, then wait until every peer has published s.  world == 0: single GPU, nothing to wait for.
struct PeerWait {
    unsigned* flags_mine;             // [world], fine-grained LOCAL memory, slot r written remotely by rank r
    unsigned* flags_peer[kMaxRanks];  // peer r's flag array (mapped over IPC); this rank writes slot [rank]
    const unsigned* seq;              // local: number of all-reduces completed so far (this one is seq + 1)
    int* status;                      // local: set to 1 when a wait timed out (results are then garbage; the host raises)
    long long timeout_ticks;          // wall_clock64 ticks (100 MHz)
    int rank, world;
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
// seq_inc: device word incremented by the AdamW launch (the all-reduce sequence number), or nullptr.
int launch_reduce_clip_adamw(float* params, const ReduceSrc& src, float* out, float* exp_avg, float* exp_avg_sq, int64_t n,
                             const rlx_adamw_params* p, float* stats, int32_t* step_state, void* workspace, size_t workspace_bytes,
                             const PeerWait* wait, unsigned* seq_inc, hipStream_t stream);
// the reduce alone (validation / plain all-reduce): out = scale * sum; no optimizer state touched
int launch_reduce_only(const ReduceSrc& src, float* out, int64_t n, float scale, void* workspace, size_t workspace_bytes,
                       const PeerWait* wait, unsigned* seq_inc, hipStream_t stream);

}  // namespace opt
}  // namespace rlx
