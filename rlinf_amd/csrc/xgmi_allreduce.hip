// xgmi_allreduce.hip -- the data-parallel gradient all-reduce as peer reads over xGMI, gfx950.
//
// Replaces the gradient synchronisation of the reference's NO_SHARD data-parallel learner (rlinf/hybrid_engines/fsdp/
// strategy/fsdp.py:480-496: FSDP's all-reduce behind backward; the averaged gradient is what optimizer_step clips and
// applies, fsdp_model_manager.py:429-463).  Not a translation of an NCCL call pattern: xGMI is a full mesh of
// point-to-point links, the payload (1.15 MB) is latency-bound, so every rank simply READS every peer's staged gradient
// (IPC-mapped fine-grained memory) and adds them in rank order -- one hop, W - 1 links busy in parallel, results
// bit-identical on all ranks -- fused with the squared-norm pass the clip needs anyway.  Two launches (stage, reduce), no
// host round trip: the optimizer step stays a pure kernel chain that a hipGraph can hold.
//
// Memory: hipExtMallocWithFlags(hipDeviceMallocFinegrained): stores write through and remote reads are not served from a
// stale L2 line; the flags are system-scope atomics.  Layout of a rank's buffer: [flags: kMaxRanks x u32, padded to 4 KiB]
// [slot 0: n_max f32][slot 1: n_max f32].  Two slots suffice: a rank can only reach the staging launch of all-reduce s + 2
// after every peer has published s + 1, i.e. after every peer finished reading slot s.
//
// Two forms of the exchange (rlx_xgmi_configure, `algo`):
//   direct  every rank reads all W staged gradients in full (n floats per link, ONE hand-shake)        -- W = 2, 3
//   rs + ag reduce-scatter: rank r sums shard r of every rank's staged gradient into its shard area;
//           all-gather fused into the AdamW launch, which reads shard q from rank q
//           (2 n / W floats per link, TWO hand-shakes; the norm partials travel with the shards)        -- W >= 4
// and two forms of the hand-shake (`wait_mode`): inline in the consuming launch (every block polls: lowest latency, one GPU per
// rank) or as a one-wave launch of its own in front of it (ranks sharing a device: a device-wide spin would starve the peer).

#include <string.h>

#include <algorithm>

#include "opt_common.h"

struct rlx_xgmi_comm {
    int rank, world;
    int64_t n_max;
    long long timeout_ticks;
    int algo;                              // 0 direct, 1 reduce-scatter + all-gather
    int wait_mode;                         // 0 inline, 1 own launch
    char* base_local;                      // own buffer
    char* base_peer[RLX_XGMI_MAX_RANKS];   // mapped peers (own slot = base_local)
    unsigned* seq;                         // device word (plain memory): all-reduces completed
    unsigned* seq_snapshot;                // device word: copy of *seq taken by the reduce-scatter launch for the gather + AdamW launch
    int* status;                           // device word: timeout flag
    unsigned long long* xsync;             // device words of the one-launch exchange: [0] its epoch, [1] "a poll expired" (sticky).  The
                                           // communicator's own, NOT the caller's sync words: the tags in the exchange areas count THIS
                                           // communicator's exchanges, whichever parameter set (learner, start-up validation) ran them
    bool connected, local_peers;           // local_peers: same-process emulation, nothing to unmap
    bool self_alias;                       // rlx_xgmi_connect_self: every peer is this rank's own buffer (timing tool, results invalid)
    bool self_timing;                      // rlx_xgmi_self_timing: the one-launch exchange's timing emulation instead of the exact one
};

namespace rlx {
namespace {

using namespace opt;

constexpr size_t kFlagBytes = 4096;                      // flags[2 phases][kMaxRanks] u32, padded
constexpr int kPartsPerRank = kMaxParts / kMaxRanks;     // norm partials a rank publishes with its shard

// [flags][slot 0][slot 1][shard 0][shard 1][parts 0][parts 1] | the one-launch exchange's areas, 64-bit tagged words:
// [xparts: kExchangeSlots x 2][xgather: n_max][xinbox: world x n_max]
inline size_t chain_bytes(int64_t n_max) {
    return (kFlagBytes + 4 * (size_t)n_max * sizeof(float) + 2 * kPartsPerRank * sizeof(double) + 255) / 256 * 256;
}
// (rows are laid out in groups of 64 float4s = 256 words: a row holds n_max words rounded up to whole groups)
inline size_t xrow_words(int64_t n_max) { return ((size_t)n_max + 255) / 256 * 256; }
inline size_t buffer_bytes(int64_t n_max, int world) {
    return chain_bytes(n_max) + 8 * ((size_t)kExchangeSlots * 2 + xrow_words(n_max) * (1 + (size_t)world));
}
inline unsigned long long* xparts_ptr(char* base, int64_t n_max) { return reinterpret_cast<unsigned long long*>(base + chain_bytes(n_max)); }
inline unsigned long long* xgather_ptr(char* base, int64_t n_max) { return xparts_ptr(base, n_max) + (size_t)kExchangeSlots * 2; }
inline unsigned long long* xinbox_ptr(char* base, int64_t n_max) { return xgather_ptr(base, n_max) + xrow_words(n_max); }
inline float* slot_ptr(char* base, int64_t n_max, int slot) {
    return reinterpret_cast<float*>(base + kFlagBytes) + (size_t)slot * (size_t)n_max;
}
inline float* shard_ptr(char* base, int64_t n_max) { return reinterpret_cast<float*>(base + kFlagBytes) + 2 * (size_t)n_max; }
inline double* parts_ptr(char* base, int64_t n_max) {
    return reinterpret_cast<double*>(base + kFlagBytes + 4 * (size_t)n_max * sizeof(float));
}

// stage: dst = sum of this rank's slabs, into slot (seq + 1) & 1 of the local buffer
template <bool DEFER>
__global__ __launch_bounds__(256) void xgmi_stage_kernel(const float* __restrict__ g, long long n, int nslab, float* __restrict__ slot0,
                                                         long long n_max, const unsigned* __restrict__ seq, DeferredScale dfr) {
    float* dst = slot0 + (size_t)((*seq + 1u) & 1u) * (size_t)n_max;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (n % 4 == 0) && (n_max % 4 == 0) && (reinterpret_cast<uintptr_t>(g) % 16 == 0);
    const long long n4 = vec ? n / 4 : 0;
    for (long long i = tid0; i < n4; i += stride) {
        float4 s;
        if constexpr (DEFER) s = sum_slab_groups_f4(reinterpret_cast<const float4*>(g), i, n4, nslab, dfr);
        else s = sum_slabs_f4(reinterpret_cast<const float4*>(g) + i, n4, nslab);  // all slabs in flight, ascending adds
        reinterpret_cast<float4*>(dst)[i] = s;
    }
    for (long long i = n4 * 4 + tid0; i < n; i += stride) {
        float s;
        if constexpr (DEFER) {
            s = sum_slab_groups(g, i, n, nslab, dfr);
        } else {
            s = g[i];
            for (int k = 1; k < nslab; ++k) s += g[(long long)k * n + i];
        }
        dst[i] = s;
    }
}

int check(const rlx_xgmi_comm* c, int64_t n, const char* who) {
    RLX_REQUIRE(c != nullptr, "%s: NULL communicator", who);
    RLX_REQUIRE(c->connected, "%s: rlx_xgmi_connect has not been called", who);
    RLX_REQUIRE(n >= 0 && n <= c->n_max, "%s: n=%lld exceeds the communicator's n_max=%lld", who, (long long)n, (long long)c->n_max);
    return RLX_OK;
}

void fill_wait(const rlx_xgmi_comm* c, PeerWait& w, int phase) {
    w.flags_mine = reinterpret_cast<unsigned*>(c->base_local);
    for (int r = 0; r < c->world; ++r) w.flags_peer[r] = reinterpret_cast<unsigned*>(c->base_peer[r]);
    // self-aliased: "peer r's slot [rank]" must land in OWN slot r, or the wait for peer r would never be satisfied
    if (c->self_alias)
        for (int r = 0; r < c->world; ++r) w.flags_peer[r] = w.flags_mine + r - c->rank;
    w.seq = c->seq;
    w.status = c->status;
    w.timeout_ticks = c->timeout_ticks;
    w.rank = c->rank;
    w.world = c->world;
    w.phase = phase;
    w.fence = 0;
}

// The hand-shake of `phase` in front of the next launch: inline (returns the PeerWait that launch executes itself) or as its own
// one-wave launch (the next launch then only fences).
int handshake(const rlx_xgmi_comm* c, int phase, PeerWait& for_next, hipStream_t st) {
    fill_wait(c, for_next, phase);
    if (c->world <= 1) { for_next.world = 0; return RLX_OK; }
    if (c->wait_mode == 1) {
        if (int rc = launch_peer_wait(for_next, st)) return rc;
        for_next.world = 0;
        for_next.fence = 1;
    }
    return RLX_OK;
}

void fill_reduce_src(const rlx_xgmi_comm* c, ReduceSrc& src) {
    src.nbase = c->world;
    src.nslab = 1;
    for (int r = 0; r < c->world; ++r) src.base[r] = slot_ptr(c->base_peer[r], c->n_max, 0);
    src.slot_stride = c->n_max;
    src.seq = c->seq;
}

struct ShardPlan { long long shard4, lo4, cnt4; int nparts; };

// rs + ag needs whole float4 and at least one float4 per rank; the plan depends on (n, world) only, so every rank agrees
bool use_rsag(const rlx_xgmi_comm* c, int64_t n) { return c->algo == 1 && c->world >= 2 && n % 4 == 0 && n / 4 >= c->world; }

ShardPlan plan_shards(const rlx_xgmi_comm* c, int64_t n) {
    ShardPlan p{};
    const long long n4 = n / 4;
    p.shard4 = (n4 + c->world - 1) / c->world;
    p.lo4 = (long long)c->rank * p.shard4;
    p.cnt4 = std::max<long long>(0, std::min<long long>(p.shard4, n4 - p.lo4));
    p.nparts = rsag_parts(p.shard4);
    return p;
}

void fill_gather_src(const rlx_xgmi_comm* c, const ShardPlan& p, GatherSrc& g) {
    for (int r = 0; r < c->world; ++r) {
        g.shard[r] = shard_ptr(c->base_peer[r], c->n_max);
        g.parts[r] = parts_ptr(c->base_peer[r], c->n_max);
    }
    g.shard4 = p.shard4;
    g.slot_stride = c->n_max;
    g.nparts = p.nparts;
    g.world = c->world;
    g.seq = c->seq;
}

// (`deferred`: the per-micro-batch scales of a decoupled rlx_ppo_step's actor gradients, applied as the slabs are summed -- per rank,
//  before the exchange, like every rank's own loss denominator)
int stage(const rlx_xgmi_comm* c, const float* in, int slabs, int64_t n, hipStream_t st, const rlx_adamw_params* deferred = nullptr) {
    if (deferred != nullptr && deferred->deferred_scale != nullptr) {
        if (int rc = check_deferred(deferred, slabs, n, "rlx_xgmi_clip_adamw_step")) return rc;
        hipLaunchKernelGGL(xgmi_stage_kernel<true>, dim3(grid_for(n)), dim3(256), 0, st, in, (long long)n, slabs,
                           slot_ptr(c->base_local, c->n_max, 0), (long long)c->n_max, c->seq, deferred_of(deferred));
    } else {
        hipLaunchKernelGGL(xgmi_stage_kernel<false>, dim3(grid_for(n)), dim3(256), 0, st, in, (long long)n, slabs,
                           slot_ptr(c->base_local, c->n_max, 0), (long long)c->n_max, c->seq, DeferredScale{});
    }
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

void release(rlx_xgmi_comm* c) {
    if (c == nullptr) return;
    if (!c->local_peers)
        for (int r = 0; r < c->world; ++r)
            if (r != c->rank && c->base_peer[r] != nullptr) (void)hipIpcCloseMemHandle(c->base_peer[r]);
    if (c->base_local) (void)hipFree(c->base_local);
    if (c->seq) (void)hipFree(c->seq);
    delete c;
}

}  // namespace
}  // namespace rlx

using namespace rlx;
using namespace rlx::opt;

extern "C" int rlx_xgmi_create(int rank, int world, int64_t n_max, int timeout_ms, int mem_kind, rlx_xgmi_comm** comm,
                               void* handle_out) {
    RLX_REQUIRE(comm != nullptr && handle_out != nullptr, "rlx_xgmi_create: NULL output");
    RLX_REQUIRE(world >= 1 && world <= RLX_XGMI_MAX_RANKS && rank >= 0 && rank < world, "rlx_xgmi_create: rank %d of %d (max %d ranks)",
                rank, world, RLX_XGMI_MAX_RANKS);
    RLX_REQUIRE(n_max >= 1, "rlx_xgmi_create: n_max=%lld", (long long)n_max);
    static_assert(sizeof(hipIpcMemHandle_t) == RLX_XGMI_HANDLE_BYTES, "handle size");
    n_max = (n_max + 3) / 4 * 4;
    rlx_xgmi_comm* c = new rlx_xgmi_comm();  // value-initialised: every pointer NULL, so release() is safe at any point below
    c->rank = rank; c->world = world; c->n_max = n_max;
    c->timeout_ticks = (long long)(timeout_ms > 0 ? timeout_ms : 300000) * 100000ll;  // wall_clock64: 100 MHz
    c->algo = world >= 4 ? 1 : 0;
    const size_t bytes = buffer_bytes(n_max, world);
    void *p = nullptr, *words = nullptr;
    hipError_t e = mem_kind == 2 ? hipMalloc(&p, bytes)
                                 : hipExtMallocWithFlags(&p, bytes, mem_kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    const char* what = "allocating the exchange buffer";
    if (e == hipSuccess) {
        c->base_local = static_cast<char*>(p);
        c->base_peer[rank] = c->base_local;
        what = "hipMemset";
        e = hipMemset(p, 0, bytes);
    }
    if (e == hipSuccess) { what = "hipMalloc (sequence / status words)"; e = hipMalloc(&words, 256); }
    if (e == hipSuccess) {
        c->seq = static_cast<unsigned*>(words);
        c->status = reinterpret_cast<int*>(static_cast<char*>(words) + 128);
        c->seq_snapshot = reinterpret_cast<unsigned*>(static_cast<char*>(words) + 64);
        c->xsync = reinterpret_cast<unsigned long long*>(static_cast<char*>(words) + 192);
        what = "hipMemset";
        e = hipMemset(words, 0, 256);
    }
    if (e == hipSuccess) { what = "hipDeviceSynchronize"; e = hipDeviceSynchronize(); }
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof(h));
    if (e == hipSuccess && world > 1) { what = "hipIpcGetMemHandle"; e = hipIpcGetMemHandle(&h, p); }
    if (e != hipSuccess) {  // one cleanup path for every failure above
        set_error("rlx_xgmi_create: %s failed (%zu bytes, mem_kind %d): %s", what, bytes, mem_kind, hipGetErrorString(e));
        release(c);
        return RLX_EHIP;
    }
    memcpy(handle_out, &h, sizeof(h));
    c->connected = world == 1;
    *comm = c;
    return RLX_OK;
}

extern "C" int rlx_xgmi_connect(rlx_xgmi_comm* c, const void* all_handles) {
    RLX_REQUIRE(c != nullptr && all_handles != nullptr, "rlx_xgmi_connect: NULL argument");
    RLX_REQUIRE(!c->connected || c->world == 1, "rlx_xgmi_connect: already connected");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(all_handles) + (size_t)r * RLX_XGMI_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_error("rlx_xgmi_connect: hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(e));
            for (int q = 0; q < r; ++q)  // unmap what was mapped so far: a failed connect leaves nothing behind
                if (q != c->rank && c->base_peer[q] != nullptr) {
                    (void)hipIpcCloseMemHandle(c->base_peer[q]);
                    c->base_peer[q] = nullptr;
                }
            return RLX_EHIP;
        }
        c->base_peer[r] = static_cast<char*>(p);
    }
    c->connected = true;
    return RLX_OK;
}

extern "C" int rlx_xgmi_connect_self(rlx_xgmi_comm* c) {
    RLX_REQUIRE(c != nullptr, "rlx_xgmi_connect_self: NULL communicator");
    RLX_REQUIRE(!c->connected || c->world == 1, "rlx_xgmi_connect_self: already connected");
    for (int q = 0; q < c->world; ++q) c->base_peer[q] = c->base_local;
    c->connected = true;
    c->local_peers = true;
    c->self_alias = true;
    return RLX_OK;
}

extern "C" int rlx_xgmi_self_timing(rlx_xgmi_comm* c, int on) {
    RLX_REQUIRE(c != nullptr && c->self_alias, "rlx_xgmi_self_timing: not a self-connected communicator");
    c->self_timing = on != 0;
    return RLX_OK;
}

extern "C" int rlx_xgmi_connect_local(rlx_xgmi_comm* const* comms, int world) {
    RLX_REQUIRE(comms != nullptr && world >= 1 && world <= RLX_XGMI_MAX_RANKS, "rlx_xgmi_connect_local: bad arguments");
    for (int r = 0; r < world; ++r)
        RLX_REQUIRE(comms[r] != nullptr && comms[r]->world == world && comms[r]->rank == r && comms[r]->n_max == comms[0]->n_max,
                    "rlx_xgmi_connect_local: communicator %d does not belong to this group", r);
    for (int r = 0; r < world; ++r) {
        for (int q = 0; q < world; ++q) comms[r]->base_peer[q] = comms[q]->base_local;
        comms[r]->connected = true;
        comms[r]->local_peers = true;
    }
    return RLX_OK;
}

extern "C" int rlx_xgmi_configure(rlx_xgmi_comm* c, int algo, int wait_mode, int timeout_ms) {
    RLX_REQUIRE(c != nullptr, "rlx_xgmi_configure: NULL communicator");
    RLX_REQUIRE(algo >= -1 && algo <= 1 && wait_mode >= -1 && wait_mode <= 1, "rlx_xgmi_configure: algo %d / wait_mode %d", algo, wait_mode);
    if (algo >= 0) c->algo = algo;
    if (wait_mode >= 0) c->wait_mode = wait_mode;
    if (timeout_ms > 0) c->timeout_ticks = (long long)timeout_ms * 100000ll;
    return RLX_OK;
}

extern "C" int rlx_xgmi_destroy(rlx_xgmi_comm* c) {
    if (c == nullptr) return RLX_OK;
    (void)hipDeviceSynchronize();
    release(c);
    return RLX_OK;
}

extern "C" int rlx_xgmi_status(rlx_xgmi_comm* c) {
    RLX_REQUIRE(c != nullptr, "rlx_xgmi_status: NULL communicator");
    int st = 0;
    RLX_HIP_CHECK(hipMemcpy(&st, c->status, sizeof(int), hipMemcpyDeviceToHost));
    if (st != 0) RLX_HIP_CHECK(hipMemset(c->status, 0, sizeof(int)));
    return st != 0 ? 1 : 0;
}

__global__ void xgmi_status_snapshot_kernel(const int* __restrict__ status, float* __restrict__ dst) { *dst = (float)*status; }

extern "C" int rlx_xgmi_status_snapshot(rlx_xgmi_comm* c, float* dst, rlx_stream_t stream) {
    RLX_REQUIRE(c != nullptr && dst != nullptr, "rlx_xgmi_status_snapshot: NULL argument");
    hipLaunchKernelGGL(xgmi_status_snapshot_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), c->status, dst);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_xgmi_allreduce_f32(rlx_xgmi_comm* c, const float* in, int slabs, float* out, int64_t n, float scale,
                                      void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    if (int rc = check(c, n, "rlx_xgmi_allreduce_f32")) return rc;
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(in && out && slabs >= 1, "rlx_xgmi_allreduce_f32: NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = stage(c, in, slabs, n, st)) return rc;
    ReduceSrc src{};
    fill_reduce_src(c, src);
    PeerWait w{};
    if (int rc = handshake(c, 0, w, st)) return rc;
    if (!use_rsag(c, n)) return launch_reduce_only(src, out, n, scale, workspace, workspace_bytes, &w, c->seq, st);
    const ShardPlan p = plan_shards(c, n);
    if (int rc = launch_reduce_scatter(src, shard_ptr(c->base_local, c->n_max), c->n_max, parts_ptr(c->base_local, c->n_max), p.lo4,
                                       p.cnt4, p.nparts, scale, nullptr, nullptr, &w, st))
        return rc;
    GatherSrc g{};
    fill_gather_src(c, p, g);
    PeerWait w1{};
    if (int rc = handshake(c, 1, w1, st)) return rc;
    return launch_gather_only(g, out, n, &w1, c->seq, st);
}

extern "C" int rlx_xgmi_clip_adamw_step(rlx_xgmi_comm* c, float* params, const float* grads, float* grad_flat, float* exp_avg,
                                        float* exp_avg_sq, int64_t n, const rlx_adamw_params* p, float* stats, int32_t* step_state,
                                        void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    if (int rc = check(c, n, "rlx_xgmi_clip_adamw_step")) return rc;
    RLX_REQUIRE(p != nullptr && p->grad_partials >= 1 && grads && grad_flat, "rlx_xgmi_clip_adamw_step: NULL argument");
    if (n == 0) return RLX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // One launch per rank and step (rlx_adamw_params.sync_words; adamw_clip.hip, XchgPeers): the gradient travels as pushed,
    // self-validating words inside the optimizer launch.  Needs every rank's kernel resident at once, so not with the hand-shake as
    // its own launch (ranks that share a device); in-process groups of one device count all their ranks against the residency.
    if (p->sync_words != nullptr && c->world >= 2 && c->wait_mode == 0) {
        ExchangeBuffers xb{};
        for (int r = 0; r < c->world; ++r) {
            xb.inbox[r] = xinbox_ptr(c->base_peer[r], c->n_max);
            xb.gather[r] = xgather_ptr(c->base_peer[r], c->n_max);
            xb.parts[r] = xparts_ptr(c->base_peer[r], c->n_max);
        }
        xb.n_cap = (long long)xrow_words(c->n_max), xb.rank = c->rank, xb.world = c->world, xb.self_alias = c->self_alias ? (c->self_timing ? 2 : 1) : 0;
        bool used = false;
        const int co_resident = (c->local_peers && !c->self_alias) ? c->world : 1;
        if (int rc = launch_exchange_clip_adamw_one_launch(params, grads, p->grad_partials, grad_flat, exp_avg, exp_avg_sq, n, p, stats,
                                                           step_state, xb, c->xsync, co_resident, c->timeout_ticks, c->status, st, &used))
            return rc;
        if (used) return RLX_OK;
    }
    if (int rc = stage(c, grads, p->grad_partials, n, st, p)) return rc;  // (applies p->deferred_scale, if any, to this rank's slabs)
    ReduceSrc src{};
    fill_reduce_src(c, src);
    PeerWait w{};
    if (int rc = handshake(c, 0, w, st)) return rc;
    if (!use_rsag(c, n))
        return launch_reduce_clip_adamw(params, src, grad_flat, exp_avg, exp_avg_sq, n, p, stats, step_state, workspace, workspace_bytes,
                                        &w, c->seq, c->status, st);
    const ShardPlan sp = plan_shards(c, n);
    if (int rc = launch_reduce_scatter(src, shard_ptr(c->base_local, c->n_max), c->n_max, parts_ptr(c->base_local, c->n_max), sp.lo4,
                                       sp.cnt4, sp.nparts, p->grad_scale, step_state, c->seq_snapshot, &w, st))
        return rc;
    GatherSrc g{};
    fill_gather_src(c, sp, g);
    PeerWait w1{};
    if (int rc = handshake(c, 1, w1, st)) return rc;
    // the gather + AdamW launch increments c->seq itself: its blocks read the reduce-scatter launch's snapshot instead
    g.seq = c->seq_snapshot;
    w1.seq = c->seq_snapshot;
    return launch_gather_clip_adamw(params, g, grad_flat, exp_avg, exp_avg_sq, n, p, stats, step_state, &w1, c->seq, c->status, st);
}
