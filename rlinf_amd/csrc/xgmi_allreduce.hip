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

#include <string.h>

#include "opt_common.h"

struct rlx_xgmi_comm {
    int rank, world;
    int64_t n_max;
    long long timeout_ticks;
    char* base_local;                      // own buffer
    char* base_peer[RLX_XGMI_MAX_RANKS];   // mapped peers (own slot = base_local)
    unsigned* seq;                         // device word (plain memory): all-reduces completed
    int* status;                           // device word: timeout flag
    bool connected;
};

namespace rlx {
namespace {

using namespace opt;

constexpr size_t kFlagBytes = 4096;

inline float* slot_ptr(char* base, int64_t n_max, int slot) {
    return reinterpret_cast<float*>(base + kFlagBytes) + (size_t)slot * (size_t)n_max;
}

// stage: dst = sum of this rank's slabs, into slot (seq + 1) & 1 of the local buffer
__global__ __launch_bounds__(256) void xgmi_stage_kernel(const float* __restrict__ g, long long n, int nslab, float* __restrict__ slot0,
                                                         long long n_max, const unsigned* __restrict__ seq) {
    float* dst = slot0 + (size_t)((*seq + 1u) & 1u) * (size_t)n_max;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long tid0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool vec = (n % 4 == 0) && (n_max % 4 == 0) && (reinterpret_cast<uintptr_t>(g) % 16 == 0);
    const long long n4 = vec ? n / 4 : 0;
    for (long long i = tid0; i < n4; i += stride) {
        const float4 s = sum_slabs_f4(reinterpret_cast<const float4*>(g) + i, n4, nslab);  // all slabs in flight, ascending adds
        reinterpret_cast<float4*>(dst)[i] = s;
    }
    for (long long i = n4 * 4 + tid0; i < n; i += stride) {
        float s = g[i];
        for (int k = 1; k < nslab; ++k) s += g[(long long)k * n + i];
        dst[i] = s;
    }
}

int check(const rlx_xgmi_comm* c, int64_t n, const char* who) {
    RLX_REQUIRE(c != nullptr, "%s: NULL communicator", who);
    RLX_REQUIRE(c->connected, "%s: rlx_xgmi_connect has not been called", who);
    RLX_REQUIRE(n >= 0 && n <= c->n_max, "%s: n=%lld exceeds the communicator's n_max=%lld", who, (long long)n, (long long)c->n_max);
    return RLX_OK;
}

void fill_wait(const rlx_xgmi_comm* c, PeerWait& w) {
    w.flags_mine = reinterpret_cast<unsigned*>(c->base_local);
    for (int r = 0; r < c->world; ++r) w.flags_peer[r] = reinterpret_cast<unsigned*>(c->base_peer[r]);
    w.seq = c->seq;
    w.status = c->status;
    w.timeout_ticks = c->timeout_ticks;
    w.rank = c->rank;
    w.world = c->world;
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
    rlx_xgmi_comm* c = new rlx_xgmi_comm();
    c->rank = rank; c->world = world; c->n_max = n_max;
    c->timeout_ticks = (long long)(timeout_ms > 0 ? timeout_ms : 120000) * 100000ll;  // wall_clock64: 100 MHz
    const size_t bytes = kFlagBytes + 2 * (size_t)n_max * sizeof(float);
    void* p = nullptr;
    hipError_t e = mem_kind == 2 ? hipMalloc(&p, bytes)
                                 : hipExtMallocWithFlags(&p, bytes, mem_kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        set_error("rlx_xgmi_create: allocating %zu bytes (mem_kind %d) failed: %s", bytes, mem_kind, hipGetErrorString(e));
        delete c;
        return RLX_EHIP;
    }
    c->base_local = static_cast<char*>(p);
    c->base_peer[rank] = c->base_local;
    RLX_HIP_CHECK(hipMemset(p, 0, bytes));
    void* words = nullptr;
    RLX_HIP_CHECK(hipMalloc(&words, 256));
    RLX_HIP_CHECK(hipMemset(words, 0, 256));
    c->seq = static_cast<unsigned*>(words);
    c->status = reinterpret_cast<int*>(static_cast<char*>(words) + 128);
    RLX_HIP_CHECK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        set_error("rlx_xgmi_create: hipIpcGetMemHandle failed (mem_kind %d): %s", mem_kind, hipGetErrorString(e));
        (void)hipFree(p);
        (void)hipFree(words);
        delete c;
        return RLX_EHIP;
    }
    memcpy(handle_out, &h, sizeof(h));
    c->connected = world == 1;
    *comm = c;
    return RLX_OK;
}

extern "C" int rlx_xgmi_connect(rlx_xgmi_comm* c, const void* all_handles) {
    RLX_REQUIRE(c != nullptr && all_handles != nullptr, "rlx_xgmi_connect: NULL argument");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(all_handles) + (size_t)r * RLX_XGMI_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            set_error("rlx_xgmi_connect: hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(e));
            return RLX_EHIP;
        }
        c->base_peer[r] = static_cast<char*>(p);
    }
    c->connected = true;
    return RLX_OK;
}

extern "C" int rlx_xgmi_destroy(rlx_xgmi_comm* c) {
    if (c == nullptr) return RLX_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->base_peer[r] != nullptr) (void)hipIpcCloseMemHandle(c->base_peer[r]);
    if (c->base_local) (void)hipFree(c->base_local);
    if (c->seq) (void)hipFree(c->seq);
    delete c;
    return RLX_OK;
}

extern "C" int rlx_xgmi_status(rlx_xgmi_comm* c) {
    RLX_REQUIRE(c != nullptr, "rlx_xgmi_status: NULL communicator");
    int st = 0;
    RLX_HIP_CHECK(hipMemcpy(&st, c->status, sizeof(int), hipMemcpyDeviceToHost));
    if (st != 0) RLX_HIP_CHECK(hipMemset(c->status, 0, sizeof(int)));
    return st != 0 ? 1 : 0;
}

extern "C" int rlx_xgmi_allreduce_f32(rlx_xgmi_comm* c, const float* in, int slabs, float* out, int64_t n, float scale,
                                      void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    if (int rc = check(c, n, "rlx_xgmi_allreduce_f32")) return rc;
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(in && out && slabs >= 1, "rlx_xgmi_allreduce_f32: NULL argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(xgmi_stage_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, (long long)n, slabs,
                       slot_ptr(c->base_local, c->n_max, 0), (long long)c->n_max, c->seq);
    RLX_LAUNCH_CHECK();
    ReduceSrc src{};
    src.nbase = c->world;
    src.nslab = 1;
    for (int r = 0; r < c->world; ++r) src.base[r] = slot_ptr(c->base_peer[r], c->n_max, 0);
    src.slot_stride = c->n_max;
    src.seq = c->seq;
    PeerWait w{};
    fill_wait(c, w);
    return launch_reduce_only(src, out, n, scale, workspace, workspace_bytes, &w, c->seq, st);
}

extern "C" int rlx_xgmi_clip_adamw_step(rlx_xgmi_comm* c, float* params, const float* grads, float* grad_flat, float* exp_avg,
                                        float* exp_avg_sq, int64_t n, const rlx_adamw_params* p, float* stats, int32_t* step_state,
                                        void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    if (int rc = check(c, n, "rlx_xgmi_clip_adamw_step")) return rc;
    RLX_REQUIRE(p != nullptr && p->grad_partials >= 1 && grads && grad_flat, "rlx_xgmi_clip_adamw_step: NULL argument");
    if (n == 0) return RLX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(xgmi_stage_kernel, dim3(grid_for(n)), dim3(256), 0, st, grads, (long long)n, p->grad_partials,
                       slot_ptr(c->base_local, c->n_max, 0), (long long)c->n_max, c->seq);
    RLX_LAUNCH_CHECK();
    ReduceSrc src{};
    src.nbase = c->world;
    src.nslab = 1;
    for (int r = 0; r < c->world; ++r) src.base[r] = slot_ptr(c->base_peer[r], c->n_max, 0);
    src.slot_stride = c->n_max;
    src.seq = c->seq;
    PeerWait w{};
    fill_wait(c, w);
    return launch_reduce_clip_adamw(params, src, grad_flat, exp_avg, exp_avg_sq, n, p, stats, step_state, workspace, workspace_bytes,
                                    &w, c->seq, st);
}
