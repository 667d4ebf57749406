// gather_rows.hip -- minibatch shuffle: flatten [T,B,...] -> [N,...] and gather rows by a permutation.
//
// Replaces process_nested_dict_for_train, rlinf/utils/nested_dict_process.py:272-285
// (value.reshape(-1, *shape[2:])[shuffle_id] for every field of the rollout batch): one launch moves
// every field (states 168 B, action/logprob rows 32 B, scalars 4 B, bool flags 1 B per sample).
// Pure data movement: bit-exact.  HBM-bound, ~2 x 312 B per sample; reads are random at row
// granularity (rows are 1..3 cache lines), writes are fully coalesced.

#include <algorithm>

#include "rlx_common.h"

namespace rlx {
namespace {

struct GatherArgs {
    rlx_gather_field f[RLX_GATHER_MAX_FIELDS];
    const int64_t* index;
    long long n_rows;
};

// grid.y = field; each thread moves one word (W = 16, 4 or 1 bytes) of one destination row.
template <typename word_t>
__device__ __forceinline__ void gather_field(const rlx_gather_field& fd, const int64_t* __restrict__ index, long long n_rows) {
    const long long wpr = fd.row_bytes / (long long)sizeof(word_t);  // words per row
    const long long total = n_rows * wpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const word_t* __restrict__ src = static_cast<const word_t*>(fd.src);
    word_t* __restrict__ dst = static_cast<word_t*>(fd.dst);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long row = i / wpr, w = i - row * wpr;
        dst[i] = src[index[row] * wpr + w];
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(GatherArgs a) {
    const rlx_gather_field& fd = a.f[blockIdx.y];
    const uintptr_t align = reinterpret_cast<uintptr_t>(fd.src) | reinterpret_cast<uintptr_t>(fd.dst) | (uintptr_t)fd.row_bytes;
    if (align % 16 == 0) gather_field<uint4>(fd, a.index, a.n_rows);
    else if (align % 8 == 0) gather_field<uint2>(fd, a.index, a.n_rows);
    else if (align % 4 == 0) gather_field<uint32_t>(fd, a.index, a.n_rows);
    else gather_field<uint8_t>(fd, a.index, a.n_rows);
}

// a6: r[b, C-1] += gamma * V(final_obs)[b] where the env finished (env_worker.py:744-758)
__global__ __launch_bounds__(256) void bootstrap_rewards_kernel(float* __restrict__ r, const uint8_t* __restrict__ flags,
                                                                const float* __restrict__ v, int B, int C, int vs, float gamma) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t i = (size_t)b * C + (C - 1);
    if (flags[i]) r[i] = fadd(r[i], fmul(gamma, v[(size_t)b * vs]));
}

// a7: one env step's outputs -> trajectory-buffer rows (rewards[t], terminations / truncations / dones[t+1])
__global__ __launch_bounds__(256) void store_env_rows_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ term,
                                                             const uint8_t* __restrict__ trunc, float* __restrict__ r_row,
                                                             uint8_t* __restrict__ d_row, uint8_t* __restrict__ te_row,
                                                             uint8_t* __restrict__ tr_row, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t te = term[i] != 0, tr = trunc[i] != 0;
    r_row[i] = rewards[i];
    te_row[i] = te;
    tr_row[i] = tr;
    d_row[i] = te | tr;  // dones = terminations | truncations (maniskill_env.py:343-350)
}

// development / calibration: one dword per lane streaming copy of n floats (known traffic: 4n bytes read, 4n written) --
// the byte-count reference for rocprofv3's FETCH_SIZE / WRITE_SIZE in the gae_scan access pattern (MI355X_MICROARCH.md:
// "calibrate on a known byte count in your own access pattern")
__global__ __launch_bounds__(64) void dev_stream_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows,
                                                             long long width) {
    const long long col = (long long)blockIdx.x * 64 + threadIdx.x;
    if (col >= width) return;
    for (long long r = 0; r < rows; ++r) dst[r * width + col] = src[r * width + col];
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" int rlx_dev_stream_copy(const float* src, float* dst, int64_t rows, int64_t width, rlx_stream_t stream) {
    RLX_REQUIRE(src && dst && rows > 0 && width > 0, "rlx_dev_stream_copy: bad argument");
    hipLaunchKernelGGL(dev_stream_copy_kernel, dim3(ceil_div(width, 64)), dim3(64), 0, static_cast<hipStream_t>(stream), src, dst,
                       (long long)rows, (long long)width);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_store_env_rows(const float* rewards, const uint8_t* terminations, const uint8_t* truncations,
                                  float* reward_row, uint8_t* done_row, uint8_t* termination_row, uint8_t* truncation_row,
                                  int64_t n, rlx_stream_t stream) {
    RLX_REQUIRE(n >= 0, "rlx_store_env_rows: negative size");
    if (n == 0) return RLX_OK;
    RLX_REQUIRE(rewards && terminations && truncations && reward_row && done_row && termination_row && truncation_row,
                "rlx_store_env_rows: NULL argument");
    hipLaunchKernelGGL(store_env_rows_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rewards,
                       terminations, truncations, reward_row, done_row, termination_row, truncation_row, (int)n);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_bootstrap_rewards(float* rewards, const uint8_t* flags, const float* bootstrap_values, int batch, int chunk,
                                     int value_stride, float gamma, rlx_stream_t stream) {
    RLX_REQUIRE(batch >= 0 && chunk >= 1 && value_stride >= 1, "rlx_bootstrap_rewards: bad sizes");
    if (batch == 0) return RLX_OK;
    RLX_REQUIRE(rewards && flags && bootstrap_values, "rlx_bootstrap_rewards: NULL argument");
    hipLaunchKernelGGL(bootstrap_rewards_kernel, dim3(ceil_div(batch, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rewards, flags, bootstrap_values, batch, chunk, value_stride, gamma);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_gather_rows(const rlx_gather_field* fields, int n_fields, const int64_t* index, int64_t n_rows,
                               rlx_stream_t stream) {
    RLX_REQUIRE(n_fields >= 0 && n_fields <= RLX_GATHER_MAX_FIELDS, "rlx_gather_rows: n_fields=%d (max %d)", n_fields,
                RLX_GATHER_MAX_FIELDS);
    RLX_REQUIRE(n_rows >= 0, "rlx_gather_rows: negative n_rows");
    if (n_fields == 0 || n_rows == 0) return RLX_OK;
    RLX_REQUIRE(fields != nullptr && index != nullptr, "rlx_gather_rows: NULL argument");
    GatherArgs a;
    long long max_words = 0;
    for (int i = 0; i < n_fields; ++i) {
        RLX_REQUIRE(fields[i].src && fields[i].dst && fields[i].row_bytes > 0, "rlx_gather_rows: field %d is invalid", i);
        a.f[i] = fields[i];
        max_words = std::max<long long>(max_words, n_rows * ((fields[i].row_bytes + 3) / 4));
    }
    a.index = index;
    a.n_rows = n_rows;
    const int gx = (int)std::max<long long>(1, std::min<long long>((max_words + 255) / 256, (long long)num_cu() * 8));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, n_fields), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
