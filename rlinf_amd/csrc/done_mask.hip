// done_mask.hip -- loss mask = "no done seen yet" prefix scan over the trajectory buffer, gfx950.
//
// Replaces compute_loss_mask, rlinf/utils/metric_utils.py:516-537:
//     flat = dones.transpose(1,2).reshape(-1,B)[-(n*C+1):];  mask = (flat.cumsum(0) == 0)[:-1]
//     mask_sum = mask.sum over time, broadcast back
// Integer/byte work: results are bit-exact.  HBM-bound at 2 B per env-step (1 read + 1 write).
//
// Lanes run along the env axis (VEC bool bytes per lane packed in one word, so a prefix-OR of the
// packed word scans VEC envs at once); the time axis of an env group is split over the waves of the
// block: pass 1 ORs each segment, the per-segment flags are exchanged through LDS, pass 2 re-reads
// its segment (L1/L2-resident) with the incoming "seen" word and writes mask bytes + counts.

#include "rlx_common.h"

namespace rlx {
namespace {

template <int VEC> struct Bw;
template <> struct Bw<1> { typedef uint8_t type; static constexpr uint32_t ones = 0x01u; };
template <> struct Bw<4> { typedef uint32_t type; static constexpr uint32_t ones = 0x01010101u; };

template <int VEC>
__global__ __launch_bounds__(512) void done_prefix_mask_c1(const uint8_t* __restrict__ d, uint8_t* __restrict__ m,
                                                            int64_t* __restrict__ cnt, int T, int B) {
    extern __shared__ uint32_t sm[];
    typedef typename Bw<VEC>::type word_t;
    const int lane = threadIdx.x & 63;
    const int nseg = blockDim.x >> 6;
    const int seg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long e0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    const bool active = e0 < B;
    const int seg_len = (T + nseg - 1) / nseg;
    const int t_lo = seg * seg_len, t_hi = min(T, t_lo + seg_len);
    uint32_t* s_any = sm;                       // [nseg][64]
    uint32_t* s_cnt = sm + nseg * 64;           // [nseg][64*VEC]

    uint32_t any = 0;
    if (active && nseg > 1) {
#pragma unroll 8
        for (int t = t_lo; t < t_hi; ++t) any |= (uint32_t)*reinterpret_cast<const word_t*>(d + (size_t)t * B + e0);
    }
    if (nseg > 1) {
        s_any[seg * 64 + lane] = any;
        __syncthreads();
    }
    uint32_t seen = 0;
    for (int s = 0; s < seg; ++s) seen |= s_any[s * 64 + lane];
    uint32_t c[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) c[k] = 0;
    if (active) {
#pragma unroll 8
        for (int t = t_lo; t < t_hi; ++t) {
            seen |= (uint32_t)*reinterpret_cast<const word_t*>(d + (size_t)t * B + e0);
            const uint32_t ok = (~seen) & Bw<VEC>::ones;  // bool bytes are 0/1
            *reinterpret_cast<word_t*>(m + (size_t)t * B + e0) = (word_t)ok;
#pragma unroll
            for (int k = 0; k < VEC; ++k) c[k] += (ok >> (8 * k)) & 1u;
        }
    }
    if (nseg > 1) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) s_cnt[(seg * 64 + lane) * VEC + k] = c[k];
        __syncthreads();
        if (seg == 0 && active) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                uint32_t tot = 0;
                for (int s = 0; s < nseg; ++s) tot += s_cnt[(s * 64 + lane) * VEC + k];
                cnt[e0 + k] = (int64_t)tot;
            }
        }
    } else if (active) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) cnt[e0 + k] = (int64_t)c[k];
    }
}

// Generic time-chunk layout: flat step f = k*C + c at ((k*B + b)*C + c); mask row t looks at done
// rows (C-1) .. (C-1)+t of the flattened [(n+1)*C, B] matrix.
__global__ __launch_bounds__(64) void done_prefix_mask_chunked(const uint8_t* __restrict__ d, uint8_t* __restrict__ m,
                                                               int64_t* __restrict__ cnt, int T, int B, int C) {
    const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= (size_t)B) return;
    bool seen = false;
    int64_t c = 0;
    for (int t = 0; t < T; ++t) {
        const int f = t + (C - 1);
        seen = seen || d[((size_t)(f / C) * B + b) * C + (f % C)] != 0;
        m[((size_t)(t / C) * B + b) * C + (t % C)] = seen ? 0 : 1;
        c += seen ? 0 : 1;
    }
    cnt[b] = c;
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" int rlx_done_prefix_mask(const uint8_t* dones, uint8_t* loss_mask, int64_t* mask_sum, int n_chunk,
                                    int batch, int chunk, rlx_stream_t stream) {
    RLX_REQUIRE(n_chunk >= 0 && batch >= 0 && chunk >= 1, "rlx_done_prefix_mask: bad sizes");
    if (batch == 0) return RLX_OK;
    RLX_REQUIRE(dones && mask_sum && (loss_mask || n_chunk == 0), "rlx_done_prefix_mask: NULL argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T = n_chunk * chunk;
    if (chunk != 1) {
        hipLaunchKernelGGL(done_prefix_mask_chunked, dim3(ceil_div(batch, 64)), dim3(64), 0, s, dones, loss_mask,
                           mask_sum, T, batch, chunk);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
    const bool vec4 = batch % 4 == 0 && (reinterpret_cast<uintptr_t>(dones) | reinterpret_cast<uintptr_t>(loss_mask)) % 4 == 0 &&
                      batch / 256 >= num_cu();
    const int vec = vec4 ? 4 : 1;
    const int groups = ceil_div(batch, 64 * vec);
    int nseg = 1;
    while (nseg < 8 && groups * nseg < 4 * num_cu() && T / (nseg * 2) >= 8) nseg *= 2;
    const size_t lds = nseg > 1 ? (size_t)nseg * 64 * (1 + vec) * sizeof(uint32_t) : 0;
    if (vec == 4)
        hipLaunchKernelGGL(done_prefix_mask_c1<4>, dim3(groups), dim3(64 * nseg), lds, s, dones, loss_mask, mask_sum, T, batch);
    else
        hipLaunchKernelGGL(done_prefix_mask_c1<1>, dim3(groups), dim3(64 * nseg), lds, s, dones, loss_mask, mask_sum, T, batch);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
