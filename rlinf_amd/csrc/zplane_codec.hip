// zplane_codec.hip -- lossless codec for the weight-patch transport streams (rows, cols, value bytes), gfx950.
//
// Replaces NVCompCompressor._compress_tensor / _decompress_tensor (rlinf/hybrid_engines/weight_syncer/compressor.py:148-199: the
// three patch fields go through nvCOMP's LZ4 codec as byte streams) behind the same PatchCompressor / CompressedWeightPatch
// surface (patch_syncer.py:205-250).  nvCOMP is NVIDIA-only and its container is not a public format, so the PAYLOAD here is this
// build's own; what is kept is the transport contract (field names, dtype codes, "compressed byte tensor + dtype code" per
// field, byte-exact round trip).
//
// Why not LZ4: the streams are delta-encoded COO indices (rows: almost all 0 with an occasional 1; cols: small gaps whose high
// bytes are 0) and raw bf16 / f32 value bytes (the mantissa bytes are noise; LZ4's >= 4-byte matches find nothing in them).  What
// compresses is ZERO BYTES, plane by plane -- which is also what a 64-lane wavefront is natively good at: one __ballot over 64
// consecutive bytes of a plane IS the 64-bit occupancy mask, popcounts give the packed positions.  So:
//
//   format "RLXZ" v1 (little endian, everything 8-byte aligned):
//     header   u32 magic 'RLXZ' | u8 version | u8 elem_size | u16 block_log2 (12) | u64 n_elems | u64 payload_bytes
//     directory  n_blocks x elem_size entries, u64 each: (mode << 62) | payload offset          [entry = block * elem_size + plane]
//     payload  per (block of 4096 elements, byte plane):
//        mode 0  the plane is all zero                                   -- nothing stored
//        mode 1  u64 top mask (bit g: 64-byte group g has a nonzero byte) | u64 group mask per set bit | the nonzero bytes (+ pad to 8)
//        mode 2  the plane's bytes as they are (+ pad to 8)              -- chosen whenever neither masked form is smaller
//        mode 3  as mode 1, of the plane XORed with itself shifted by one byte (b[i] ^ b[i-1], b[-1] = 0): runs of EQUAL bytes
//                become zeros -- the dense-update case, where the column deltas are all 1 and the row deltas all 0
//
// compress = measure (sizes per entry) -> scan (offsets, one workgroup) -> pack; decompress = one launch.  A workgroup owns one
// block: it loads the block's elements with coalesced 16-byte loads and splits them into byte planes in LDS (byte streams skip
// this: a lane's 16 bytes come straight from global memory); each of its four waves then owns one 1024-byte span of every plane,
// 16 bytes per lane in registers (see "cells" below): masks from register arithmetic, offsets from one wave scan per cell, the
// nonzero bytes compacted through LDS and moved to / from global memory in whole 8-byte words.  HBM-bound: the input is read
// twice (measure, pack), the output written once.
// Integer / byte work: bit-exact round trip (tests/test_gpu_weight_patch.py, against a numpy restatement of the format).

#include <stdlib.h>

#include <algorithm>
#include <string.h>

#include "rlx_common.h"

namespace rlx {
namespace {

constexpr int kBlockLog2 = 12;
constexpr int kBlock = 1 << kBlockLog2;      // elements per block = bytes per plane-block
constexpr int kGroups = kBlock / 64;         // 64-byte groups per plane-block (= 64: one top-mask word)
constexpr uint32_t kMagic = 0x5A584C52u;     // "RLXZ"
constexpr size_t kHeaderBytes = 24;
constexpr int kThreads = 256;

struct Header {
    uint32_t magic;
    uint8_t version, elem_size;
    uint16_t block_log2;
    uint64_t n_elems, payload_bytes;
};
static_assert(sizeof(Header) == kHeaderBytes, "header layout");

__device__ __forceinline__ uint64_t lanes_below(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }
__host__ __device__ inline size_t pad8(size_t x) { return (x + 7) & ~(size_t)7; }

// v_perm_b32: result byte i = byte sel[i] of the 8-byte pool {lo (0..3), hi (4..7)}
__device__ __forceinline__ uint32_t bperm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// 4 x 4 byte transpose: elements w[0..3] (one dword each) -> p[j] = (w0.bj, w1.bj, w2.bj, w3.bj)
__device__ __forceinline__ void transpose4(const uint32_t (&w)[4], uint32_t (&p)[4]) {
    const uint32_t a = bperm(w[1], w[0], 0x05010400u), b = bperm(w[1], w[0], 0x07030602u);  // (w0.b0 w1.b0 w0.b1 w1.b1), (.. b2 .. b3)
    const uint32_t c = bperm(w[3], w[2], 0x05010400u), d = bperm(w[3], w[2], 0x07030602u);
    p[0] = bperm(c, a, 0x05040100u);
    p[1] = bperm(c, a, 0x07060302u);
    p[2] = bperm(d, b, 0x05040100u);
    p[3] = bperm(d, b, 0x07060302u);
}

// Load one block (up to 4096 elements of ES bytes) and split it into byte planes in LDS; bytes past n_bytes read as zero.
// Coalesced 16-byte global loads; the byte de-interleave runs in registers (v_perm_b32), so a lane's LDS stores are whole words:
// ES = 1 one b128, ES = 2 two b64, ES = 4 four b32, ES = 8 eight b32 per 32 input bytes (byte stores made this the codec's
// slowest phase by far).
template <int ES>
__device__ __forceinline__ void load_planes(const uint8_t* __restrict__ in, size_t block_byte0, size_t n_bytes_total, uint8_t* planes) {
    constexpr int BYTES = kBlock * ES, STEP = ES == 8 ? 32 : 16;  // input bytes per lane and trip
    const uint8_t* src = in + block_byte0;
    const size_t avail = n_bytes_total > block_byte0 ? n_bytes_total - block_byte0 : 0;
    for (int c = threadIdx.x; c < BYTES / STEP; c += kThreads) {
        const size_t b0 = (size_t)c * STEP;
        union { uint4 v[STEP / 16]; uint32_t w[STEP / 4]; uint8_t b[STEP]; } u;
        if (b0 + STEP <= avail) {
#pragma unroll
            for (int q = 0; q < STEP / 16; ++q) u.v[q] = *reinterpret_cast<const uint4*>(src + b0 + 16 * q);
        } else {
#pragma unroll
            for (int k = 0; k < STEP; ++k) u.b[k] = (b0 + k < avail) ? src[b0 + k] : (uint8_t)0;
        }
        const int e0 = (int)(b0 / ES);  // first element of this chunk
        if constexpr (ES == 1) {
            *reinterpret_cast<uint4*>(planes + e0) = u.v[0];
        } else if constexpr (ES == 2) {  // 8 elements: even bytes -> plane 0, odd bytes -> plane 1
            uint2 p0, p1;
            p0.x = bperm(u.w[1], u.w[0], 0x06040200u); p1.x = bperm(u.w[1], u.w[0], 0x07050301u);
            p0.y = bperm(u.w[3], u.w[2], 0x06040200u); p1.y = bperm(u.w[3], u.w[2], 0x07050301u);
            *reinterpret_cast<uint2*>(planes + e0) = p0;
            *reinterpret_cast<uint2*>(planes + kBlock + e0) = p1;
        } else if constexpr (ES == 4) {  // 4 elements
            uint32_t w[4] = {u.w[0], u.w[1], u.w[2], u.w[3]}, p[4];
            transpose4(w, p);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t*>(planes + j * kBlock + e0) = p[j];
        } else {  // ES == 8: 4 elements = 8 dwords; low dwords feed planes 0..3, high dwords planes 4..7
            uint32_t lo[4] = {u.w[0], u.w[2], u.w[4], u.w[6]}, hi[4] = {u.w[1], u.w[3], u.w[5], u.w[7]}, p[4], q[4];
            transpose4(lo, p);
            transpose4(hi, q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<uint32_t*>(planes + j * kBlock + e0) = p[j];
                *reinterpret_cast<uint32_t*>(planes + (4 + j) * kBlock + e0) = q[j];
            }
        }
    }
}

// inverse of load_planes: the block's planes -> interleaved elements, 16 (ES = 8: 32) output bytes per lane and trip
template <int ES>
__device__ __forceinline__ void store_planes(const uint8_t* planes, uint8_t* __restrict__ dst, size_t valid) {
    constexpr int BYTES = kBlock * ES, STEP = ES == 8 ? 32 : 16;
    for (int c = threadIdx.x; c < BYTES / STEP; c += kThreads) {
        const size_t b0 = (size_t)c * STEP;
        if (b0 >= valid) break;
        union { uint4 v[STEP / 16]; uint32_t w[STEP / 4]; uint8_t b[STEP]; } u;
        const int e0 = (int)(b0 / ES);
        if constexpr (ES == 1) {
            u.v[0] = *reinterpret_cast<const uint4*>(planes + e0);
        } else if constexpr (ES == 2) {
            const uint2 p0 = *reinterpret_cast<const uint2*>(planes + e0), p1 = *reinterpret_cast<const uint2*>(planes + kBlock + e0);
            u.w[0] = bperm(p1.x, p0.x, 0x05010400u); u.w[1] = bperm(p1.x, p0.x, 0x07030602u);  // (p0.b0 p1.b0 p0.b1 p1.b1), (.. b2 .. b3)
            u.w[2] = bperm(p1.y, p0.y, 0x05010400u); u.w[3] = bperm(p1.y, p0.y, 0x07030602u);
        } else if constexpr (ES == 4) {  // the 4 x 4 byte transpose is its own inverse
            uint32_t p[4], w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = *reinterpret_cast<const uint32_t*>(planes + j * kBlock + e0);
            transpose4(p, w);
#pragma unroll
            for (int j = 0; j < 4; ++j) u.w[j] = w[j];
        } else {
            uint32_t p[4], q[4], lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = *reinterpret_cast<const uint32_t*>(planes + j * kBlock + e0);
                q[j] = *reinterpret_cast<const uint32_t*>(planes + (4 + j) * kBlock + e0);
            }
            transpose4(p, lo);
            transpose4(q, hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) { u.w[2 * j] = lo[j]; u.w[2 * j + 1] = hi[j]; }
        }
        if (b0 + STEP <= valid) {
#pragma unroll
            for (int q = 0; q < STEP / 16; ++q) *reinterpret_cast<uint4*>(dst + b0 + 16 * q) = u.v[q];
        } else {
            for (int k = 0; k < STEP && b0 + k < valid; ++k) dst[b0 + k] = u.b[k];
        }
    }
}

// 4-bit "byte is nonzero" mask of a dword (bit j = byte j): the classic has-zero-byte carry trick, then a multiply that gathers
// bits 0, 8, 16, 24 into one nibble (all partial products land on distinct bit positions: no carries)
__device__ __forceinline__ uint32_t nz4(uint32_t w) {
    const uint32_t t = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
    return ((t >> 7) * 0x01020408u) >> 24;
}
__device__ __forceinline__ uint32_t nz16(const uint4& v) { return nz4(v.x) | (nz4(v.y) << 4) | (nz4(v.z) << 8) | (nz4(v.w) << 12); }

// ---- cells ----------------------------------------------------------------------------------------------------------------------
// The unit of wave work is a CELL: one 1024-byte span of one plane-block.  Lane L holds bytes [16 L, 16 L + 16) of the span in a
// uint4 (four lanes make a 64-byte group, so a lane's 16-bit nonzero mask is a quarter of the group's 64-bit mask, and 64 of them
// laid out as u16 in lane order ARE the span's 16 group masks in memory order).  Wave w of the workgroup owns span w of EVERY plane
// of the block: all four waves are busy for every element size (one wave per plane left three idle on byte streams), and every
// cross-lane step -- counts, offsets, the XOR filter's running value -- is one wave scan per cell instead of one per group.
constexpr int kSpan = 1024, kSpans = kBlock / kSpan;
static_assert(kSpans == kThreads / 64, "one wave per span");

// the lane's 16 bytes under the XOR filter (b[i] ^ b[i-1]); `before` = the byte in front of the span (0 for the plane's first)
__device__ __forceinline__ uint4 xor_filter(const uint4& v, uint32_t before) {
    const int lane = threadIdx.x & 63;
    // the last byte of the lane below: wave_shr:1 (a DPP move on the VALU; __shfl_up would be a ds_bpermute through the LDS pipe)
    uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v.w >> 24), 0x138, 0xf, 0xf, false);
    if (lane == 0) prev = before;
    uint4 x;
    x.x = v.x ^ ((v.x << 8) | prev);
    x.y = v.y ^ ((v.y << 8) | (v.x >> 24));
    x.z = v.z ^ ((v.z << 8) | (v.y >> 24));
    x.w = v.w ^ ((v.w << 8) | (v.z >> 24));
    return x;
}

// inverse within the lane: byte j <- XOR of bytes 0..j; returns the lane's total (its last byte) for the scan across lanes
__device__ __forceinline__ uint32_t xor_prefix16(uint4& v) {
    v.x ^= v.x << 8; v.x ^= v.x << 16;
    v.y ^= v.y << 8; v.y ^= v.y << 16; v.y ^= (v.x >> 24) * 0x01010101u;
    v.z ^= v.z << 8; v.z ^= v.z << 16; v.z ^= (v.y >> 24) * 0x01010101u;
    v.w ^= v.w << 8; v.w ^= v.w << 16; v.w ^= (v.z >> 24) * 0x01010101u;
    return v.w >> 24;
}

// Cross-lane sums of SMALL values without the LDS pipe: one ballot per bit of the value; v_mbcnt counts the set bits below a
// lane (a prefix), s_bcnt1 all of them (a total).  A 6-step __shfl_up / __shfl_xor ladder is six ds_bpermute trips through the
// CU's one LDS pipe per value, shared by all four SIMDs -- it was what the codec's launches were bound by.
__device__ __forceinline__ int lane_prefix_count(uint64_t mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive prefix sum over the lanes of x in [0, 31]
__device__ __forceinline__ int wave_incl_sum(int x) {
    int excl = 0;
#pragma unroll
    for (int b2 = 0; b2 < 5; ++b2) excl += lane_prefix_count(__ballot((x >> b2) & 1)) << b2;
    return excl + x;
}

// sum over the lanes of x in [0, 31] (wave-uniform)
__device__ __forceinline__ int wave_total(int x) {
    int t = 0;
#pragma unroll
    for (int b2 = 0; b2 < 5; ++b2) t += __popcll(__ballot((x >> b2) & 1)) << b2;
    return t;
}

// inclusive prefix XOR over the lanes of an 8-bit value
__device__ __forceinline__ uint32_t wave_incl_xor(uint32_t x) {
    uint32_t r = 0;
#pragma unroll
    for (int b2 = 0; b2 < 8; ++b2) {
        const uint32_t bit = (x >> b2) & 1u;
        r |= (((uint32_t)lane_prefix_count(__ballot(bit != 0)) + bit) & 1u) << b2;
    }
    return r;
}

// the lane's 16 bytes of cell (plane 0, span `span`) of a BYTE stream straight from global memory (no plane split to do);
// bytes past `avail` read as zero
__device__ __forceinline__ uint4 load_cell_bytes(const uint8_t* __restrict__ src, size_t avail, int span) {
    const size_t b0 = (size_t)span * kSpan + (threadIdx.x & 63) * 16;
    if (b0 + 16 <= avail) return *reinterpret_cast<const uint4*>(src + b0);
    union { uint4 v; uint8_t b[16]; } u;
#pragma unroll
    for (int k = 0; k < 16; ++k) u.b[k] = (b0 + k < avail) ? src[b0 + k] : (uint8_t)0;
    return u.v;
}

// byte k (0..15, compile time) of a uint4
template <int K>
__device__ __forceinline__ uint32_t byte_of(const uint4& v) {
    const uint32_t w = K < 4 ? v.x : (K < 8 ? v.y : (K < 12 ? v.z : v.w));
    return (w >> ((K & 3) * 8)) & 0xffu;
}
template <int K>
__device__ __forceinline__ void or_byte(uint4& v, uint32_t b) {
    const uint32_t s = b << ((K & 3) * 8);
    if (K < 4) v.x |= s; else if (K < 8) v.y |= s; else if (K < 12) v.z |= s; else v.w |= s;
}

// the lane's nonzero bytes (mask m) -> stage[off ...], in byte order
template <int K = 0>
__device__ __forceinline__ void compact16(const uint4& v, uint32_t m, uint8_t* stage, int off) {
    if constexpr (K < 16) {
        if ((m >> K) & 1u) stage[off++] = (uint8_t)byte_of<K>(v);
        compact16<K + 1>(v, m, stage, off);
    }
}
// inverse: stage[off ...] -> the byte positions set in m
template <int K = 0>
__device__ __forceinline__ void expand16(uint4& v, uint32_t m, const uint8_t* stage, int off) {
    if constexpr (K < 16) {
        if ((m >> K) & 1u) or_byte<K>(v, stage[off++]);
        expand16<K + 1>(v, m, stage, off);
    }
}

// A workgroup works on Q "virtual planes": the ES planes of one block, or -- byte streams, whose blocks are only 4 KiB --
// plane 0 of kBytesBlocks consecutive blocks (more bytes in flight per workgroup, fewer workgroup launches).
constexpr int kBytesBlocks = 4;
template <int ES>
struct Geo {
    static constexpr int BPW = ES == 1 ? kBytesBlocks : 1;  // blocks per workgroup
    static constexpr int Q = BPW * ES;                      // virtual planes per workgroup: q -> (block q / ES, plane q % ES)
};

template <int ES>
__global__ __launch_bounds__(kThreads) void zplane_measure(const uint8_t* __restrict__ in, long long n_elems, long long n_blocks,
                                                           uint64_t* __restrict__ sizes) {
    constexpr int BPW = Geo<ES>::BPW, Q = Geo<ES>::Q;
    __shared__ __attribute__((aligned(16))) uint8_t planes[ES == 1 ? 16 : ES * kBlock];
    __shared__ int cnt[2][kSpans][Q][4];  // nz, ng, nzx, ngx of each cell (plain stores, summed by the deciding thread); two sets, alternating between trips
    const int span = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t total = (size_t)n_elems * ES;
    const long long n_wg = (n_blocks + BPW - 1) / BPW;
    // grid-stride over the workgroup-sized pieces (the host caps the grid for byte streams, see rlx_zplane_compress)
    int par = 0;
    for (long long wg = blockIdx.x; wg < n_wg; wg += gridDim.x, par ^= 1) {
        const long long blk0 = wg * BPW;
        if constexpr (ES > 1) load_planes<ES>(in, (size_t)blk0 * kBlock * ES, total, planes);
        __syncthreads();  // planes staged (and the previous trip's decisions have read their counters: this trip writes the other set)
        uint4 v[Q];
        uint32_t before[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {  // all of the workgroup's loads in flight before the first use
            before[q] = 0;
            if constexpr (ES == 1) {
                const size_t byte0 = (size_t)(blk0 + q) * kBlock, avail = total > byte0 ? total - byte0 : 0;
                v[q] = load_cell_bytes(in + byte0, avail, span);
                if (span > 0 && (size_t)span * kSpan - 1 < avail) before[q] = in[byte0 + (size_t)span * kSpan - 1];
            } else {
                v[q] = *reinterpret_cast<const uint4*>(planes + q * kBlock + span * kSpan + lane * 16);
                if (span > 0) before[q] = planes[q * kBlock + span * kSpan - 1];
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint4 x = xor_filter(v[q], before[q]);
            const uint32_t m = nz16(v[q]), mx = nz16(x);
            const int c = wave_total(__popc(m)), cx = wave_total(__popc(mx));
            // a group is nonzero iff any of its four lanes holds a nonzero byte: OR over the quad, counted once per quad
            const uint64_t any = __ballot(m != 0), anyx = __ballot(mx != 0);
            const uint64_t g = any | (any >> 1) | (any >> 2) | (any >> 3), gx = anyx | (anyx >> 1) | (anyx >> 2) | (anyx >> 3);
            if (lane == 0) {
                cnt[par][span][q][0] = c;
                cnt[par][span][q][1] = __popcll(g & 0x1111111111111111ull);
                cnt[par][span][q][2] = cx;
                cnt[par][span][q][3] = __popcll(gx & 0x1111111111111111ull);
            }
        }
        __syncthreads();  // this trip's counters are complete; nobody reads planes[] any more
        if (threadIdx.x < Q) {
            const int q = threadIdx.x;
            const long long blk = blk0 + q / ES;
            int nz = 0, ng = 0, nzx = 0, ngx = 0;
#pragma unroll
            for (int s2 = 0; s2 < kSpans; ++s2) {
                nz += cnt[par][s2][q][0]; ng += cnt[par][s2][q][1]; nzx += cnt[par][s2][q][2]; ngx += cnt[par][s2][q][3];
            }
            if (blk < n_blocks) {
                const long long in_block = min((long long)kBlock, n_elems - blk * kBlock);
                const size_t masked = 8 + 8 * (size_t)ng + pad8((size_t)nz), xored = 8 + 8 * (size_t)ngx + pad8((size_t)nzx),
                             raw = pad8((size_t)in_block);
                uint64_t mode, size;
                if (nz == 0) { mode = 0; size = 0; }
                else if (masked <= xored && masked < raw) { mode = 1; size = masked; }
                else if (xored < raw) { mode = 3; size = xored; }
                else { mode = 2; size = raw; }
                sizes[blk * ES + q % ES] = (mode << 62) | size;
            }
        }
    }
}

// Offsets: exclusive scan of the entry sizes in two levels.  (One workgroup walking all entries was the slowest launch of the
// codec: ~90 us for 49 k entries, every one of its 48 dependent trips a full memory latency.)
//   scan_tiles   one workgroup per tile of 4096 entries: directory[i] = mode | offset WITHIN the tile, tile_total[t]
//   scan_bases   one workgroup: tile_total -> tile_base (exclusive), the header, the stream length
// The pack launch adds tile_base to its own entries and stores the final directory words.
constexpr int kScanTile = 4096;
constexpr uint64_t kSizeMask = (1ull << 62) - 1;

__device__ __forceinline__ uint64_t block_excl_scan_1024(uint64_t x, uint64_t* s_wave, uint64_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = (uint64_t)__shfl_up((long long)incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint64_t t = s_wave[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();  // s_wave may be reused by the caller's next round
    total = tot;
    return base + incl - x;
}

__global__ __launch_bounds__(1024) void zplane_scan_tiles(const uint64_t* __restrict__ sizes, long long n_entries, uint64_t* __restrict__ directory,
                                                          uint64_t* __restrict__ tile_total) {
    __shared__ uint64_t s_wave[16];
    const long long i0 = (long long)blockIdx.x * kScanTile + threadIdx.x * 4;
    uint64_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = i0 + k < n_entries ? sizes[i0 + k] : 0ull;
    const uint64_t mine = (e[0] & kSizeMask) + (e[1] & kSizeMask) + (e[2] & kSizeMask) + (e[3] & kSizeMask);
    uint64_t total;
    uint64_t run = block_excl_scan_1024(mine, s_wave, total);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n_entries) directory[i0 + k] = (e[k] & ~kSizeMask) | run;
        run += e[k] & kSizeMask;
    }
    if (threadIdx.x == 0) tile_total[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void zplane_scan_bases(uint64_t* __restrict__ tile_total, long long n_tiles, long long n_entries,
                                                          Header* __restrict__ header, long long n_elems, int elem_size,
                                                          uint64_t* __restrict__ out_bytes) {
    __shared__ uint64_t s_wave[16];
    uint64_t carry = 0;
    for (long long t0 = 0; t0 < n_tiles; t0 += 1024) {
        const long long t = t0 + threadIdx.x;
        const uint64_t x = t < n_tiles ? tile_total[t] : 0ull;
        uint64_t total;
        const uint64_t excl = block_excl_scan_1024(x, s_wave, total);
        if (t < n_tiles) tile_total[t] = carry + excl;  // in place: tile_total becomes tile_base
        carry += total;
    }
    if (threadIdx.x == 0) {
        header->magic = kMagic;
        header->version = 1;
        header->elem_size = (uint8_t)elem_size;
        header->block_log2 = kBlockLog2;
        header->n_elems = (uint64_t)n_elems;
        header->payload_bytes = carry;
        *out_bytes = kHeaderBytes + 8ull * (uint64_t)n_entries + carry;
    }
}

// pack: every cell's bytes are taken into registers first (so the plane area of LDS can be reused as the staging buffer), the
// nonzero ones are compacted into LDS at their final order, and the workgroup then copies masks and bytes out in whole 8-byte words.
template <int ES>
__global__ __launch_bounds__(kThreads) void zplane_pack(const uint8_t* __restrict__ in, long long n_elems, long long n_blocks,
                                                        uint64_t* __restrict__ directory, const uint64_t* __restrict__ tile_base,
                                                        uint8_t* __restrict__ payload) {
    constexpr int BPW = Geo<ES>::BPW, Q = Geo<ES>::Q;
    __shared__ __attribute__((aligned(16))) uint8_t planes[Q * kBlock];
    __shared__ __attribute__((aligned(8))) uint16_t masks[Q][kThreads];  // = u64 group mask [Q][64]
    __shared__ int span_cnt[Q][kSpans];
    __shared__ int s_ng[Q];
    const long long blk0 = (long long)blockIdx.x * BPW;
    const int span = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t total = (size_t)n_elems * ES;
    if constexpr (ES > 1) {
        load_planes<ES>(in, (size_t)blk0 * kBlock * ES, total, planes);
        __syncthreads();
    }
    uint4 v[Q];
    uint32_t m[Q];
    int incl[Q], mode[Q], n8_raw[Q];
    uint8_t* dst[Q];
    uint64_t word[Q];  // the final directory words: mode | offset in the payload
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const long long blk = blk0 + q / ES, entry = blk * ES + q % ES;
        m[q] = 0;
        incl[q] = 0;
        mode[q] = 0;
        n8_raw[q] = 0;
        dst[q] = payload;
        word[q] = 0;
        if (blk >= n_blocks) continue;
        const uint64_t e = directory[entry];  // mode | offset within its scan tile
        const uint64_t off = (e & kSizeMask) + tile_base[entry / kScanTile];
        word[q] = (e & ~kSizeMask) | off;
        mode[q] = (int)(e >> 62);
        dst[q] = payload + off;
        n8_raw[q] = (int)pad8((size_t)min((long long)kBlock, n_elems - blk * kBlock));
    }
    __syncthreads();  // every thread of the workgroup (the entries' only reader) has its copy: now they may be overwritten
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (threadIdx.x == q && blk0 + q / ES < n_blocks) directory[(blk0 + q / ES) * ES + q % ES] = word[q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (mode[q] == 0) continue;
        uint32_t before = 0;
        if constexpr (ES == 1) {
            const size_t byte0 = (size_t)(blk0 + q) * kBlock, avail = total > byte0 ? total - byte0 : 0;
            v[q] = load_cell_bytes(in + byte0, avail, span);
            if (mode[q] == 3 && span > 0 && (size_t)span * kSpan - 1 < avail) before = in[byte0 + (size_t)span * kSpan - 1];
        } else {
            v[q] = *reinterpret_cast<const uint4*>(planes + q * kBlock + span * kSpan + lane * 16);
            if (mode[q] == 3 && span > 0) before = planes[q * kBlock + span * kSpan - 1];
        }
        if (mode[q] == 2) continue;
        if (mode[q] == 3) v[q] = xor_filter(v[q], before);
        m[q] = nz16(v[q]);
        masks[q][threadIdx.x] = (uint16_t)m[q];
        incl[q] = wave_incl_sum(__popc(m[q]));
        if (lane == 63) span_cnt[q][span] = incl[q];
    }
    __syncthreads();  // every cell is in registers: planes[] is free; masks and span counts are complete
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (mode[q] == 2) {
            if constexpr (ES == 1) {  // byte stream, raw: the lane's 16 bytes go straight out (the plane was never staged)
                const int b0 = span * kSpan + lane * 16;
                if (b0 + 8 <= n8_raw[q]) *reinterpret_cast<uint2*>(dst[q] + b0) = uint2{v[q].x, v[q].y};
                if (b0 + 16 <= n8_raw[q]) *reinterpret_cast<uint2*>(dst[q] + b0 + 8) = uint2{v[q].z, v[q].w};
            }
            continue;
        }
        if (mode[q] == 0) continue;
        int base = 0;
#pragma unroll
        for (int s2 = 0; s2 < kSpans; ++s2)
            if (s2 < span) base += span_cnt[q][s2];
        compact16(v[q], m[q], planes + q * kBlock, base + incl[q] - __popc(m[q]));
        if (span == (q & (kSpans - 1))) {  // one wave writes the plane's [top][group masks] and pads the byte area
            const uint64_t gm = reinterpret_cast<const uint64_t*>(&masks[q][0])[lane];
            const uint64_t top = __ballot(gm != 0);
            uint64_t* d64 = reinterpret_cast<uint64_t*>(dst[q]);
            if (gm != 0) d64[1 + __popcll(top & lanes_below(lane))] = gm;
            if (lane == 0) {
                d64[0] = top;
                s_ng[q] = __popcll(top);
            }
            const int nz = span_cnt[q][0] + span_cnt[q][1] + span_cnt[q][2] + span_cnt[q][3];
            if (lane < (int)(pad8((size_t)nz) - nz)) planes[q * kBlock + nz + lane] = 0;  // deterministic pad bytes (nobody compacts past nz)
        }
    }
    __syncthreads();  // the staged bytes are complete
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (mode[q] == 0) continue;
        if (mode[q] == 2) {
            if constexpr (ES > 1)
                for (int i = threadIdx.x * 8; i < n8_raw[q]; i += kThreads * 8)
                    *reinterpret_cast<uint2*>(dst[q] + i) = *reinterpret_cast<const uint2*>(planes + q * kBlock + i);
            continue;
        }
        const int nz = span_cnt[q][0] + span_cnt[q][1] + span_cnt[q][2] + span_cnt[q][3];
        uint8_t* bytes = dst[q] + 8 + 8 * (size_t)s_ng[q];
        for (int i = threadIdx.x * 8; i < (int)pad8((size_t)nz); i += kThreads * 8)
            *reinterpret_cast<uint2*>(bytes + i) = *reinterpret_cast<const uint2*>(planes + q * kBlock + i);
    }
}

// ---- single-pass encoder ---------------------------------------------------------------------------------------------------------
// measure + offsets + pack in ONE launch: the input is read once.  A workgroup takes its cells into registers, sizes its planes
// (as zplane_measure), and needs the payload offset of its first plane = the sum of the sizes of everything before it.  That sum
// comes from a decoupled look-back: every workgroup publishes (flag, value) words -- first its own total ("aggregate"), later the
// total of everything up to and including itself ("prefix") -- and one of its waves walks the words of its predecessors, 64 at a
// time, adding aggregates until it meets a prefix.  While that wave looks back, the other three compact their cells into LDS (the
// compaction does not need the offset; only the copy-out does).  The stream comes out byte-identical to the multi-launch encoder
// and to the numpy restatement: offsets follow block order, whatever order the workgroups ran in.
// Forward progress: the block a workgroup encodes is its TICKET (an atomic counter taken at start), not blockIdx.x, so everything a
// workgroup waits for has already started; a bounded wait (2 s) turns a lost predecessor into a length-0 result instead of a hang.
// MEASURED (profiles/r03_zplane_single_pass_*.{jsonl,txt}; 200 MB streams): 205-239 us against 124-129 us for measure + two scan
// launches + pack.  A workgroup's work is short (one 4-16 KiB block, ~3 us), ~2000 of them are resident at once and they publish
// their aggregates together, so a look-back has to walk back over most of the resident set before it meets a prefix: dozens of
// dependent device-scope reads per workgroup, more than the second read of the input costs.  The multi-launch encoder stays the
// default; this one is kept selectable (RLX_ZPLANE_SINGLE_PASS=1) and tested for byte identity.
constexpr uint64_t kFlagAggregate = 1ull << 62, kFlagPrefix = 2ull << 62;

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += (uint64_t)__shfl_xor((long long)x, d, 64);
    return x;
}

// wave-wide: exclusive prefix of workgroup `wg` (its own total `mine` already known); publishes aggregate then prefix
__device__ __forceinline__ uint64_t lookback_exclusive(unsigned long long* states, long long wg, uint64_t mine, unsigned long long* fail) {
    const int lane = threadIdx.x & 63;
    unsigned long long* my = states + wg;
    if (wg == 0) {
        if (lane == 0) __hip_atomic_store(my, kFlagPrefix | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(my, kFlagAggregate | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t excl = 0;
    long long pos = wg - 1;
    const long long t0 = wall_clock64();
    while (true) {
        const long long idx = pos - lane;
        const uint64_t w = idx >= 0 ? __hip_atomic_load(states + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFlagPrefix;  // nothing before block 0
        const int flag = (int)(w >> 62);
        const uint64_t pmask = __ballot(flag == 2), emask = __ballot(flag == 0);
        const int first_p = pmask ? __builtin_ctzll(pmask) : 64;
        const uint64_t needed = first_p >= 63 ? ~0ull : ((2ull << first_p) - 1);  // lanes 0 .. first_p
        if (emask & needed) {  // a predecessor in the window has not published yet
            if (wall_clock64() - t0 > 200000000ll) {  // 2 s at 100 MHz
                if (lane == 0) atomicExch(fail, 1ull);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        excl += wave_sum_u64(lane <= first_p ? (w & kSizeMask) : 0ull);
        if (first_p < 64) break;
        pos -= 64;
    }
    if (lane == 0) __hip_atomic_store(my, kFlagPrefix | ((excl + mine) & kSizeMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

template <int ES>
__global__ __launch_bounds__(kThreads) void zplane_encode(const uint8_t* __restrict__ in, long long n_elems, long long n_blocks,
                                                          uint64_t* __restrict__ directory, uint8_t* __restrict__ payload,
                                                          unsigned long long* __restrict__ sync_words,  // [0] ticket, [1] fail, [2 ...] states
                                                          Header* __restrict__ header, uint64_t* __restrict__ out_bytes) {
    constexpr int BPW = Geo<ES>::BPW, Q = Geo<ES>::Q;
    __shared__ __attribute__((aligned(16))) uint8_t planes[Q * kBlock];
    __shared__ __attribute__((aligned(8))) uint16_t masks[Q][kThreads];
    __shared__ int span_cnt[Q][kSpans];
    __shared__ int cnt[kSpans][Q][4];
    __shared__ int s_ng[Q], s_mode[Q];
    __shared__ unsigned long long s_size[Q], s_excl, s_wg;
    const int span = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) s_wg = atomicAdd(sync_words, 1ull);
    __syncthreads();
    const long long wg = (long long)s_wg, blk0 = wg * BPW, n_wg = (n_blocks + BPW - 1) / BPW;
    const size_t total = (size_t)n_elems * ES;
    if constexpr (ES > 1) {
        load_planes<ES>(in, (size_t)blk0 * kBlock * ES, total, planes);
        __syncthreads();
    }
    uint4 v[Q];
    uint32_t before[Q], m[Q], mx[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        before[q] = 0;
        if constexpr (ES == 1) {
            const size_t byte0 = (size_t)(blk0 + q) * kBlock, avail = total > byte0 ? total - byte0 : 0;
            v[q] = load_cell_bytes(in + byte0, avail, span);
            if (span > 0 && (size_t)span * kSpan - 1 < avail) before[q] = in[byte0 + (size_t)span * kSpan - 1];
        } else {
            v[q] = *reinterpret_cast<const uint4*>(planes + q * kBlock + span * kSpan + lane * 16);
            if (span > 0) before[q] = planes[q * kBlock + span * kSpan - 1];
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {  // the statistics of zplane_measure
        const uint4 x = xor_filter(v[q], before[q]);
        m[q] = nz16(v[q]);
        mx[q] = nz16(x);
        const int c = wave_total(__popc(m[q])), cx = wave_total(__popc(mx[q]));
        const uint64_t any = __ballot(m[q] != 0), anyx = __ballot(mx[q] != 0);
        const uint64_t g = any | (any >> 1) | (any >> 2) | (any >> 3), gx = anyx | (anyx >> 1) | (anyx >> 2) | (anyx >> 3);
        if (lane == 0) {
            cnt[span][q][0] = c;
            cnt[span][q][1] = __popcll(g & 0x1111111111111111ull);
            cnt[span][q][2] = cx;
            cnt[span][q][3] = __popcll(gx & 0x1111111111111111ull);
        }
    }
    __syncthreads();  // plane statistics complete; every cell is in registers (planes[] is free from here on)
    if (threadIdx.x < Q) {
        const int q = threadIdx.x;
        const long long blk = blk0 + q / ES;
        uint64_t mode = 0, size = 0;
        if (blk < n_blocks) {
            const long long in_block = min((long long)kBlock, n_elems - blk * kBlock);
            int nz = 0, ng = 0, nzx = 0, ngx = 0;
#pragma unroll
            for (int s2 = 0; s2 < kSpans; ++s2) {
                nz += cnt[s2][q][0]; ng += cnt[s2][q][1]; nzx += cnt[s2][q][2]; ngx += cnt[s2][q][3];
            }
            const size_t masked = 8 + 8 * (size_t)ng + pad8((size_t)nz), xored = 8 + 8 * (size_t)ngx + pad8((size_t)nzx),
                         raw = pad8((size_t)in_block);
            if (nz == 0) { mode = 0; size = 0; }
            else if (masked <= xored && masked < raw) { mode = 1; size = masked; }
            else if (xored < raw) { mode = 3; size = xored; }
            else { mode = 2; size = raw; }
        }
        s_mode[q] = (int)mode;
        s_size[q] = size;
    }
    __syncthreads();
    int mode[Q];
    uint64_t local_off[Q], mine = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        mode[q] = s_mode[q];
        local_off[q] = mine;
        mine += s_size[q];
    }
    int incl[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {  // masks + per-span counts of the chosen form
        incl[q] = 0;
        if (mode[q] != 1 && mode[q] != 3) continue;
        if (mode[q] == 3) {
            v[q] = xor_filter(v[q], before[q]);
            m[q] = mx[q];
        }
        masks[q][threadIdx.x] = (uint16_t)m[q];
        incl[q] = wave_incl_sum(__popc(m[q]));
        if (lane == 63) span_cnt[q][span] = incl[q];
    }
    __syncthreads();
    if (span == kSpans - 1) {  // this wave looks back while the others already compact; it compacts its own cells afterwards
        const uint64_t excl = lookback_exclusive(sync_words + 2, wg, mine, sync_words + 1);
        if (lane == 0) s_excl = excl;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (mode[q] != 1 && mode[q] != 3) continue;
        int base = 0;
#pragma unroll
        for (int s2 = 0; s2 < kSpans; ++s2)
            if (s2 < span) base += span_cnt[q][s2];
        compact16(v[q], m[q], planes + q * kBlock, base + incl[q] - __popc(m[q]));
        if (span == (q & (kSpans - 1))) {
            const int nz = span_cnt[q][0] + span_cnt[q][1] + span_cnt[q][2] + span_cnt[q][3];
            if (lane < (int)(pad8((size_t)nz) - nz)) planes[q * kBlock + nz + lane] = 0;
            const uint64_t top = __ballot(reinterpret_cast<const uint64_t*>(&masks[q][0])[lane] != 0);
            if (lane == 0) s_ng[q] = __popcll(top);
        }
    }
    __syncthreads();  // the staged bytes are complete and the payload offset is known
    const uint64_t excl = s_excl;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const long long blk = blk0 + q / ES;
        if (blk >= n_blocks) continue;
        uint8_t* dst = payload + excl + local_off[q];
        if (threadIdx.x == q) directory[blk * ES + q % ES] = ((uint64_t)mode[q] << 62) | (excl + local_off[q]);
        if (mode[q] == 0) continue;
        if (mode[q] == 2) {
            const int n8_raw = (int)pad8((size_t)min((long long)kBlock, n_elems - blk * kBlock));
            const int b0 = span * kSpan + lane * 16;  // the raw plane: every lane's 16 bytes of its cell, straight from the registers
            if (b0 + 8 <= n8_raw) *reinterpret_cast<uint2*>(dst + b0) = uint2{v[q].x, v[q].y};
            if (b0 + 16 <= n8_raw) *reinterpret_cast<uint2*>(dst + b0 + 8) = uint2{v[q].z, v[q].w};
            continue;
        }
        if (span == (q & (kSpans - 1))) {  // [top][group masks]
            const uint64_t gm = reinterpret_cast<const uint64_t*>(&masks[q][0])[lane];
            const uint64_t top = __ballot(gm != 0);
            uint64_t* d64 = reinterpret_cast<uint64_t*>(dst);
            if (gm != 0) d64[1 + __popcll(top & lanes_below(lane))] = gm;
            if (lane == 0) d64[0] = top;
        }
        const int nz = span_cnt[q][0] + span_cnt[q][1] + span_cnt[q][2] + span_cnt[q][3];
        uint8_t* bytes = dst + 8 + 8 * (size_t)s_ng[q];
        for (int i = threadIdx.x * 8; i < (int)pad8((size_t)nz); i += kThreads * 8)
            *reinterpret_cast<uint2*>(bytes + i) = *reinterpret_cast<const uint2*>(planes + q * kBlock + i);
    }
    if (wg == n_wg - 1 && threadIdx.x == 0) {  // the last block in stream order closes the stream
        const uint64_t payload_bytes = excl + mine;
        const bool failed = __hip_atomic_load(sync_words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        header->magic = kMagic;
        header->version = 1;
        header->elem_size = (uint8_t)ES;
        header->block_log2 = kBlockLog2;
        header->n_elems = (uint64_t)n_elems;
        header->payload_bytes = payload_bytes;
        *out_bytes = failed ? 0ull : kHeaderBytes + 8ull * (uint64_t)n_blocks * ES + payload_bytes;
    }
}

// unpack: the headers (one wave per plane: group masks expanded to LDS, per-span byte counts), then the workgroup copies every
// plane's byte area (or raw plane) into LDS in whole words, then the cells pull their bytes out of that staging area into
// registers, undo the XOR filter with one scan per cell, and write the planes (byte streams: straight to global memory).
template <int ES>
__global__ __launch_bounds__(kThreads) void zplane_unpack(const uint8_t* __restrict__ in, size_t in_bytes, long long n_elems,
                                                          uint8_t* __restrict__ out, int* __restrict__ status) {
    constexpr int BPW = Geo<ES>::BPW, Q = Geo<ES>::Q;
    __shared__ __attribute__((aligned(16))) uint8_t planes[Q * kBlock];
    __shared__ uint64_t gmask[Q][kGroups];
    __shared__ int span_cnt[Q][kSpans];
    __shared__ uint32_t span_xor[Q][kSpans];
    __shared__ int s_mode[Q];                  // 0 / 1 / 2 / 3, or 0 with the status word set when the entry does not fit the stream
    __shared__ unsigned long long s_src64[Q];  // offset of the plane's byte area (mode 1 / 3) or raw plane (mode 2) in the stream
    __shared__ int s_copy[Q];                  // bytes to stage (multiple of 8)
    const long long blk0 = (long long)blockIdx.x * BPW;
    const long long n_blocks = (n_elems + kBlock - 1) / kBlock;
    const uint64_t* directory = reinterpret_cast<const uint64_t*>(in + kHeaderBytes);
    const size_t payload0 = kHeaderBytes + 8 * (size_t)n_blocks * ES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = wave; q < Q; q += kThreads / 64) {
        const long long blk = blk0 + q / ES;
        int mode = 0, copy = 0;
        size_t off = 0;
        if (blk < n_blocks) {
            const uint64_t e = directory[blk * ES + q % ES];
            mode = (int)(e >> 62);
            off = payload0 + (size_t)(e & kSizeMask);
        }
        if (mode == 2) {
            copy = (int)pad8((size_t)min((long long)kBlock, n_elems - blk * kBlock));
            if (off + copy > in_bytes) { if (lane == 0) *status = 2; mode = 0; }
        } else if (mode != 0) {
            uint64_t top = 0;
            if (off + 8 > in_bytes) { if (lane == 0) *status = 3; mode = 0; }
            else top = *reinterpret_cast<const uint64_t*>(in + off);
            const int ng = __popcll(top);
            if (mode != 0 && off + 8 + 8 * (size_t)ng > in_bytes) { if (lane == 0) *status = 4; mode = 0; top = 0; }
            // lane g owns group g's mask
            const bool has = (top >> lane) & 1;
            const uint64_t gm = has ? reinterpret_cast<const uint64_t*>(in + off + 8)[__popcll(top & lanes_below(lane))] : 0ull;
            gmask[q][lane] = gm;
            int c = __popcll(gm);  // bytes of this group; summed over the 16 groups of each span
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) c += __shfl_xor(c, d, 64);
            if ((lane & 15) == 0) span_cnt[q][lane >> 4] = c;
            const int total = __shfl(c, 0, 64) + __shfl(c, 16, 64) + __shfl(c, 32, 64) + __shfl(c, 48, 64);
            off += 8 + 8 * (size_t)ng;
            copy = (int)pad8((size_t)total);
            if (mode != 0 && off + (size_t)copy > in_bytes) { if (lane == 0) *status = 5; mode = 0; }  // byte area incl. its pad
        }
        if (lane == 0) {
            s_mode[q] = mode;
            s_src64[q] = off;
            s_copy[q] = mode == 0 ? 0 : copy;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const uint8_t* src = in + s_src64[q];
        for (int i = threadIdx.x * 8; i < s_copy[q]; i += kThreads * 8)
            *reinterpret_cast<uint2*>(planes + q * kBlock + i) = *reinterpret_cast<const uint2*>(src + i);
    }
    __syncthreads();
    const int span = wave;
    uint4 v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int mode = s_mode[q];
        v[q] = uint4{0, 0, 0, 0};
        if (mode == 0) continue;
        if (mode == 2) {
            v[q] = *reinterpret_cast<const uint4*>(planes + q * kBlock + span * kSpan + lane * 16);
            continue;
        }
        const uint32_t m = reinterpret_cast<const uint16_t*>(&gmask[q][0])[threadIdx.x];
        const int incl = wave_incl_sum(__popc(m));
        int base = 0;
#pragma unroll
        for (int s2 = 0; s2 < kSpans; ++s2)
            if (s2 < span) base += span_cnt[q][s2];
        expand16(v[q], m, planes + q * kBlock, base + incl - __popc(m));
        if (mode == 3) {  // undo b[i] ^ b[i-1]: prefix XOR inside the lane, one scan over the lanes, the spans before via LDS
            const uint32_t tot = xor_prefix16(v[q]);
            const uint32_t incl_x = wave_incl_xor(tot);
            const uint32_t k = (incl_x ^ tot) * 0x01010101u;
            v[q].x ^= k; v[q].y ^= k; v[q].z ^= k; v[q].w ^= k;
            if (lane == 63) span_xor[q][span] = incl_x;
        }
    }
    __syncthreads();  // every cell is in registers: the staging area may be overwritten by the planes; span_xor is complete
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (s_mode[q] == 3) {
            uint32_t carry = 0;
#pragma unroll
            for (int s2 = 0; s2 < kSpans; ++s2)
                if (s2 < span) carry ^= span_xor[q][s2];
            const uint32_t k = (carry & 0xffu) * 0x01010101u;
            v[q].x ^= k; v[q].y ^= k; v[q].z ^= k; v[q].w ^= k;
        }
        if constexpr (ES == 1) {
            const long long blk = blk0 + q;
            if (blk < n_blocks) {
                const size_t valid = (size_t)min((long long)kBlock, n_elems - blk * kBlock);
                uint8_t* dst = out + (size_t)blk * kBlock;
                const size_t b0 = (size_t)span * kSpan + lane * 16;
                if (b0 + 16 <= valid) {
                    *reinterpret_cast<uint4*>(dst + b0) = v[q];
                } else {
                    union { uint4 w; uint8_t b[16]; } u;
                    u.w = v[q];
                    for (int k = 0; k < 16 && b0 + k < valid; ++k) dst[b0 + k] = u.b[k];
                }
            }
        } else {
            *reinterpret_cast<uint4*>(planes + q * kBlock + span * kSpan + lane * 16) = v[q];
        }
    }
    if constexpr (ES > 1) {
        __syncthreads();
        const size_t valid = (size_t)min((long long)kBlock, n_elems - blk0 * kBlock) * ES;
        store_planes<ES>(planes, out + (size_t)blk0 * kBlock * ES, valid);  // interleave the planes back into elements (coalesced 16-byte stores)
    }
}

inline long long blocks_of(int64_t n) { return (n + kBlock - 1) / kBlock; }
inline bool es_ok(int es) { return es == 1 || es == 2 || es == 4 || es == 8; }

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_zplane_bound_bytes(int64_t n_elems, int elem_size) {
    if (n_elems < 0 || !es_ok(elem_size)) return 0;
    const size_t nb = (size_t)blocks_of(n_elems);
    return kHeaderBytes + 8 * nb * elem_size + nb * elem_size * (size_t)kBlock;  // every plane raw
}

extern "C" size_t rlx_zplane_workspace_bytes(int64_t n_elems, int elem_size) {
    if (n_elems < 0 || !es_ok(elem_size)) return 0;
    const size_t n_entries = (size_t)blocks_of(n_elems) * elem_size;
    // multi-launch encoder: entry sizes | scan tile totals; single-pass encoder: ticket | fail | one look-back word per workgroup
    return 8 * n_entries + 8 * ((n_entries + kScanTile - 1) / kScanTile) + 8 * ((size_t)blocks_of(n_elems) + 2) + 64;
}

extern "C" int rlx_zplane_compress(const void* in, int64_t n_elems, int elem_size, void* out, size_t out_capacity, uint64_t* out_bytes,
                                   void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(n_elems >= 0 && es_ok(elem_size), "rlx_zplane_compress: n_elems=%lld elem_size=%d (1, 2, 4 or 8)", (long long)n_elems, elem_size);
    RLX_REQUIRE(out && out_bytes && (in || n_elems == 0), "rlx_zplane_compress: NULL argument");
    RLX_REQUIRE(reinterpret_cast<uintptr_t>(in) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 8 == 0,
                "rlx_zplane_compress: input must be 16-byte aligned, output 8-byte aligned");
    if (out_capacity < rlx_zplane_bound_bytes(n_elems, elem_size) || workspace_bytes < rlx_zplane_workspace_bytes(n_elems, elem_size) ||
        (workspace == nullptr)) {
        set_error("rlx_zplane_compress: output (%zu) or workspace (%zu) smaller than rlx_zplane_bound_bytes / _workspace_bytes", out_capacity,
                  workspace_bytes);
        return RLX_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long nb = blocks_of(n_elems), n_entries = nb * elem_size, n_tiles = (n_entries + kScanTile - 1) / kScanTile;
    uint64_t* sizes = static_cast<uint64_t*>(workspace);
    uint64_t* tile_total = sizes + n_entries;
    uint8_t* o = static_cast<uint8_t*>(out);
    uint64_t* directory = reinterpret_cast<uint64_t*>(o + kHeaderBytes);
    uint8_t* payload = o + kHeaderBytes + 8 * (size_t)n_entries;
    const uint8_t* src = static_cast<const uint8_t*>(in);
    const unsigned grid1 = (unsigned)((nb + kBytesBlocks - 1) / kBytesBlocks);  // byte streams: several blocks per workgroup
#define RLX_ZP_DISPATCH(KERNEL, ...)                                                                              \
    switch (elem_size) {                                                                                          \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(grid1), dim3(kThreads), 0, st, __VA_ARGS__); break;           \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3((unsigned)nb), dim3(kThreads), 0, st, __VA_ARGS__); break;    \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3((unsigned)nb), dim3(kThreads), 0, st, __VA_ARGS__); break;    \
        default: hipLaunchKernelGGL(KERNEL<8>, dim3((unsigned)nb), dim3(kThreads), 0, st, __VA_ARGS__); break;   \
    }
    // development: RLX_ZPLANE_SINGLE_PASS=1 selects the one-launch encoder (same bytes; measured SLOWER, see its comment); read per
    // call so that the tests can switch it
#ifdef RLX_DEV_VARIANTS
    const char* sp_env = getenv("RLX_ZPLANE_SINGLE_PASS");
    const bool single_pass = sp_env != nullptr && atoi(sp_env) != 0;
    if (single_pass && nb > 0) {
        unsigned long long* sync_words = reinterpret_cast<unsigned long long*>(tile_total + n_tiles);
        const long long n_wg = elem_size == 1 ? (long long)grid1 : nb;
        RLX_HIP_CHECK(hipMemsetAsync(sync_words, 0, 8 * (size_t)(n_wg + 2), st));
        RLX_ZP_DISPATCH(zplane_encode, src, (long long)n_elems, nb, directory, payload, sync_words, reinterpret_cast<Header*>(o), out_bytes);
        RLX_LAUNCH_CHECK();
        return RLX_OK;
    }
#endif
    if (nb > 0) {
        {  // measure can walk its pieces grid-stride with at most `cap` workgroups (development: RLX_ZPLANE_MEASURE_GRID; 0 = one
           // workgroup per piece).  Measured (profiles/r03_zplane_codec_v4_kernels.txt): byte streams 52.6 -> 48.4 us with 2048,
           // 4-byte elements 57.1 -> 59.0 us -- the launch is NOT bound by the workgroup start rate; byte streams keep the walk.
#ifdef RLX_DEV_VARIANTS
            const char* ge = getenv("RLX_ZPLANE_MEASURE_GRID");
#else
            const char* ge = nullptr;
#endif
            const long long cap = ge != nullptr ? atoll(ge) : (elem_size == 1 ? 2048 : 0), pieces = elem_size == 1 ? (long long)grid1 : nb;
            const unsigned gm = (unsigned)(cap > 0 ? std::min(cap, pieces) : pieces);
            switch (elem_size) {
                case 1: hipLaunchKernelGGL(zplane_measure<1>, dim3(gm), dim3(kThreads), 0, st, src, (long long)n_elems, nb, sizes); break;
                case 2: hipLaunchKernelGGL(zplane_measure<2>, dim3(gm), dim3(kThreads), 0, st, src, (long long)n_elems, nb, sizes); break;
                case 4: hipLaunchKernelGGL(zplane_measure<4>, dim3(gm), dim3(kThreads), 0, st, src, (long long)n_elems, nb, sizes); break;
                default: hipLaunchKernelGGL(zplane_measure<8>, dim3(gm), dim3(kThreads), 0, st, src, (long long)n_elems, nb, sizes); break;
            }
        }
        RLX_LAUNCH_CHECK();
        hipLaunchKernelGGL(zplane_scan_tiles, dim3((unsigned)n_tiles), dim3(1024), 0, st, sizes, n_entries, directory, tile_total);
        RLX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(zplane_scan_bases, dim3(1), dim3(1024), 0, st, tile_total, n_tiles, n_entries, reinterpret_cast<Header*>(o),
                       (long long)n_elems, elem_size, out_bytes);
    RLX_LAUNCH_CHECK();
    if (nb > 0) {
        RLX_ZP_DISPATCH(zplane_pack, src, (long long)n_elems, nb, directory, (const uint64_t*)tile_total, payload);
        RLX_LAUNCH_CHECK();
    }
    return RLX_OK;
}

extern "C" int rlx_zplane_parse_header(const void* header_host, int64_t* n_elems, int* elem_size, uint64_t* total_bytes) {
    RLX_REQUIRE(header_host && n_elems && elem_size && total_bytes, "rlx_zplane_parse_header: NULL argument");
    Header h;
    memcpy(&h, header_host, sizeof(h));
    RLX_REQUIRE(h.magic == kMagic && h.version == 1 && h.block_log2 == kBlockLog2 && es_ok(h.elem_size),
                "rlx_zplane_parse_header: not an RLXZ v1 stream (magic %08x version %d elem_size %d)", h.magic, h.version, h.elem_size);
    *n_elems = (int64_t)h.n_elems;
    *elem_size = h.elem_size;
    *total_bytes = kHeaderBytes + 8ull * (uint64_t)blocks_of((int64_t)h.n_elems) * h.elem_size + h.payload_bytes;
    return RLX_OK;
}

extern "C" int rlx_zplane_decompress(const void* in, size_t in_bytes, void* out, int64_t n_elems, int elem_size, int* status,
                                     rlx_stream_t stream) {
    RLX_REQUIRE(n_elems >= 0 && es_ok(elem_size), "rlx_zplane_decompress: n_elems=%lld elem_size=%d", (long long)n_elems, elem_size);
    RLX_REQUIRE(in && status && (out || n_elems == 0), "rlx_zplane_decompress: NULL argument");
    RLX_REQUIRE(reinterpret_cast<uintptr_t>(in) % 8 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0,
                "rlx_zplane_decompress: input must be 8-byte aligned, output 16-byte aligned");
    const long long nb = blocks_of(n_elems);
    RLX_REQUIRE(in_bytes >= kHeaderBytes + 8 * (size_t)nb * elem_size, "rlx_zplane_decompress: stream shorter than its directory");
    if (nb == 0) return RLX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint8_t* src = static_cast<const uint8_t*>(in);
    uint8_t* dst = static_cast<uint8_t*>(out);
    const unsigned grid1 = (unsigned)((nb + kBytesBlocks - 1) / kBytesBlocks);
    RLX_ZP_DISPATCH(zplane_unpack, src, in_bytes, (long long)n_elems, dst, status);
    RLX_LAUNCH_CHECK();
#undef RLX_ZP_DISPATCH
    return RLX_OK;
}
