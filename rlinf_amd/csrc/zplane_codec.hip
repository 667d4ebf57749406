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
// block: it loads the block's elements with coalesced 16-byte loads, splits them into byte planes in LDS, and every wave walks
// planes with one ballot per 64-byte group.  HBM-bound: the input is read twice (measure, pack), the output written once.
// Integer / byte work: bit-exact round trip (tests/test_gpu_weight_patch.py, against a numpy restatement of the format).

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

// byte i of the plane under the XOR filter (b[i] ^ b[i-1], b[-1] = 0): the predecessor of a group's first byte is the previous
// group's last byte
__device__ __forceinline__ uint8_t xor_prev(const uint8_t* plane, int i) { return plane[i] ^ (i > 0 ? plane[i - 1] : (uint8_t)0); }

// 4-bit "byte is nonzero" mask of a dword (bit j = byte j): the classic has-zero-byte carry trick, then a multiply that gathers
// bits 0, 8, 16, 24 into one nibble (all partial products land on distinct bit positions: no carries)
__device__ __forceinline__ uint32_t nz4(uint32_t w) {
    const uint32_t t = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
    return ((t >> 7) * 0x01020408u) >> 24;
}
__device__ __forceinline__ uint32_t nz16(const uint4& v) { return nz4(v.x) | (nz4(v.y) << 4) | (nz4(v.z) << 8) | (nz4(v.w) << 12); }

// mask bookkeeping of one plane-block, one wave: nz = nonzero bytes, ng = nonzero 64-byte groups, plain and under the XOR filter.
// A lane reads 16 consecutive bytes (one b128) of a 1024-byte span -- four lanes make a group -- and works on them in registers;
// the byte in front of a lane's first one comes from its neighbour (or the previous span's last lane).
__device__ __forceinline__ void plane_stats(const uint8_t* plane, int& nz, int& ng, int& nzx, int& ngx) {
    const int lane = threadIdx.x & 63;
    int cnt = 0, cntx = 0;
    ng = ngx = 0;
    uint32_t carry = 0;  // last byte of the previous span (in the low byte)
#pragma unroll
    for (int sp = 0; sp < kBlock / 1024; ++sp) {
        const uint4 v = *reinterpret_cast<const uint4*>(plane + sp * 1024 + lane * 16);
        uint32_t prev = (uint32_t)__shfl_up((int)(v.w >> 24), 1, 64);
        if (lane == 0) prev = carry;
        carry = (uint32_t)__shfl((int)(v.w >> 24), 63, 64);
        uint4 x;  // v XOR (v shifted up by one byte, with the predecessor byte coming in at the bottom)
        x.x = v.x ^ ((v.x << 8) | prev);
        x.y = v.y ^ ((v.y << 8) | (v.x >> 24));
        x.z = v.z ^ ((v.z << 8) | (v.y >> 24));
        x.w = v.w ^ ((v.w << 8) | (v.z >> 24));
        const uint32_t m = nz16(v), mx = nz16(x);
        cnt += __popc(m);
        cntx += __popc(mx);
        // a group is nonzero iff any of its four lanes holds a nonzero byte: OR over the quad, counted once per quad
        const uint64_t any = __ballot(m != 0), anyx = __ballot(mx != 0);
        uint64_t q = any | (any >> 1) | (any >> 2) | (any >> 3), qx = anyx | (anyx >> 1) | (anyx >> 2) | (anyx >> 3);
        ng += __popcll(q & 0x1111111111111111ull);
        ngx += __popcll(qx & 0x1111111111111111ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_xor(cnt, off, 64);
        cntx += __shfl_xor(cntx, off, 64);
    }
    nz = cnt;
    nzx = cntx;
}

template <int ES>
__global__ __launch_bounds__(kThreads) void zplane_measure(const uint8_t* __restrict__ in, long long n_elems, uint64_t* __restrict__ sizes) {
    __shared__ __attribute__((aligned(16))) uint8_t planes[ES * kBlock];
    const long long blk = blockIdx.x;
    load_planes<ES>(in, (size_t)blk * kBlock * ES, (size_t)n_elems * ES, planes);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long in_block = min((long long)kBlock, n_elems - blk * kBlock);
    for (int p = wave; p < ES; p += kThreads / 64) {
        int nz, ng, nzx, ngx;
        plane_stats(planes + p * kBlock, nz, ng, nzx, ngx);
        if (lane == 0) {
            const size_t masked = 8 + 8 * (size_t)ng + pad8((size_t)nz), xored = 8 + 8 * (size_t)ngx + pad8((size_t)nzx),
                         raw = pad8((size_t)in_block);
            uint64_t mode, size;
            if (nz == 0) { mode = 0; size = 0; }
            else if (masked <= xored && masked < raw) { mode = 1; size = masked; }
            else if (xored < raw) { mode = 3; size = xored; }
            else { mode = 2; size = raw; }
            sizes[blk * ES + p] = (mode << 62) | size;
        }
    }
}

// exclusive scan of the entry sizes -> directory entries (mode | offset) + the header; ONE workgroup of 1024 threads
__global__ __launch_bounds__(1024) void zplane_scan(const uint64_t* __restrict__ sizes, long long n_entries, uint64_t* __restrict__ directory,
                                                    Header* __restrict__ header, long long n_elems, int elem_size,
                                                    uint64_t* __restrict__ out_bytes) {
    __shared__ uint64_t s_part[1024];
    const long long chunk = (n_entries + 1023) / 1024;
    const long long lo = min(n_entries, (long long)threadIdx.x * chunk), hi = min(n_entries, lo + chunk);
    uint64_t sum = 0;
    for (long long i = lo; i < hi; ++i) sum += sizes[i] & ((1ull << 62) - 1);
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan over the 1024 partial sums
        const uint64_t v = threadIdx.x >= (unsigned)off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    uint64_t run = threadIdx.x == 0 ? 0 : s_part[threadIdx.x - 1];
    for (long long i = lo; i < hi; ++i) {
        const uint64_t e = sizes[i];
        directory[i] = (e & (3ull << 62)) | run;
        run += e & ((1ull << 62) - 1);
    }
    if (threadIdx.x == 1023) {
        const uint64_t payload = s_part[1023];
        header->magic = kMagic;
        header->version = 1;
        header->elem_size = (uint8_t)elem_size;
        header->block_log2 = kBlockLog2;
        header->n_elems = (uint64_t)n_elems;
        header->payload_bytes = payload;
        *out_bytes = kHeaderBytes + 8ull * (uint64_t)n_entries + payload;
    }
}

template <int ES>
__global__ __launch_bounds__(kThreads) void zplane_pack(const uint8_t* __restrict__ in, long long n_elems, const uint64_t* __restrict__ directory,
                                                        uint8_t* __restrict__ payload) {
    __shared__ __attribute__((aligned(16))) uint8_t planes[ES * kBlock];
    const long long blk = blockIdx.x;
    load_planes<ES>(in, (size_t)blk * kBlock * ES, (size_t)n_elems * ES, planes);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long in_block = min((long long)kBlock, n_elems - blk * kBlock);
    for (int p = wave; p < ES; p += kThreads / 64) {
        const uint64_t e = directory[blk * ES + p];
        const int mode = (int)(e >> 62);
        uint8_t* dst = payload + (e & ((1ull << 62) - 1));
        const uint8_t* plane = planes + p * kBlock;
        if (mode == 0) continue;
        if (mode == 2) {
            const int n8 = (int)pad8((size_t)in_block);  // the pad bytes were loaded as zero
            for (int i = lane * 8; i < n8; i += 64 * 8) *reinterpret_cast<uint2*>(dst + i) = *reinterpret_cast<const uint2*>(plane + i);
            continue;
        }
        // mode 1 / 3: [top][group masks][nonzero bytes] of the plane (3: under the XOR filter).  First the masks (so that the byte
        // area's start is known), then the bytes.
        const bool filt = mode == 3;
        uint64_t top = 0;
        int ng = 0;
        for (int g = 0; g < kGroups; ++g) {
            const uint8_t b = filt ? xor_prev(plane, g * 64 + lane) : plane[g * 64 + lane];
            const uint64_t m = __ballot(b != 0);
            if (m != 0) {
                top |= 1ull << g;
                if (lane == 0) reinterpret_cast<uint64_t*>(dst)[1 + ng] = m;
                ++ng;
            }
        }
        if (lane == 0) reinterpret_cast<uint64_t*>(dst)[0] = top;
        uint8_t* bytes = dst + 8 + 8 * (size_t)ng;
        int run = 0;
        for (int g = 0; g < kGroups; ++g) {
            const uint8_t b = filt ? xor_prev(plane, g * 64 + lane) : plane[g * 64 + lane];
            const uint64_t m = __ballot(b != 0);
            if (b != 0) bytes[run + __popcll(m & lanes_below(lane))] = b;
            run += __popcll(m);
        }
        if (lane < (int)(pad8((size_t)run) - run)) bytes[run + lane] = 0;  // deterministic pad bytes
    }
}

template <int ES>
__global__ __launch_bounds__(kThreads) void zplane_unpack(const uint8_t* __restrict__ in, size_t in_bytes, long long n_elems,
                                                          uint8_t* __restrict__ out, int* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t planes[ES * kBlock];
    const long long blk = blockIdx.x;
    const long long n_blocks = (n_elems + kBlock - 1) / kBlock;
    const uint64_t* directory = reinterpret_cast<const uint64_t*>(in + kHeaderBytes);
    const size_t payload0 = kHeaderBytes + 8 * (size_t)n_blocks * ES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long in_block = min((long long)kBlock, n_elems - blk * kBlock);
    for (int p = wave; p < ES; p += kThreads / 64) {
        const uint64_t e = directory[blk * ES + p];
        const int mode = (int)(e >> 62);
        const size_t off = payload0 + (size_t)(e & ((1ull << 62) - 1));
        uint8_t* plane = planes + p * kBlock;
        if (mode == 0) {
            for (int i = lane * 8; i < kBlock; i += 64 * 8) *reinterpret_cast<uint2*>(plane + i) = uint2{0, 0};
            continue;
        }
        if (mode == 2) {
            const int n8 = (int)pad8((size_t)in_block);
            if (off + n8 > in_bytes) { if (lane == 0) *status = 2; continue; }
            for (int i = lane * 8; i < n8; i += 64 * 8) *reinterpret_cast<uint2*>(plane + i) = *reinterpret_cast<const uint2*>(in + off + i);
            continue;
        }
        if (off + 8 > in_bytes) { if (lane == 0) *status = 3; continue; }
        const uint64_t top = *reinterpret_cast<const uint64_t*>(in + off);
        const int ng = __popcll(top);
        if (off + 8 + 8 * (size_t)ng > in_bytes) { if (lane == 0) *status = 4; continue; }
        // lane g owns group g's mask and the offset of its bytes (wave-wide exclusive scan of the popcounts)
        const bool has = (top >> lane) & 1;
        const uint64_t gm = has ? reinterpret_cast<const uint64_t*>(in + off + 8)[__popcll(top & lanes_below(lane))] : 0ull;
        int incl = __popcll(gm);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const int excl = incl - __popcll(gm);
        const int total = __shfl(incl, 63, 64);
        const uint8_t* bytes = in + off + 8 + 8 * (size_t)ng;
        if (off + 8 + 8 * (size_t)ng + (size_t)total > in_bytes) { if (lane == 0) *status = 5; continue; }
        unsigned carry = 0;  // mode 3: the running XOR up to the previous group's last byte
        for (int g = 0; g < kGroups; ++g) {
            const uint64_t mg = __shfl(gm, g, 64);
            const int og = __shfl(excl, g, 64);
            unsigned b = ((mg >> lane) & 1) ? bytes[og + __popcll(mg & lanes_below(lane))] : 0u;
            if (mode == 3) {  // undo b[i] ^ b[i-1]: inclusive XOR scan over the lanes, then the carry of the groups before
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned v = __shfl_up(b, d, 64);
                    if (lane >= d) b ^= v;
                }
                b ^= carry;
                carry = __shfl(b, 63, 64);
            }
            plane[g * 64 + lane] = (uint8_t)b;
        }
    }
    __syncthreads();
    // interleave the planes back into elements (coalesced 16-byte stores)
    store_planes<ES>(planes, out + (size_t)blk * kBlock * ES, (size_t)in_block * ES);
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
    return 8 * (size_t)blocks_of(n_elems) * elem_size + 64;  // entry sizes
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
    const long long nb = blocks_of(n_elems), n_entries = nb * elem_size;
    uint64_t* sizes = static_cast<uint64_t*>(workspace);
    uint8_t* o = static_cast<uint8_t*>(out);
    uint64_t* directory = reinterpret_cast<uint64_t*>(o + kHeaderBytes);
    uint8_t* payload = o + kHeaderBytes + 8 * (size_t)n_entries;
    const uint8_t* src = static_cast<const uint8_t*>(in);
#define RLX_ZP_DISPATCH(KERNEL, ...)                                                                              \
    switch (elem_size) {                                                                                          \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3((unsigned)nb), dim3(kThreads), 0, st, __VA_ARGS__); break;    \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3((unsigned)nb), dim3(kThreads), 0, st, __VA_ARGS__); break;    \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3((unsigned)nb), dim3(kThreads), 0, st, __VA_ARGS__); break;    \
        default: hipLaunchKernelGGL(KERNEL<8>, dim3((unsigned)nb), dim3(kThreads), 0, st, __VA_ARGS__); break;   \
    }
    if (nb > 0) {
        RLX_ZP_DISPATCH(zplane_measure, src, (long long)n_elems, sizes);
        RLX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(zplane_scan, dim3(1), dim3(1024), 0, st, sizes, n_entries, directory, reinterpret_cast<Header*>(o), (long long)n_elems,
                       elem_size, out_bytes);
    RLX_LAUNCH_CHECK();
    if (nb > 0) {
        RLX_ZP_DISPATCH(zplane_pack, src, (long long)n_elems, directory, payload);
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
    RLX_ZP_DISPATCH(zplane_unpack, src, in_bytes, (long long)n_elems, dst, status);
    RLX_LAUNCH_CHECK();
#undef RLX_ZP_DISPATCH
    return RLX_OK;
}
