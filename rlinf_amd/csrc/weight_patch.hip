// weight_patch.hip -- the sparse weight-patch wire format of the actor -> rollout weight sync, gfx950.
//
// Replaces the tensor-op chain of GPUSnapshotPatchBuilder.create_patch (rlinf/hybrid_engines/weight_syncer/
// patch_syncer.py:648-774: to(dtype) -> ne -> nonzero -> gather -> scatter into the snapshot -> delta_encode :290-327)
// and of the receiver (PatchWeightSyncer.apply :1040-1137: delta_decode :329-370 -> index_put), integer / byte work:
//
//   patch_scan   ONE read of the new tensor and the snapshot (the HBM stream): a changed-bit per element (n/8 bytes),
//                per-16384-element block the changed count and the last changed index
//   patch_offsets one block: exclusive sum of the counts, exclusive running max of the last-changed indices, total
//   patch_emit   reads the bit mask (1/16 of a bf16 tensor) and touches only changed elements: COO (row, col) absolute
//                or delta-encoded exactly like PatchBuilder.delta_encode, value bytes in the snapshot's dtype, snapshot
//                updated in place; also the maxima the host needs to pick the index dtype (downscale_nonnegative_indices)
//   patch_apply  decode (sum-scan of row deltas, segmented sum-scan of column deltas) + scatter of the value bytes
//
// Order is nonzero()'s: ascending linear index.  Float comparison is torch.ne's: NaN != NaN (a NaN is resent every
// time), +0 == -0.  Everything is deterministic and bit-exact against the reference classes.

#include "rlx_common.h"
#include "rlx_convert.h"

namespace rlx {
namespace {

constexpr int PT = 256;                 // threads per block
constexpr int WORD = 64;                // elements per mask word (one thread of patch_emit)
constexpr int BLOCK_ELEMS = PT * WORD;  // 16384
constexpr int BLOCK_BYTES = BLOCK_ELEMS / 8;

// element access (Conv<SRC, DST>, Pack, RawPack): rlx_convert.h -- SRC = dtype of the sender's tensor, DST = dtype of
// the snapshot / the wire
template <typename T>
__device__ __forceinline__ bool differs(T a, T b) { return a != b; }  // floats: IEEE (NaN != NaN, +0 == -0); ints: bits
template <>
__device__ __forceinline__ bool differs<__bf16>(__bf16 a, __bf16 b) { return (float)a != (float)b; }
template <>
__device__ __forceinline__ bool differs<_Float16>(_Float16 a, _Float16 b) { return (float)a != (float)b; }

// ---- block scans --------------------------------------------------------------------------------------------------
// exclusive sum and exclusive running max over the PT threads of a block, in thread order; totals returned too
struct ScanOut {
    long long ex_sum, total_sum, ex_max, total_max;
};
template <int NT = PT>
__device__ __forceinline__ ScanOut block_scan_sum_max(long long s, long long m, long long* lds /* 2 * NT/64 */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    long long is = s, im = m;  // inclusive within the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long os = __shfl_up(is, off, 64), om = __shfl_up(im, off, 64);
        if (lane >= off) {
            is += os;
            im = om > im ? om : im;
        }
    }
    __syncthreads();
    if (lane == 63) lds[wid] = is, lds[NW + wid] = im;
    __syncthreads();
    long long base_s = 0, base_m = -1, tot_s = 0, tot_m = -1;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const long long ws = lds[w], wm = lds[NW + w];
        if (w < wid) base_s += ws, base_m = wm > base_m ? wm : base_m;
        tot_s += ws, tot_m = wm > tot_m ? wm : tot_m;
    }
    ScanOut o;
    const long long prev_s = __shfl_up(is, 1, 64), prev_m = __shfl_up(im, 1, 64);
    o.ex_sum = base_s + (lane ? prev_s : 0);
    const long long pm = lane ? prev_m : -1;
    o.ex_max = base_m > pm ? base_m : pm;
    o.total_sum = tot_s, o.total_max = tot_m;
    return o;
}

// ---- pass A: compare ------------------------------------------------------------------------------------------------
template <typename SRC, typename DST, bool VEC>
__global__ __launch_bounds__(PT) void patch_scan_kernel(const SRC* __restrict__ value, const DST* __restrict__ snap,
                                                        long long n, uint8_t* __restrict__ mask,
                                                        int* __restrict__ block_count, long long* __restrict__ block_last) {
    __shared__ long long lds[2 * (PT / 64) + 2];
    const long long byte0 = (long long)blockIdx.x * BLOCK_BYTES;
    int cnt = 0;
    long long last = -1;
    constexpr int ITERS = BLOCK_BYTES / PT;
    if (VEC && (byte0 + BLOCK_BYTES) * 8 <= n) {
        // interior block (block-uniform test): every load is issued before the first compare -- a per-lane bounds guard
        // around each load would serialise them behind s_waitcnt vmcnt(0)
        RawPack<sizeof(SRC) * 8> ra[ITERS];
        RawPack<sizeof(DST) * 8> rb[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const long long e0 = (byte0 + it * PT + threadIdx.x) * 8;
            ra[it].load(value + e0);
            rb[it].load(snap + e0);
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const long long k = byte0 + it * PT + threadIdx.x;
            const SRC* a = reinterpret_cast<const SRC*>(ra[it].w);
            const DST* b = reinterpret_cast<const DST*>(rb[it].w);
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) bits |= differs<DST>(Conv<SRC, DST>::cvt(a[i]), b[i]) ? (1u << i) : 0u;
            mask[k] = (uint8_t)bits;
            cnt += __popc(bits);
            if (bits) last = k * 8 + (31 - __clz(bits));
        }
    } else {
        for (int it = 0; it < ITERS; ++it) {
            const long long k = byte0 + it * PT + threadIdx.x;
            const long long e0 = k * 8;
            unsigned bits = 0;
            for (int i = 0; i < 8 && e0 + i < n; ++i)
                bits |= differs<DST>(Conv<SRC, DST>::cvt(value[e0 + i]), snap[e0 + i]) ? (1u << i) : 0u;
            if (e0 < ((n + 7) & ~7ll)) mask[k] = (uint8_t)bits;
            if (bits) {
                cnt += __popc(bits);
                last = e0 + (31 - __clz(bits));  // bytes are visited in ascending order: the latest non-empty byte wins
            }
        }
    }
    const ScanOut s = block_scan_sum_max(cnt, last, lds);
    if (threadIdx.x == 0) block_count[blockIdx.x] = (int)s.total_sum, block_last[blockIdx.x] = s.total_max;
}

// ---- pass B: block offsets, two levels ------------------------------------------------------------------------------
// B1: every group of GROUP scan blocks gets its local exclusive sums / running maxima (coalesced, one workgroup per group);
// B2: one workgroup turns the group totals into group bases.  patch_emit adds the two.
constexpr int GT = 1024;        // threads of a B1 workgroup
constexpr int GI = 4;           // scan blocks per thread
constexpr int GROUP = GT * GI;  // 4096 scan blocks = 64 Mi elements per group
__global__ __launch_bounds__(GT) void patch_offsets_local_kernel(const int* __restrict__ block_count,
                                                                 const long long* __restrict__ block_last, long long nblocks,
                                                                 long long* __restrict__ block_offset,
                                                                 long long* __restrict__ block_prev,
                                                                 long long* __restrict__ group_sum,
                                                                 long long* __restrict__ group_max) {
    __shared__ long long lds[2 * (GT / 64)];
    const long long i0 = (long long)blockIdx.x * GROUP + (long long)threadIdx.x * GI;
    long long c[GI], l[GI], sum = 0, mx = -1;
#pragma unroll
    for (int k = 0; k < GI; ++k) {
        const bool in = i0 + k < nblocks;
        c[k] = in ? block_count[i0 + k] : 0;
        l[k] = in ? block_last[i0 + k] : -1;
        sum += c[k];
        mx = l[k] > mx ? l[k] : mx;
    }
    const ScanOut s = block_scan_sum_max<GT>(sum, mx, lds);
    long long rs = s.ex_sum, rm = s.ex_max;
#pragma unroll
    for (int k = 0; k < GI; ++k) {
        if (i0 + k < nblocks) block_offset[i0 + k] = rs, block_prev[i0 + k] = rm;
        rs += c[k];
        rm = l[k] > rm ? l[k] : rm;
    }
    if (threadIdx.x == 0) group_sum[blockIdx.x] = s.total_sum, group_max[blockIdx.x] = s.total_max;
}
__global__ __launch_bounds__(GT) void patch_offsets_groups_kernel(long long* __restrict__ group_sum,
                                                                  long long* __restrict__ group_max, long long ngroups,
                                                                  long long* nnz_out) {
    __shared__ long long lds[2 * (GT / 64)];
    __shared__ long long s_base[2];
    if (threadIdx.x == 0) s_base[0] = 0, s_base[1] = -1;
    __syncthreads();
    for (long long g0 = 0; g0 < ngroups; g0 += GT) {  // one sweep covers 1024 groups = 64 Gi elements
        const long long g = g0 + threadIdx.x;
        const long long cs = g < ngroups ? group_sum[g] : 0, cm = g < ngroups ? group_max[g] : -1;
        const ScanOut s = block_scan_sum_max<GT>(cs, cm, lds);
        const long long bs = s_base[0], bm = s_base[1];
        if (g < ngroups) group_sum[g] = bs + s.ex_sum, group_max[g] = bm > s.ex_max ? bm : s.ex_max;
        __syncthreads();
        if (threadIdx.x == 0) s_base[0] = bs + s.total_sum, s_base[1] = bm > s.total_max ? bm : s.total_max;
        __syncthreads();
    }
    if (threadIdx.x == 0) nnz_out[0] = s_base[0];
}

// ---- pass C: emit ---------------------------------------------------------------------------------------------------
// EW = mask words per thread (one emit workgroup spans EW scan blocks): 8 for sparse patches, where the block scans dominate,
// 1 for dense ones, where the scattered index / value stores do and more lanes in flight help
template <typename SRC, typename DST, int EW>
__global__ __launch_bounds__(PT) void patch_emit_kernel(const SRC* __restrict__ value, DST* __restrict__ snap, long long n,
                                                        long long cols, int delta, const uint8_t* __restrict__ mask,
                                                        const long long* __restrict__ block_offset,
                                                        const long long* __restrict__ block_prev,
                                                        const long long* __restrict__ group_base,
                                                        const long long* __restrict__ group_prev,
                                                        long long* __restrict__ out_rows, long long* __restrict__ out_cols,
                                                        DST* __restrict__ out_values, long long* __restrict__ block_max_r,
                                                        long long* __restrict__ block_max_c) {
    __shared__ long long lds[2 * (PT / 64) + 2];
    __shared__ unsigned long long s_mx[2 * (PT / 64)];
    // EW consecutive mask words (EW * 64 elements) per thread: one block scan per EW scan blocks instead of one each -- at
    // 0.1 % density the kernel is bound by the scans, not by the mask bytes it reads
    const long long w0 = ((long long)blockIdx.x * PT + threadIdx.x) * EW;
    const long long nbytes = (n + 7) >> 3;
    unsigned long long words[EW];
    if ((w0 + EW) * 8 <= nbytes) {
        if constexpr (EW == 1) {
            words[0] = *reinterpret_cast<const unsigned long long*>(mask + w0 * 8);
        } else {
#pragma unroll
            for (int j = 0; j < EW; j += 2) {
                const ulonglong2 q = *reinterpret_cast<const ulonglong2*>(mask + (w0 + j) * 8);
                words[j] = q.x, words[j + 1] = q.y;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < EW; ++j) {
            words[j] = 0;
            for (long long bq = (w0 + j) * 8; bq < nbytes && bq < (w0 + j + 1) * 8; ++bq)
                words[j] |= (unsigned long long)mask[bq] << (8 * (bq - (w0 + j) * 8));
        }
    }
    int pc = 0;
    long long my_last = -1;
#pragma unroll
    for (int j = 0; j < EW; ++j) {
        pc += __popcll(words[j]);
        if (words[j]) my_last = (w0 + j) * WORD + (63 - __clzll(words[j]));
    }
    const ScanOut s = block_scan_sum_max(pc, my_last, lds);
    const long long sb0 = (long long)blockIdx.x * EW;  // first scan block under this workgroup (GROUP % EW == 0)
    const long long grp = sb0 / GROUP;
    long long pos = group_base[grp] + block_offset[sb0] + s.ex_sum;
    const long long gp = group_prev[grp], lp = block_prev[sb0];
    const long long bp = gp > lp ? gp : lp;
    long long prev = bp > s.ex_max ? bp : s.ex_max;  // last changed element before this thread's words (-1: none)
    unsigned long long max_r = 0, max_c = 0;
    const bool small = n <= 0xffffffffll;
#pragma unroll 1
    for (int j = 0; j < EW; ++j) {
        unsigned long long word = words[j];
        while (word) {
            const int bit = __ffsll((long long)word) - 1;
            word &= word - 1;
            const long long idx = (w0 + j) * WORD + bit;
            long long r, c, pr = -1, pcx = 0;
            if (small) {
                r = (unsigned)idx / (unsigned)cols, c = (unsigned)idx % (unsigned)cols;
                if (prev >= 0) pr = (unsigned)prev / (unsigned)cols, pcx = (unsigned)prev % (unsigned)cols;
            } else {
                r = idx / cols, c = idx % cols;
                if (prev >= 0) pr = prev / cols, pcx = prev % cols;
            }
            long long er = r, ec = c;
            if (delta && prev >= 0) {  // PatchBuilder.delta_encode: first entry absolute, column restarts on a new row
                er = r - pr;
                ec = (r == pr) ? c - pcx : c;
            }
            const DST v = Conv<SRC, DST>::cvt(value[idx]);
            out_rows[pos] = er, out_cols[pos] = ec, out_values[pos] = v;
            snap[idx] = v;
            max_r = (unsigned long long)er > max_r ? (unsigned long long)er : max_r;
            max_c = (unsigned long long)ec > max_c ? (unsigned long long)ec : max_c;
            prev = idx;
            ++pos;
        }
    }
    // per-block maxima (a same-address atomic per wave serialises: measured 10 ms for 5e5 waves); reduced by patch_maxima
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long orr = __shfl_xor(max_r, off, 64), occ = __shfl_xor(max_c, off, 64);
        max_r = orr > max_r ? orr : max_r, max_c = occ > max_c ? occ : max_c;
    }
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = max_r, s_mx[PT / 64 + (threadIdx.x >> 6)] = max_c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long mr = 0, mc = 0;
        for (int wv = 0; wv < PT / 64; ++wv) {
            mr = s_mx[wv] > mr ? s_mx[wv] : mr;
            mc = s_mx[PT / 64 + wv] > mc ? s_mx[PT / 64 + wv] : mc;
        }
        block_max_r[blockIdx.x] = (long long)mr, block_max_c[blockIdx.x] = (long long)mc;
    }
}

__global__ __launch_bounds__(1024) void patch_maxima_kernel(const long long* __restrict__ block_max_r,
                                                            const long long* __restrict__ block_max_c, long long nblocks,
                                                            unsigned long long* maxima) {
    __shared__ long long s_r[1024], s_c[1024];
    long long r = 0, c = 0;
    for (long long i = threadIdx.x; i < nblocks; i += 1024) {
        r = block_max_r[i] > r ? block_max_r[i] : r;
        c = block_max_c[i] > c ? block_max_c[i] : c;
    }
    s_r[threadIdx.x] = r, s_c[threadIdx.x] = c;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            s_r[threadIdx.x] = s_r[threadIdx.x + off] > s_r[threadIdx.x] ? s_r[threadIdx.x + off] : s_r[threadIdx.x];
            s_c[threadIdx.x] = s_c[threadIdx.x + off] > s_c[threadIdx.x] ? s_c[threadIdx.x + off] : s_c[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {  // the caller's maxima accumulate over the tensors of one patch
        maxima[0] = (unsigned long long)s_r[0] > maxima[0] ? (unsigned long long)s_r[0] : maxima[0];
        maxima[1] = (unsigned long long)s_c[0] > maxima[1] ? (unsigned long long)s_c[0] : maxima[1];
    }
}

// ---- receiver: decode + scatter -----------------------------------------------------------------------------------
constexpr int AI = 4;  // items per thread
constexpr int ABLOCK = PT * AI;
struct Agg {           // aggregate of a run of patch entries under delta_decode's two scans
    long long rsum;    // sum of row deltas
    long long csum;    // sum of column deltas since the last segment start inside the run (or over the whole run)
    int has_start;     // the run contains a segment start (row delta != 0, or global entry 0)
};
__device__ __forceinline__ Agg agg_combine(const Agg& a, const Agg& b) {
    Agg o;
    o.rsum = a.rsum + b.rsum;
    o.csum = b.has_start ? b.csum : a.csum + b.csum;
    o.has_start = a.has_start | b.has_start;
    return o;
}
__device__ __forceinline__ long long load_index(const void* p, int code, long long i) {
    switch (code) {
        case 0: return static_cast<const uint8_t*>(p)[i];
        case 1: return static_cast<const int32_t*>(p)[i];
        default: return static_cast<const long long*>(p)[i];
    }
}

__global__ __launch_bounds__(PT) void patch_apply_reduce_kernel(const void* rows, const void* cols, int rcode, int ccode,
                                                                long long nnz, Agg* __restrict__ block_agg) {
    __shared__ Agg s_agg[PT];
    const long long i0 = (long long)blockIdx.x * ABLOCK + (long long)threadIdx.x * AI;
    Agg a{0, 0, 0};
    for (int k = 0; k < AI; ++k) {
        const long long i = i0 + k;
        if (i >= nnz) break;
        const long long dr = load_index(rows, rcode, i), dc = load_index(cols, ccode, i);
        a = agg_combine(a, Agg{dr, dc, (dr != 0 || i == 0) ? 1 : 0});
    }
    s_agg[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        Agg t = s_agg[0];
        for (int j = 1; j < PT; ++j) t = agg_combine(t, s_agg[j]);
        block_agg[blockIdx.x] = t;
    }
}
__global__ void patch_apply_carry_kernel(Agg* block_agg, long long nblocks) {  // one thread: exclusive scan of the aggregates
    Agg run{0, 0, 0};
    for (long long b = 0; b < nblocks; ++b) {
        const Agg cur = block_agg[b];
        block_agg[b] = run;
        run = agg_combine(run, cur);
    }
}
template <int ES>
__global__ __launch_bounds__(PT) void patch_apply_scatter_kernel(uint8_t* __restrict__ target, long long tcols,
                                                                 const void* rows, const void* cols, int rcode, int ccode,
                                                                 int delta, const uint8_t* __restrict__ values, long long nnz,
                                                                 const Agg* __restrict__ block_agg) {
    __shared__ Agg s_agg[PT];
    const long long i0 = (long long)blockIdx.x * ABLOCK + (long long)threadIdx.x * AI;
    long long dr[AI], dc[AI];
    Agg mine{0, 0, 0};
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const long long i = i0 + k;
        dr[k] = dc[k] = 0;
        if (i < nnz) {
            dr[k] = load_index(rows, rcode, i), dc[k] = load_index(cols, ccode, i);
            if (delta) mine = agg_combine(mine, Agg{dr[k], dc[k], (dr[k] != 0 || i == 0) ? 1 : 0});
        }
    }
    Agg carry{0, 0, 0};
    if (delta) {
        s_agg[threadIdx.x] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {  // exclusive scan in place, seeded with the block's carry-in
            Agg run = block_agg[blockIdx.x];
            for (int j = 0; j < PT; ++j) {
                const Agg cur = s_agg[j];
                s_agg[j] = run;
                run = agg_combine(run, cur);
            }
        }
        __syncthreads();
        carry = s_agg[threadIdx.x];
    }
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const long long i = i0 + k;
        if (i >= nnz) break;
        long long r = dr[k], c = dc[k];
        if (delta) {
            carry = agg_combine(carry, Agg{dr[k], dc[k], (dr[k] != 0 || i == 0) ? 1 : 0});
            r = carry.rsum, c = carry.csum;
        }
        uint8_t* dst = target + (r * tcols + c) * ES;
        const uint8_t* src = values + i * ES;
#pragma unroll
        for (int b = 0; b < ES; ++b) dst[b] = src[b];
    }
}

struct Scratch {
    uint8_t* mask;
    int* block_count;
    long long *block_last, *block_offset, *block_prev, *block_max_c;  // block_last doubles as block_max_r after pass B
    long long *group_sum, *group_max, ngroups;
    long long nblocks;
};
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
long long scan_blocks(long long n) { return (n + BLOCK_ELEMS - 1) / BLOCK_ELEMS; }
Scratch carve(void* ws, long long n) {
    Scratch s;
    s.nblocks = scan_blocks(n);
    uint8_t* p = static_cast<uint8_t*>(ws);
    s.mask = p, p += align_up((size_t)s.nblocks * BLOCK_BYTES);
    s.block_count = reinterpret_cast<int*>(p), p += align_up((size_t)s.nblocks * sizeof(int));
    s.block_last = reinterpret_cast<long long*>(p), p += align_up((size_t)s.nblocks * 8);
    s.block_offset = reinterpret_cast<long long*>(p), p += align_up((size_t)s.nblocks * 8);
    s.block_prev = reinterpret_cast<long long*>(p), p += align_up((size_t)s.nblocks * 8);
    s.block_max_c = reinterpret_cast<long long*>(p), p += align_up((size_t)s.nblocks * 8);
    s.ngroups = (s.nblocks + GROUP - 1) / GROUP;
    s.group_sum = reinterpret_cast<long long*>(p), p += align_up((size_t)s.ngroups * 8);
    s.group_max = reinterpret_cast<long long*>(p);
    return s;
}

template <typename SRC, typename DST>
int run_scan(const void* value, const void* snap, long long n, const Scratch& s, long long* nnz, hipStream_t st) {
    const bool vec = (reinterpret_cast<uintptr_t>(value) % alignof(Pack<SRC, 8>)) == 0 &&
                     (reinterpret_cast<uintptr_t>(snap) % alignof(Pack<DST, 8>)) == 0;
    const SRC* v = static_cast<const SRC*>(value);
    const DST* sn = static_cast<const DST*>(snap);
    if (vec)
        hipLaunchKernelGGL((patch_scan_kernel<SRC, DST, true>), dim3((unsigned)s.nblocks), dim3(PT), 0, st, v, sn, n, s.mask,
                           s.block_count, s.block_last);
    else
        hipLaunchKernelGGL((patch_scan_kernel<SRC, DST, false>), dim3((unsigned)s.nblocks), dim3(PT), 0, st, v, sn, n, s.mask,
                           s.block_count, s.block_last);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(patch_offsets_local_kernel, dim3((unsigned)s.ngroups), dim3(GT), 0, st, s.block_count, s.block_last,
                       s.nblocks, s.block_offset, s.block_prev, s.group_sum, s.group_max);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(patch_offsets_groups_kernel, dim3(1), dim3(GT), 0, st, s.group_sum, s.group_max, s.ngroups, nnz);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
template <typename SRC, typename DST, int EW>
int launch_emit(const void* value, void* snap, long long n, long long cols, int delta, const Scratch& s, void* rows, void* colsb,
                void* values, void* maxima, hipStream_t st) {
    const long long eblocks = (s.nblocks + EW - 1) / EW;
    hipLaunchKernelGGL((patch_emit_kernel<SRC, DST, EW>), dim3((unsigned)eblocks), dim3(PT), 0, st, static_cast<const SRC*>(value),
                       static_cast<DST*>(snap), n, cols, delta, s.mask, s.block_offset, s.block_prev, s.group_sum, s.group_max,
                       static_cast<long long*>(rows), static_cast<long long*>(colsb), static_cast<DST*>(values), s.block_last,
                       s.block_max_c);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(patch_maxima_kernel, dim3(1), dim3(1024), 0, st, s.block_last, s.block_max_c, eblocks,
                       static_cast<unsigned long long*>(maxima));
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
template <typename SRC, typename DST>
int run_emit(const void* value, void* snap, long long n, long long cols, int delta, const Scratch& s, void* rows, void* colsb,
             void* values, void* maxima, long long nnz, hipStream_t st) {
    if (nnz * 64 < n) return launch_emit<SRC, DST, 8>(value, snap, n, cols, delta, s, rows, colsb, values, maxima, st);
    return launch_emit<SRC, DST, 1>(value, snap, n, cols, delta, s, rows, colsb, values, maxima, st);
}

// dispatch over (value dtype, snapshot dtype): equal dtypes, or an f32 sender feeding a bf16 / f16 receiver
#define RLX_PATCH_DISPATCH(FN, ...)                                                                              \
    do {                                                                                                         \
        if (src == dst) {                                                                                        \
            switch (dst) {                                                                                       \
                case RLX_DTYPE_F32: return FN<float, float>(__VA_ARGS__);                                        \
                case RLX_DTYPE_BF16: return FN<__bf16, __bf16>(__VA_ARGS__);                                     \
                case RLX_DTYPE_F16: return FN<_Float16, _Float16>(__VA_ARGS__);                                  \
                case RLX_DTYPE_RAW8: return FN<uint8_t, uint8_t>(__VA_ARGS__);                                   \
                case RLX_DTYPE_RAW16: return FN<uint16_t, uint16_t>(__VA_ARGS__);                                \
                case RLX_DTYPE_RAW32: return FN<uint32_t, uint32_t>(__VA_ARGS__);                                \
                case RLX_DTYPE_RAW64: return FN<unsigned long long, unsigned long long>(__VA_ARGS__);            \
                default: break;                                                                                  \
            }                                                                                                    \
        } else if (src == RLX_DTYPE_F32 && dst == RLX_DTYPE_BF16) {                                              \
            return FN<float, __bf16>(__VA_ARGS__);                                                               \
        } else if (src == RLX_DTYPE_F32 && dst == RLX_DTYPE_F16) {                                               \
            return FN<float, _Float16>(__VA_ARGS__);                                                             \
        }                                                                                                        \
        set_error("weight patch: unsupported dtype pair (%d -> %d)", src, dst);                                  \
        return RLX_EINVAL;                                                                                       \
    } while (0)

int elem_size(int dtype) {
    switch (dtype) {
        case RLX_DTYPE_F32: case RLX_DTYPE_RAW32: return 4;
        case RLX_DTYPE_BF16: case RLX_DTYPE_F16: case RLX_DTYPE_RAW16: return 2;
        case RLX_DTYPE_RAW8: return 1;
        case RLX_DTYPE_RAW64: return 8;
        default: return 0;
    }
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_patch_workspace_bytes(int64_t n_elems) {
    const long long nb = scan_blocks(n_elems > 0 ? n_elems : 1);
    const long long ng = (nb + GROUP - 1) / GROUP;
    return align_up((size_t)nb * BLOCK_BYTES) + align_up((size_t)nb * sizeof(int)) + 4 * align_up((size_t)nb * 8) +
           2 * align_up((size_t)ng * 8) + 256;
}

extern "C" int rlx_patch_scan(const void* value, int value_dtype, const void* snapshot, int snapshot_dtype, int64_t n_elems,
                              void* workspace, size_t workspace_bytes, int64_t* nnz, rlx_stream_t stream) {
    RLX_REQUIRE(n_elems >= 1 && value && snapshot && workspace && nnz, "rlx_patch_scan: bad argument");
    RLX_REQUIRE(n_elems < (1ll << 45), "rlx_patch_scan: tensor too large");
    if (workspace_bytes < rlx_patch_workspace_bytes(n_elems)) {
        set_error("rlx_patch_scan: workspace too small");
        return RLX_ENOSPC;
    }
    const Scratch s = carve(workspace, n_elems);
    const int src = value_dtype, dst = snapshot_dtype;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RLX_PATCH_DISPATCH(run_scan, value, snapshot, (long long)n_elems, s, reinterpret_cast<long long*>(nnz), st);
}

extern "C" int rlx_patch_emit(const void* value, int value_dtype, void* snapshot, int snapshot_dtype, int64_t n_elems,
                              int64_t cols, int delta_encoding, const void* workspace, int64_t nnz, int64_t* out_rows,
                              int64_t* out_cols, void* out_values, uint64_t* maxima, rlx_stream_t stream) {
    RLX_REQUIRE(nnz >= 1 && nnz <= n_elems, "rlx_patch_emit: nnz %lld out of range", (long long)nnz);
    RLX_REQUIRE(n_elems >= 1 && cols >= 1 && n_elems % cols == 0, "rlx_patch_emit: bad 2-D view (%lld elements, %lld cols)",
                (long long)n_elems, (long long)cols);
    RLX_REQUIRE(value && snapshot && workspace && out_rows && out_cols && out_values && maxima, "rlx_patch_emit: NULL argument");
    const Scratch s = carve(const_cast<void*>(workspace), n_elems);
    const int src = value_dtype, dst = snapshot_dtype;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RLX_PATCH_DISPATCH(run_emit, value, snapshot, (long long)n_elems, (long long)cols, delta_encoding, s, out_rows, out_cols,
                       out_values, maxima, (long long)nnz, st);
}

extern "C" size_t rlx_patch_apply_workspace_bytes(int64_t nnz) {
    return (size_t)((nnz + ABLOCK - 1) / ABLOCK + 1) * sizeof(Agg);
}

extern "C" int rlx_patch_apply(void* target, int dtype, int64_t target_rows, int64_t target_cols, const void* rows,
                               int rows_index_dtype, const void* cols, int cols_index_dtype, int delta_encoded,
                               const void* values, int64_t nnz, void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    RLX_REQUIRE(nnz >= 0 && target_rows >= 1 && target_cols >= 1, "rlx_patch_apply: bad sizes");
    if (nnz == 0) return RLX_OK;
    RLX_REQUIRE(target && rows && cols && values, "rlx_patch_apply: NULL argument");
    RLX_REQUIRE(rows_index_dtype >= 0 && rows_index_dtype <= 2 && cols_index_dtype >= 0 && cols_index_dtype <= 2,
                "rlx_patch_apply: index dtype codes are 0 (u8), 1 (i32), 2 (i64)");
    const int es = elem_size(dtype);
    RLX_REQUIRE(es != 0, "rlx_patch_apply: unknown dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long nb = (nnz + ABLOCK - 1) / ABLOCK;
    Agg* agg = static_cast<Agg*>(workspace);
    if (delta_encoded) {
        RLX_REQUIRE(workspace && workspace_bytes >= rlx_patch_apply_workspace_bytes(nnz), "rlx_patch_apply: workspace too small");
        hipLaunchKernelGGL(patch_apply_reduce_kernel, dim3((unsigned)nb), dim3(PT), 0, st, rows, cols, rows_index_dtype,
                           cols_index_dtype, (long long)nnz, agg);
        RLX_LAUNCH_CHECK();
        hipLaunchKernelGGL(patch_apply_carry_kernel, dim3(1), dim3(1), 0, st, agg, nb);
        RLX_LAUNCH_CHECK();
    }
    uint8_t* t = static_cast<uint8_t*>(target);
    const uint8_t* v = static_cast<const uint8_t*>(values);
#define RLX_APPLY(ES)                                                                                                   \
    hipLaunchKernelGGL((patch_apply_scatter_kernel<ES>), dim3((unsigned)nb), dim3(PT), 0, st, t, (long long)target_cols, rows, \
                       cols, rows_index_dtype, cols_index_dtype, delta_encoded, v, (long long)nnz, agg)
    switch (es) {
        case 1: RLX_APPLY(1); break;
        case 2: RLX_APPLY(2); break;
        case 4: RLX_APPLY(4); break;
        default: RLX_APPLY(8); break;
    }
#undef RLX_APPLY
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
