// ppo_step_bf16_parts.h -- device pieces shared by the bf16 fused launches (ppo_step_bf16.hip: column-split workgroups with the
// weights streamed to registers; ppo_step_bf16_rows.hip: row-split waves with the weights staged through an LDS ring).
#pragma once

#include "ppo_step_common.h"

namespace rlx {
namespace b16 {

using namespace loss;
using namespace step;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int XSB = 272;  // bf16 slab row stride (elements): 136 dwords == 8 (mod 16) -> conflict-free b128 fragment reads

typedef float f32x2 __attribute__((ext_vector_type(2)));

// tanh for two values at once, bf16 path: 1 - 2 / (exp(2x) + 1) with v_exp_f32 / v_rcp_f32 and packed f32 arithmetic
// (v_pk_mul / v_pk_add / v_pk_fma).  ~6e-8 absolute error: far inside the 2^-9 relative rounding the result gets next.  The
// f32 kernels keep the polynomial form (2e-7 RELATIVE near zero); here the epilogues are VALU-bound (phase stamps: three of
// them were 18 k of the kernel's 65 k cycles at ~27 issue slots per element) and this is ~10 slots per element.
__device__ __forceinline__ f32x2 tanh2_b(f32x2 x) {
    const f32x2 t = x * 2.8853900817779268f;  // 2 log2(e)
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2 d = e + 1.f;
    f32x2 q;
    q.x = __builtin_amdgcn_rcpf(d.x);
    q.y = __builtin_amdgcn_rcpf(d.y);
    return __builtin_elementwise_fma(q, f32x2{-2.f, -2.f}, f32x2{1.f, 1.f});
}

// A wave's tile values in the accumulator layout (lane (r16, kq) holds rows 4 kq .. 4 kq + 3 of column c = col0 + r16) -> the
// row-major bf16 slab, as 32-BIT words.  As four ds_write_b16 per tile this was the kernel's main source of LDS bank conflicts
// (round-2 counters: SQ_LDS_BANK_CONFLICT 30 % of the LDS-active cycles): stores bank on (a / 4) mod 32 within 32-lane groups,
// the 136-dword row stride puts rows r and r + 4 (lanes kq and kq + 1 of one group) on the same banks, and two lanes share every
// dword.  Instead neighbouring lanes swap halves (one DPP quad_perm [1, 0, 3, 2]): the even lane then holds columns (c, c + 1) of
// rows 4 kq, 4 kq + 1 and its odd neighbour the same columns of rows 4 kq + 2, 4 kq + 3 -- two ds_write_b32 per tile.  Which of its
// two rows a lane stores FIRST alternates with kq: one instruction's 32-lane group then covers rows {0, 2, 5, 7} (mod 8), i.e. row
// offsets {0, 16, 8, 24} (mod 32 dwords) x 8 dwords each = all 32 banks once.  Same values, same addresses: bit-identical slab.
__device__ __forceinline__ void store_slab_quad(__bf16* Xb, int row_base /* rt * 16 */, int col0 /* multiple of 16 */, bf16x4 v) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4, odd = lane & 1;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 p = __builtin_bit_cast(u32x2, v);                     // p.x = rows 0, 1 of this column; p.y = rows 2, 3
    const unsigned keep = odd ? p.y : p.x, send = odd ? p.x : p.y;
    const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    const unsigned lo = odd ? recv : keep, hi = odd ? keep : recv;   // lo: the even column's two rows, hi: the odd column's
    const unsigned w0 = __builtin_amdgcn_perm(hi, lo, 0x05040100u);  // first row of the pair:  (lo.low16, hi.low16)
    const unsigned w1 = __builtin_amdgcn_perm(hi, lo, 0x07060302u);  // second row of the pair: (lo.high16, hi.high16)
    const int row = row_base + 4 * kq + 2 * odd, flip = kq & 1;
    unsigned* base = reinterpret_cast<unsigned*>(Xb + col0 + (r16 & ~1));
    base[(row + flip) * (XSB / 2)] = flip ? w1 : w0;
    base[(row + 1 - flip) * (XSB / 2)] = flip ? w0 : w1;
}

// tanh'(z) = 1 - h^2 from the rounded activation the forward sweep kept (two bf16 -> f32 shifts and one fma per element:
// cheaper than carrying 96 f32 registers through the whole kernel, which pushed it into scratch)
__device__ __forceinline__ float dtanh_b(__bf16 h) {
    const float hf = (float)h;
    return fmaf(-hf, hf, 1.f);
}

// ---- the f32 heads on the matrix pipe ----------------------------------------------------------------------------------
// The heads keep f32 weights and f32 outputs (file header).  An f32 number is exactly the sum of three bf16 numbers
// (hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid): 3 x 8 significand bits, the differences are exact in f32), and a
// product of two bf16 numbers is exact in f32 -- so  x . w  with x in bf16 is three bf16 MFMAs with f32 accumulation: the same
// real-number sum as the f32 dot product, only the order of the f32 additions differs.  As VALU dot products the head phases
// were LDS-bound (every (row, output) pair re-read its 1 KiB weight row: ~6 k cycles of b128 reads per tile).
struct Split3 {
    bf16x8 hi, mid, lo;
};
__device__ __forceinline__ Split3 split3(const float (&w)[8]) {
    Split3 q;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)w[j];
        const float r1 = fsub(w[j], (float)h);
        const __bf16 m = (__bf16)r1;
        const float r2 = fsub(r1, (float)m);
        q.hi[j] = h;
        q.mid[j] = m;
        q.lo[j] = (__bf16)r2;
    }
    return q;
}
__device__ __forceinline__ f32x4 mfma3(bf16x8 a, const Split3& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.mid, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.lo, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma3(const Split3& a, bf16x8 b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b, acc, 0, 0, 0);
}

struct TileGeom {
    int nrb;
    __host__ __device__ size_t mat() const { return (size_t)16 * nrb * 512; }  // elements per [M][256] image
};

// STAMPS: the development build of the same kernel with phase stamps (tools/phase_times.py); reading the cycle counter orders
// the surrounding memory operations (+5 us measured), so the product instantiation carries none.
template <bool ON>
struct StampsT {
    long long* buf;
    int n;
    __device__ __forceinline__ void mark() {
        if constexpr (ON) {
            if (buf != nullptr && blockIdx.x == 0 && blockIdx.y == 1 && threadIdx.x == 0) buf[n] = (long long)clock64();  // the policy network's tile 0
            ++n;
        }
    }
};

// The small parameter inputs of a tile: the three hidden layers' biases, the f32 head image + bias, the policy's log-std.
// issue() requests them into registers (clamped, unconditional), commit() writes them to LDS: sBias [3][HID], W4s / b4s, and
// sStd = std | var | log(std) per output (the same expf / fmul / logf the per-element code used to repeat).
template <int NT>
struct SmallInputsB {
    static constexpr int NB = (3 * HID + NT - 1) / NT, NH = (MAX_OUT * 64 + NT - 1) / NT;
    float bv[NB];
    f32x4 hw[NH];
    float b4v, lsv;
    __device__ __forceinline__ void issue(const float* __restrict__ params, const rlx_mlp_layout& lay, int y, int n_out) {
        const int tid = threadIdx.x;
        // (lay lives in the kernel arguments: indexing it with a per-lane value would turn into a VECTOR load of the argument
        //  block and a dependent round trip -- the three offsets are read as scalars and selected per lane.)
        const long long ob0 = lay.off_b[y][0], ob1 = lay.off_b[y][1], ob2 = lay.off_b[y][2];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = min(tid + u * NT, 3 * HID - 1), l = i / HID;
            bv[u] = params[(l == 0 ? ob0 : l == 1 ? ob1 : ob2) + i % HID];
        }
        const float* W4 = params + lay.off_w[y][3];
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int f = min(tid + u * NT, n_out * 64 - 1), o = f >> 6, c4 = (f & 63) * 4;
            hw[u] = *reinterpret_cast<const f32x4*>(W4 + (size_t)o * HID + c4);
        }
        // unconditional, clamped loads (a load under a branch gets its wait under the branch too)
        const int oc = min(tid, n_out - 1);
        const long long ob3 = lay.off_b[y][3];
        b4v = params[(ob3 >= 0 ? ob3 : 0) + oc];
        if (ob3 < 0) b4v = 0.f;
        lsv = params[y == 1 ? lay.off_logstd + oc : 0];
    }
    __device__ __forceinline__ void commit(int n_out, float* sBias, float* W4s, float* b4s, float* sStd) const {
        const int tid = threadIdx.x;
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (tid + u * NT < 3 * HID) sBias[tid + u * NT] = bv[u];
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            const int f = tid + u * NT;
            if (f < n_out * 64) *reinterpret_cast<f32x4*>(W4s + (f >> 6) * W4S + (f & 63) * 4) = hw[u];
        }
        if (tid < n_out) {  // (the value network's lanes compute the std terms of a junk word: never read)
            b4s[tid] = b4v;
            const float stdv = expf(lsv);
            sStd[tid] = stdv;
            sStd[MAX_OUT + tid] = fmul(stdv, stdv);
            sStd[2 * MAX_OUT + tid] = logf(stdv);
        }
    }
};

}  // namespace b16
}  // namespace rlx
