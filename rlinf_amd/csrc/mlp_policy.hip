// mlp_policy.hip -- the rollout / training forward and backward of the tanh-MLP Gaussian policy + value
// head, gfx950.  Dense layers on the f32-input MFMA, everything else fused around them.
//
// Replaces MLPPolicy (rlinf/models/embodiment/mlp_policy/mlp_policy.py): _sample_actions :238-254,
// _generate_actions :256-293, predict_action_batch :295-320 (rollout), default_forward :202-236 (training
// forward); ValueHead (rlinf/models/embodiment/modules/value_head.py:17-66); torch.distributions.Normal's
// sample / log_prob / entropy; and autograd's backward through all of it.
//
// Work decomposition
//   forward / backward-data: one workgroup (4 waves) owns a 32-row tile of the batch and ONE network
//     (blockIdx.y: 0 = value head, 1 = actor).  The tile's activations live in an LDS slab [32][260] f32 and
//     are overwritten in place layer by layer; weights stream from L2 through a double-buffered LDS chunk
//     [256 out][32 k (+4 pad)].  Wave w produces output columns [64w, 64w+64) = two 32x32 MFMA tiles.
//     v_mfma_f32_32x32x2_f32 takes one f32 of A and B per lane; a lane reads 4 consecutive k with ONE
//     ds_read_b128 per operand and feeds 4 MFMAs with them (lanes 0-31 hold k..k+3, lanes 32-63 k+4..k+7: the
//     reduction order is permuted identically for A and B).  Row strides 260 / 36 floats make every b128
//     lane group hit 16 distinct 16-byte bank slots.
//   backward-weights: dW_l = dZ_l^T H_{l-1} is a GEMM whose reduction runs over the batch rows: split-K over
//     row ranges (slabs), 128x128 output tiles, 2x2 MFMA tiles per wave; bias gradients ride along as column
//     sums of the A fragments.  Slabs are summed by the optimizer kernel (rlx_clip_adamw_step).
// Exactness: the MFMA is an exact-f32 fmaf chain, so results differ from the CPU reference only by summation
// order (tests: rtol 1e-4 / atol 1e-5 on activations-derived outputs).

#include <algorithm>

#include "rlx_common.h"

namespace rlx {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 32;     // rows per workgroup tile
constexpr int HID = 256;   // hidden width (fixed by the reference: hidden_sizes=(256,256,256))
constexpr int XS = 260;    // LDS row stride of the activation slab (floats)
constexpr int KC = 32;     // k per weight chunk
constexpr int WS = 36;     // LDS row stride of a weight chunk (floats)
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;
constexpr size_t FWD_LDS = (size_t)(BM * XS + 2 * HID * WS) * sizeof(float) + 32 * 32 * sizeof(float);

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct Packed {  // float offsets inside the packed image
    int k1p;
    __host__ __device__ size_t per_net() const { return (size_t)HID * k1p + 2 * (size_t)HID * HID; }
    __host__ __device__ size_t w1p(int y) const { return y * per_net(); }
    __host__ __device__ size_t w2t(int y) const { return y * per_net() + (size_t)HID * k1p; }
    __host__ __device__ size_t w3t(int y) const { return w2t(y) + (size_t)HID * HID; }
};

__global__ __launch_bounds__(256) void mlp_pack_kernel(const float* __restrict__ params, rlx_mlp_layout lay,
                                                       float* __restrict__ packed) {
    Packed pk{round_up(lay.obs_dim, 8)};
    const size_t total = 2 * pk.per_net();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / pk.per_net());
        size_t r = i - y * pk.per_net();
        float val;
        if (r < (size_t)HID * pk.k1p) {  // W1 zero-padded to k1p columns
            const int row = (int)(r / pk.k1p), k = (int)(r % pk.k1p);
            val = k < lay.obs_dim ? params[lay.off_w[y][0] + (size_t)row * lay.obs_dim + k] : 0.f;
        } else {
            r -= (size_t)HID * pk.k1p;
            const int l = 1 + (int)(r / ((size_t)HID * HID));  // layer 1 or 2 (second / third Linear)
            r %= (size_t)HID * HID;
            const int j = (int)(r / HID), k = (int)(r % HID);   // Wt[j = in][k = out] = W[k][j]
            val = params[lay.off_w[y][l] + (size_t)k * HID + j];
        }
        packed[i] = val;
    }
}

// ---------------------------------------------------------------------------------------------------
// acc[t] (t = 0,1) = Xs[0:32, 0:K] . Wg[64*wave + 32*t : +32, 0:K]^T        (all 256 threads must call)
//   Wg: global, row-major [256][ldw], ldw % 4 == 0, K % 8 == 0.  Ends with a workgroup barrier.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_gemm(const float* __restrict__ Wg, int ldw, int K, const float* Xs, float* Ws,
                                          f32x16 (&acc)[2]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, khalf = lane >> 5;
    const int ldrow = tid >> 3, ldc4 = (tid & 7) * 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float4 stage[8];
    auto gload = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kc + ldc4;
            stage[i] = k < K ? *reinterpret_cast<const float4*>(Wg + (size_t)(ldrow + 32 * i) * ldw + k)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto swrite = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(buf + (ldrow + 32 * i) * WS + ldc4) = stage[i];
    };
    const int nchunks = (K + KC - 1) / KC;
    gload(0);
    swrite(Ws);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float* cur = Ws + (c & 1) * HID * WS;
        if (c + 1 < nchunks) gload((c + 1) * KC);
        const int kmax = min(KC, K - c * KC);
        const float* xa = Xs + lrow * XS + c * KC + 4 * khalf;
        const float* wb = cur + (wave * 64 + lrow) * WS + 4 * khalf;
#pragma unroll 4
        for (int kk = 0; kk < kmax; kk += 8) {
            const float4 a = *reinterpret_cast<const float4*>(xa + kk);
            const float4 b0 = *reinterpret_cast<const float4*>(wb + kk);
            const float4 b1 = *reinterpret_cast<const float4*>(wb + 32 * WS + kk);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
        }
        if (c + 1 < nchunks) swrite(Ws + ((c + 1) & 1) * HID * WS);
        __syncthreads();
    }
}

// C/D fragment of the 32x32 MFMA: register r of lane l holds (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31)
__device__ __forceinline__ int frag_row(int r, int khalf) { return (r & 3) + 8 * (r >> 2) + 4 * khalf; }

// Hidden-layer epilogue of the forward pass: h = tanh(acc + bias) -> activation slab (in place) [+ global save]
__device__ __forceinline__ void fwd_epilogue(const f32x16 (&acc)[2], const float* __restrict__ bias, float* Xs,
                                             float* __restrict__ save, long long m0, long long M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = wave * 64 + t * 32 + lrow;
        const float b = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = frag_row(r, khalf);
            const float h = tanhf(acc[t][r] + b);
            Xs[row * XS + col] = h;
            if (save != nullptr && m0 + row < M) save[(size_t)(m0 + row) * HID + col] = h;
        }
    }
    __syncthreads();
}

// Backward-data epilogue: dz = acc * (1 - h^2) with h = saved activation of the layer below
__device__ __forceinline__ void bwd_epilogue(const f32x16 (&acc)[2], const float* __restrict__ hsaved, float* Xs,
                                             float* __restrict__ dz_out, long long m0, long long M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = wave * 64 + t * 32 + lrow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = frag_row(r, khalf);
            float dz = 0.f;
            if (m0 + row < M) {
                const float h = hsaved[(size_t)(m0 + row) * HID + col];
                dz = acc[t][r] * (1.f - h * h);
                dz_out[(size_t)(m0 + row) * HID + col] = dz;
            }
            Xs[row * XS + col] = dz;
        }
    }
    __syncthreads();
}

struct FwdArgs {
    const float* params;
    const float* packed;
    rlx_mlp_layout lay;
    const float* states;
    const float* eps;      // rollout: N(0,1) draws or nullptr (eval)
    const float* action;   // train: stored actions
    long long M;
    float* out_action;     // rollout
    float* out_logprob;
    float* out_entropy;    // train
    float* out_value;
    float* out_mean;       // train
    float* acts;           // train: [2][3][M][256]
};

// MODE 0: rollout (sample + logprob + value).  MODE 1: training forward (logprob of stored action, entropy,
// value; saves mean and hidden activations).
template <int MODE>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(FwdArgs a) {
    extern __shared__ __align__(16) float smem[];
    float* Xs = smem;
    float* Ws = smem + BM * XS;
    const int y = blockIdx.y;  // 0 = value head, 1 = actor
    const long long m0 = (long long)blockIdx.x * BM;
    const long long M = a.M;
    const rlx_mlp_layout& lay = a.lay;
    const int D = lay.obs_dim;
    Packed pk{round_up(D, 8)};
    const int tid = threadIdx.x;

    // obs-preprocess: the state rows themselves; zero-pad the k tail and the rows past M
    for (int i = tid; i < BM * pk.k1p; i += 256) {
        const int r = i / pk.k1p, c = i % pk.k1p;
        Xs[r * XS + c] = (c < D && m0 + r < M) ? a.states[(size_t)(m0 + r) * D + c] : 0.f;
    }
    f32x16 acc[2];
    float* save = MODE == 1 ? a.acts + (size_t)(y * 3) * M * HID : nullptr;
    tile_gemm(a.packed + pk.w1p(y), pk.k1p, pk.k1p, Xs, Ws, acc);
    fwd_epilogue(acc, a.params + lay.off_b[y][0], Xs, save, m0, M);
    tile_gemm(a.params + lay.off_w[y][1], HID, HID, Xs, Ws, acc);
    fwd_epilogue(acc, a.params + lay.off_b[y][1], Xs, save ? save + (size_t)M * HID : nullptr, m0, M);
    tile_gemm(a.params + lay.off_w[y][2], HID, HID, Xs, Ws, acc);
    fwd_epilogue(acc, a.params + lay.off_b[y][2], Xs, save ? save + 2 * (size_t)M * HID : nullptr, m0, M);

    // ---- heads (N <= a few outputs: plain FMAs from the LDS slab) + fused distribution epilogue --------------
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const float* W4 = a.params + lay.off_w[y][3];
    const float* b4 = lay.off_b[y][3] >= 0 ? a.params + lay.off_b[y][3] : nullptr;
    for (int idx = tid; idx < BM * n_out; idx += 256) {
        const int row = idx / n_out, o = idx % n_out;
        const float* xr = Xs + row * XS;
        const float* wr = W4 + (size_t)o * HID;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < HID; j += 4) {
            const float4 x = *reinterpret_cast<const float4*>(xr + j);
            const float4 w = *reinterpret_cast<const float4*>(wr + j);
            s = fmaf(x.x, w.x, s);
            s = fmaf(x.y, w.y, s);
            s = fmaf(x.z, w.z, s);
            s = fmaf(x.w, w.w, s);
        }
        if (b4) s += b4[o];
        if (m0 + row >= M) continue;
        const size_t g = (size_t)(m0 + row) * n_out + o;
        if (y == 0) {
            a.out_value[g] = s;
        } else {
            const float mean = s;
            const float logstd = a.params[lay.off_logstd + o];
            const float stdv = expf(logstd);
            float act;
            if (MODE == 0) act = a.eps ? fadd(fmul(a.eps[g], stdv), mean) : mean;  // torch.normal: eps*std + mean
            else act = a.action[g];
            const float d = fsub(act, mean);
            const float var = fmul(stdv, stdv);
            const float log_scale = logf(stdv);
            // Normal.log_prob: -((x - loc)**2) / (2*var) - log(scale) - log(sqrt(2*pi))
            const float lp = fsub(fsub((-fmul(d, d)) / fmul(2.f, var), log_scale), LOG_SQRT_2PI);
            a.out_logprob[g] = lp;
            if (MODE == 0) {
                a.out_action[g] = act;
            } else {
                a.out_mean[g] = mean;
                a.out_entropy[g] = fadd(1.4189385332046727f, log_scale);  // 0.5 + 0.5*log(2*pi) + log(scale)
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward-data chain: head gradients -> dZ3 -> dZ2 -> dZ1 for one 32-row tile of one network
// ---------------------------------------------------------------------------------------------------
struct BwdArgs {
    const float* params;
    const float* packed;
    rlx_mlp_layout lay;
    const float* action;
    const float* mean;
    const float* acts;       // [2][3][M][256]
    const float* d_logprob;  // [M][act_dim]
    const float* d_entropy;  // [M][act_dim] or nullptr
    const float* d_value;    // [M][val_dim]
    long long M;
    float* dz;               // [2][3][M][256]
    float* dmu;              // [M][act_dim]
    float* dls;              // [M][act_dim]  per-sample d/d logstd
};

__global__ __launch_bounds__(256) void mlp_bwd_dz_kernel(BwdArgs a) {
    extern __shared__ __align__(16) float smem[];
    float* Xs = smem;
    float* Ws = smem + BM * XS;
    float* sOut = Ws + 2 * HID * WS;  // [32][n_out <= 32]
    const int y = blockIdx.y;
    const long long m0 = (long long)blockIdx.x * BM, M = a.M;
    const rlx_mlp_layout& lay = a.lay;
    Packed pk{round_up(lay.obs_dim, 8)};
    const int tid = threadIdx.x;
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;

    for (int idx = tid; idx < BM * n_out; idx += 256) {
        const int row = idx / n_out, o = idx % n_out;
        float dout = 0.f;
        if (m0 + row < M) {
            const size_t g = (size_t)(m0 + row) * n_out + o;
            if (y == 0) {
                dout = a.d_value[g];
            } else {
                const float stdv = expf(a.params[lay.off_logstd + o]);
                const float var = stdv * stdv;
                const float d = a.action[g] - a.mean[g];
                const float dlp = a.d_logprob[g];
                dout = dlp * d / var;                                    // d logprob / d mean
                float dl = dlp * (d * d / var - 1.f);                    // d logprob / d logstd
                if (a.d_entropy) dl += a.d_entropy[g];                   // d entropy / d logstd = 1
                a.dmu[g] = dout;
                a.dls[g] = dl;
            }
        }
        sOut[row * 32 + o] = dout;
    }
    __syncthreads();
    {   // dZ3[row][j] = (sum_o dOut[row][o] * W4[o][j]) * (1 - H3^2); thread = column j
        const int j = tid;
        const float* W4 = a.params + lay.off_w[y][3];
        const float* h3 = a.acts + (size_t)(y * 3 + 2) * M * HID;
        float* dz3 = a.dz + (size_t)(y * 3 + 2) * M * HID;
        float w4[32];  // this thread's column of the head weight (n_out <= 32, checked on the host)
#pragma unroll
        for (int o = 0; o < 32; ++o) w4[o] = o < n_out ? W4[(size_t)o * HID + j] : 0.f;
        for (int row = 0; row < BM; ++row) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < 32; ++o)
                if (o < n_out) s = fmaf(sOut[row * 32 + o], w4[o], s);
            float dzv = 0.f;
            if (m0 + row < M) {
                const float h = h3[(size_t)(m0 + row) * HID + j];
                dzv = s * (1.f - h * h);
                dz3[(size_t)(m0 + row) * HID + j] = dzv;
            }
            Xs[row * XS + j] = dzv;
        }
    }
    f32x16 acc[2];
    tile_gemm(a.packed + pk.w3t(y), HID, HID, Xs, Ws, acc);  // dH2 = dZ3 . W3
    bwd_epilogue(acc, a.acts + (size_t)(y * 3 + 1) * M * HID, Xs, a.dz + (size_t)(y * 3 + 1) * M * HID, m0, M);
    tile_gemm(a.packed + pk.w2t(y), HID, HID, Xs, Ws, acc);  // dH1 = dZ2 . W2
    bwd_epilogue(acc, a.acts + (size_t)(y * 3 + 0) * M * HID, Xs, a.dz + (size_t)(y * 3 + 0) * M * HID, m0, M);
}

// ---------------------------------------------------------------------------------------------------
// backward-weights: slab s of dW[y][l] (l = 0..2) and db[y][l]
//   grid = (slabs, 4 output tiles of 128x128, 6 matrices)
// ---------------------------------------------------------------------------------------------------
struct DwArgs {
    rlx_mlp_layout lay;
    const float* states;
    const float* acts;
    const float* dz;
    long long M;
    int rows_per_slab;  // multiple of 32
    float* grads;       // [slabs][n_params]
};

__global__ __launch_bounds__(256) void mlp_bwd_dw_kernel(DwArgs a) {
    __shared__ __align__(16) float As[2][32][128];
    __shared__ __align__(16) float Bs[2][32][128];
    const int s = blockIdx.x, tile = blockIdx.y, mat = blockIdx.z;
    const int y = mat / 3, l = mat % 3;
    const rlx_mlp_layout& lay = a.lay;
    const long long M = a.M;
    const int Kin = l == 0 ? lay.obs_dim : HID;
    const int i0 = (tile >> 1) * 128, j0 = (tile & 1) * 128;
    if (j0 >= Kin) return;
    const float* A = a.dz + (size_t)(y * 3 + l) * M * HID;                           // [M][256]
    const float* Bm = l == 0 ? a.states : a.acts + (size_t)(y * 3 + l - 1) * M * HID;  // [M][Kin]
    const bool vecB = (Kin % 4) == 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, khalf = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const long long r_begin = (long long)s * a.rows_per_slab;
    const long long r_end = min(M, r_begin + a.rows_per_slab);
    const int nchunks = r_end > r_begin ? (int)((r_end - r_begin + 31) / 32) : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    float bsum[2] = {0.f, 0.f};

    const int ldr = tid >> 5, ldc4 = (tid & 31) * 4;  // loader: row ldr + 8*i, 4 columns at ldc4
    float4 sa[4], sb[4];
    auto gload = [&](int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long row = r_begin + (long long)c * 32 + ldr + 8 * i;
            const bool ok = row < r_end;
            sa[i] = ok ? *reinterpret_cast<const float4*>(A + (size_t)row * HID + i0 + ldc4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const int col = j0 + ldc4;
                if (vecB) {
                    if (col < Kin) v = *reinterpret_cast<const float4*>(Bm + (size_t)row * Kin + col);
                } else {
                    const float* p = Bm + (size_t)row * Kin + col;
                    if (col + 0 < Kin) v.x = p[0];
                    if (col + 1 < Kin) v.y = p[1];
                    if (col + 2 < Kin) v.z = p[2];
                    if (col + 3 < Kin) v.w = p[3];
                }
            }
            sb[i] = v;
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[buf][ldr + 8 * i][ldc4]) = sa[i];
            *reinterpret_cast<float4*>(&Bs[buf][ldr + 8 * i][ldc4]) = sb[i];
        }
    };
    if (nchunks > 0) {
        gload(0);
        swrite(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunks) gload(c + 1);
#pragma unroll 4
        for (int kp = 0; kp < 16; ++kp) {
            const int k = 2 * kp + khalf;
            const float a0 = As[cur][k][wi * 64 + lrow], a1 = As[cur][k][wi * 64 + 32 + lrow];
            const float b0 = Bs[cur][k][wj * 64 + lrow], b1 = Bs[cur][k][wj * 64 + 32 + lrow];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            bsum[0] += a0;
            bsum[1] += a1;
        }
        if (c + 1 < nchunks) swrite(cur ^ 1);
        __syncthreads();
    }
    float* slab = a.grads + (size_t)s * lay.n_params;
    float* dW = slab + lay.off_w[y][l];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = j0 + wj * 64 + u * 32 + lrow;
            if (col < Kin) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = i0 + wi * 64 + t * 32 + frag_row(r, khalf);
                    dW[(size_t)row * Kin + col] = acc[t][u][r];
                }
            }
        }
    if (j0 == 0 && wj == 0) {  // bias gradient = column sums of dZ, folded in as the A fragments go by
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float tot = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            if (khalf == 0) slab[lay.off_b[y][l] + i0 + wi * 64 + t * 32 + lrow] = tot;
        }
    }
}

// head weights / bias / logstd: slab s.  grid = (slabs, 2 nets), thread = hidden column j
__global__ __launch_bounds__(256) void mlp_head_dw_kernel(rlx_mlp_layout lay, const float* __restrict__ acts,
                                                          const float* __restrict__ dmu, const float* __restrict__ dls,
                                                          const float* __restrict__ d_value, long long M, int rows_per_slab,
                                                          float* __restrict__ grads) {
    const int s = blockIdx.x, y = blockIdx.y, j = threadIdx.x;
    const int n_out = y == 1 ? lay.act_dim : lay.val_dim;
    const float* dout = y == 1 ? dmu : d_value;
    const float* h3 = acts + (size_t)(y * 3 + 2) * M * HID;
    const long long r_begin = (long long)s * rows_per_slab, r_end = min(M, r_begin + rows_per_slab);
    float* slab = grads + (size_t)s * lay.n_params;
    for (int o0 = 0; o0 < n_out; o0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int no = min(8, n_out - o0);
        for (long long row = r_begin; row < r_end; ++row) {
            const float h = h3[(size_t)row * HID + j];
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < no) acc[o] = fmaf(dout[(size_t)row * n_out + o0 + o], h, acc[o]);
        }
        for (int o = 0; o < no; ++o) slab[lay.off_w[y][3] + (size_t)(o0 + o) * HID + j] = acc[o];
    }
    if (y == 1 && j < n_out) {
        float sb = 0.f, sl = 0.f;
        for (long long row = r_begin; row < r_end; ++row) {
            sb += dmu[(size_t)row * n_out + j];
            sl += dls[(size_t)row * n_out + j];
        }
        if (lay.off_b[1][3] >= 0) slab[lay.off_b[1][3] + j] = sb;
        slab[lay.off_logstd + j] = sl;
    }
}

int check_layout(const rlx_mlp_layout* lay, const char* who) {
    RLX_REQUIRE(lay != nullptr, "%s: NULL layout", who);
    RLX_REQUIRE(lay->hidden == HID, "%s: hidden=%d is not supported (the reference's MLP policy is 256 wide)", who, lay->hidden);
    RLX_REQUIRE(lay->obs_dim >= 1 && lay->obs_dim <= 256, "%s: obs_dim=%d out of range [1,256]", who, lay->obs_dim);
    RLX_REQUIRE(lay->act_dim >= 1 && lay->act_dim <= 32 && lay->val_dim >= 1 && lay->val_dim <= 32,
                "%s: act_dim=%d / val_dim=%d out of range [1,32]", who, lay->act_dim, lay->val_dim);
    for (int y = 0; y < 2; ++y)
        for (int l = 0; l < 4; ++l) {
            RLX_REQUIRE(lay->off_w[y][l] >= 0 && lay->off_w[y][l] < lay->n_params, "%s: weight offset out of range", who);
            RLX_REQUIRE(lay->off_w[y][l] % 4 == 0 || l == 0, "%s: weight offsets of layers 2-4 must be 16-byte aligned", who);
            RLX_REQUIRE(l == 3 || lay->off_b[y][l] >= 0, "%s: hidden layers need a bias", who);
        }
    return RLX_OK;
}

// once per kernel and process (not a stream operation: keep it out of hipGraph capture regions)
template <typename K>
int set_lds(K kern, size_t bytes) {
    static thread_local const void* done[8] = {};
    const void* key = reinterpret_cast<const void*>(kern);
    for (const void* d : done)
        if (d == key) return RLX_OK;
    RLX_HIP_CHECK(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    for (auto& d : done)
        if (d == nullptr) { d = key; break; }
    return RLX_OK;
}

}  // namespace
}  // namespace rlx

using namespace rlx;

extern "C" size_t rlx_mlp_packed_bytes(const rlx_mlp_layout* lay) {
    if (!lay) return 0;
    Packed pk{round_up(lay->obs_dim, 8)};
    return 2 * pk.per_net() * sizeof(float);
}

extern "C" int rlx_mlp_pack(const float* params, const rlx_mlp_layout* lay, float* packed, rlx_stream_t stream) {
    if (int rc = check_layout(lay, "rlx_mlp_pack")) return rc;
    RLX_REQUIRE(params && packed, "rlx_mlp_pack: NULL argument");
    hipLaunchKernelGGL(mlp_pack_kernel, dim3(num_cu() * 2), dim3(256), 0, static_cast<hipStream_t>(stream), params, *lay, packed);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_mlp_rollout(const float* params, const float* packed, const rlx_mlp_layout* lay, const float* states,
                               const float* eps, int64_t m, float* action, float* logprob, float* value,
                               rlx_stream_t stream) {
    if (int rc = check_layout(lay, "rlx_mlp_rollout")) return rc;
    RLX_REQUIRE(m >= 0, "rlx_mlp_rollout: negative batch");
    if (m == 0) return RLX_OK;
    RLX_REQUIRE(params && packed && states && action && logprob && value, "rlx_mlp_rollout: NULL argument");
    if (int rc = set_lds(mlp_fwd_kernel<0>, FWD_LDS)) return rc;
    FwdArgs a{};
    a.params = params; a.packed = packed; a.lay = *lay; a.states = states; a.eps = eps; a.M = m;
    a.out_action = action; a.out_logprob = logprob; a.out_value = value;
    hipLaunchKernelGGL(mlp_fwd_kernel<0>, dim3(ceil_div(m, BM), 2), dim3(256), FWD_LDS, static_cast<hipStream_t>(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_mlp_value(const float* params, const float* packed, const rlx_mlp_layout* lay, const float* states, int64_t m,
                             float* value, rlx_stream_t stream) {
    if (int rc = check_layout(lay, "rlx_mlp_value")) return rc;
    RLX_REQUIRE(m >= 0, "rlx_mlp_value: negative batch");
    if (m == 0) return RLX_OK;
    RLX_REQUIRE(params && packed && states && value, "rlx_mlp_value: NULL argument");
    if (int rc = set_lds(mlp_fwd_kernel<0>, FWD_LDS)) return rc;
    FwdArgs a{};
    a.params = params; a.packed = packed; a.lay = *lay; a.states = states; a.M = m; a.out_value = value;
    // gridDim.y == 1: only net 0 (the value head) runs
    hipLaunchKernelGGL(mlp_fwd_kernel<0>, dim3(ceil_div(m, BM), 1), dim3(256), FWD_LDS, static_cast<hipStream_t>(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_mlp_train_fwd(const float* params, const float* packed, const rlx_mlp_layout* lay, const float* states,
                                 const float* action, int64_t m, float* logprob, float* entropy, float* value, float* mean,
                                 float* acts, rlx_stream_t stream) {
    if (int rc = check_layout(lay, "rlx_mlp_train_fwd")) return rc;
    RLX_REQUIRE(m >= 0, "rlx_mlp_train_fwd: negative batch");
    if (m == 0) return RLX_OK;
    RLX_REQUIRE(params && packed && states && action && logprob && entropy && value && mean && acts,
                "rlx_mlp_train_fwd: NULL argument");
    if (int rc = set_lds(mlp_fwd_kernel<1>, FWD_LDS)) return rc;
    FwdArgs a{};
    a.params = params; a.packed = packed; a.lay = *lay; a.states = states; a.action = action; a.M = m;
    a.out_logprob = logprob; a.out_entropy = entropy; a.out_value = value; a.out_mean = mean; a.acts = acts;
    hipLaunchKernelGGL(mlp_fwd_kernel<1>, dim3(ceil_div(m, BM), 2), dim3(256), FWD_LDS, static_cast<hipStream_t>(stream), a);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}

extern "C" int rlx_mlp_bwd_slabs(int64_t m) {
    // 512-row slabs (16 MFMA chunks each), at least 1, at most 64
    return (int)std::max<int64_t>(1, std::min<int64_t>(64, (m + 511) / 512));
}

extern "C" size_t rlx_mlp_bwd_workspace_bytes(const rlx_mlp_layout* lay, int64_t m) {
    if (!lay || m <= 0) return 16;
    return ((size_t)6 * m * HID + 2 * (size_t)m * lay->act_dim) * sizeof(float);
}

extern "C" int rlx_mlp_train_bwd(const float* params, const float* packed, const rlx_mlp_layout* lay, const float* states,
                                 const float* action, const float* mean, const float* acts, const float* d_logprob,
                                 const float* d_entropy, const float* d_value, int64_t m, float* grads, int slabs,
                                 void* workspace, size_t workspace_bytes, rlx_stream_t stream) {
    if (int rc = check_layout(lay, "rlx_mlp_train_bwd")) return rc;
    RLX_REQUIRE(m >= 1, "rlx_mlp_train_bwd: empty batch");
    RLX_REQUIRE(slabs >= 1, "rlx_mlp_train_bwd: slabs=%d", slabs);
    RLX_REQUIRE(params && packed && states && action && mean && acts && d_logprob && d_value && grads && workspace,
                "rlx_mlp_train_bwd: NULL argument");
    if (workspace_bytes < rlx_mlp_bwd_workspace_bytes(lay, m)) {
        set_error("rlx_mlp_train_bwd: workspace %zu < %zu bytes", workspace_bytes, rlx_mlp_bwd_workspace_bytes(lay, m));
        return RLX_ENOSPC;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* dz = static_cast<float*>(workspace);
    float* dmu = dz + (size_t)6 * m * HID;
    float* dls = dmu + (size_t)m * lay->act_dim;
    const size_t lds = FWD_LDS;
    if (int rc = set_lds(mlp_bwd_dz_kernel, lds)) return rc;
    BwdArgs b{};
    b.params = params; b.packed = packed; b.lay = *lay; b.action = action; b.mean = mean; b.acts = acts;
    b.d_logprob = d_logprob; b.d_entropy = d_entropy; b.d_value = d_value; b.M = m; b.dz = dz; b.dmu = dmu; b.dls = dls;
    hipLaunchKernelGGL(mlp_bwd_dz_kernel, dim3(ceil_div(m, BM), 2), dim3(256), lds, s, b);
    RLX_LAUNCH_CHECK();
    const int rows_per_slab = round_up((int)((m + slabs - 1) / slabs), 32);
    DwArgs d{};
    d.lay = *lay; d.states = states; d.acts = acts; d.dz = dz; d.M = m; d.rows_per_slab = rows_per_slab; d.grads = grads;
    hipLaunchKernelGGL(mlp_bwd_dw_kernel, dim3(slabs, 4, 6), dim3(256), 0, s, d);
    RLX_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_head_dw_kernel, dim3(slabs, 2), dim3(256), 0, s, *lay, acts, dmu, dls, d_value, (long long)m,
                       rows_per_slab, grads);
    RLX_LAUNCH_CHECK();
    return RLX_OK;
}
