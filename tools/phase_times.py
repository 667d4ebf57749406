"""Dev tool: shader-clock stamps at the phase boundaries of the two fused kernels (block 0, thread 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import ops, _lib
from rlinf_amd._lib import PPO_OUT_FLOATS
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy

raw = ctypes.CDLL(_lib.LIB_PATH)
raw.rlx_dev_set_timing_buffer.argtypes = [ctypes.c_void_p]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(0)
pol = MLPPolicy(42, 8, 1, True, False).to("cuda")
lay = pol.layout
g = torch.Generator().manual_seed(1)
mb = dict(states=torch.randn(M, 42, generator=g), action=torch.randn(M, 8, generator=g) * 0.6,
          prev_logprobs=torch.randn(M, 8, generator=g) * 0.1 - 1.0, advantages=torch.randn(M, 1, generator=g),
          prev_values=torch.randn(M, 1, generator=g), returns=torch.randn(M, 1, generator=g))
mb = {k: v.cuda() for k, v in mb.items()}
lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                         value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
grads = torch.empty((ops.ppo_step_slabs(lay, M), lay.n_params), device="cuda")
ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.ppo_step(pol.flat.data, lay, lp, mb, grads, row, ws, grad_out=1.0)
torch.cuda.synchronize()
raw.rlx_dev_set_timing_buffer(buf.data_ptr())
if os.environ.get("RLX_F32_EXACT_MFMA", "0") in ("", "0"):  # the default f32 launches: ppo_step_f32x.hip, block (0, 1)
    names = ["start", "inputs staged + states tiles", "gemm L1", "epi1 (tanh, split, slab + images)", "gemm L2", "epi2", "gemm L3", "epi3",
             "head (mfma)", "loss pass a", "loss pass b", "loss pass c + metric sums", "head grads", "dz3 (mfma f32) + emit", "bwd gemm W3",
             "emit dz2", "bwd gemm W2", "emit dz1", "stores drained"]
else:
    names = ["start", "states+head staged", "gemm L1", "epi1+flush", "gemm L2", "epi2+flush", "gemm L3", "epi3", "head+loss+reduce",
             "head grads", "dz3+flush", "bwd gemm W3 + epi + flush", "bwd gemm W2 + epi + flush", "end"]
for rep in range(3):
    buf.zero_()
    ops.ppo_step(pol.flat.data, lay, lp, mb, grads, row, ws, grad_out=1.0)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    n = len(names)
    print(f"ppo_step_fused M={M} block (0,0): total {t[n-1]-t[0]} ticks")
    print("   " + "  ".join(f"{names[i]}: {t[i]-t[i-1]}" for i in range(1, n)))
# bf16 fused kernel (block (0,0)) and weight-gradient kernel (LDS-DMA variant): stamps of the workgroup owning GEMM item 0 land at buf[32:]
pol16 = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16).to("cuda")
grads16 = torch.empty((ops.ppo_step_slabs(lay, M, bf16=True), lay.n_params), device="cuda")
names16 = ["start", "loads issued", "states arrived", "inputs staged", "gemm L1", "epi1+tiles", "gemm L2", "epi2+tiles", "gemm L3", "epi3", "head (mfma)",
           "element pass", "loss pass (wave 0)", "dOut pass + metric sums", "head grads", "dz3+tiles", "bwd gemm W3 + epi + tiles",
           "bwd gemm W2 + epi + tiles", "stores drained"]  # ("loss pass" is stamped by the policy network's workgroup only)
for rep in range(3):
    buf.zero_()
    ops.ppo_step(pol16.flat.data, lay, lp, mb, grads16, row, ws, grad_out=1.0, bf16=True)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    n = len(names16)
    print(f"ppo_step_fused_bf16 (32 rows per workgroup) M={M} block (0,1): total {t[n-1]-t[0]} ticks")
    print("   " + "  ".join(f"{names16[i]}: {t[i]-t[i-1]}" for i in range(1, n)))
    t = t[32:48]
    print(f"dw bf16 (item 0): prologue->first barrier {t[1]-t[0]}  per k-block {[t[i+1]-t[i] for i in range(1, 11)]}  loop end {t[14]-t[0]}  stores drained {t[15]-t[14]}")
states, eps = torch.randn(1024, 42, device="cuda"), torch.randn(1024, 8, device="cuda")
rn = ["start", "kernargs arrived", "loads issued", "states arrived", "inputs staged", "gemm L1", "epi1", "gemm L2", "epi2", "gemm L3", "epi3", "head"]
for B in (1024, 128):
    for rep in range(2):
        buf.zero_()
        ops.mlp_rollout_step(pol16.flat.data, pol16.tiles(), lay, states[:B], eps[:B])
        torch.cuda.synchronize()
        t = buf.cpu().tolist()
        print(f"rollout_step B={B} block 0: total {t[len(rn)-1]-t[0]} ticks")
        print("   " + "  ".join(f"{rn[i]}: {t[i]-t[i-1]}" for i in range(1, len(rn))))
raw.rlx_dev_set_timing_buffer(None)
