"""Dev tool: a few launches of rlx_ppo_step (M from argv) for rocprofv3 counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import ops
from rlinf_amd._lib import PPO_OUT_FLOATS
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
torch.manual_seed(0)
pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16 if BF16 else torch.float32).to("cuda")
lay = pol.layout
g = torch.Generator().manual_seed(1)
mb = dict(states=torch.randn(M, 42, generator=g), action=torch.randn(M, 8, generator=g) * 0.6,
          prev_logprobs=torch.randn(M, 8, generator=g) * 0.1 - 1.0, advantages=torch.randn(M, 1, generator=g),
          prev_values=torch.randn(M, 1, generator=g), returns=torch.randn(M, 1, generator=g))
mb = {k: v.cuda() for k, v in mb.items()}
lp = ops.make_ppo_params(logprob_type="action_level", action_dim=8, chunks=1, clip_ratio_low=0.2, clip_ratio_high=0.2,
                         value_clip=1.0, huber_delta=10.0, max_episode_steps=50, has_critic=True)
grads = torch.empty((ops.ppo_step_slabs(lay, M, bf16=BF16), lay.n_params), device="cuda")
ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
m_, v_ = torch.zeros(lay.n_params, device="cuda"), torch.zeros(lay.n_params, device="cuda")
for _ in range(5):
    ops.ppo_step(pol.flat.data, lay, lp, mb, grads, row, ws, grad_out=1.0, bf16=BF16, tiles=pol.tiles())
    ops.clip_adamw_step_(pol.flat.data, grads, m_, v_, pol.group_ranges(3e-4, 3e-4), 1, max_grad_norm=0.5,
                         tile_layout=lay, tiles=pol.tiles())
torch.cuda.synchronize()
states, eps = torch.randn(1024, 42, device="cuda"), torch.randn(1024, 8, device="cuda")
for _ in range(5):
    ops.mlp_rollout_step(pol.flat.data, pol.tiles(), lay, states, eps)
torch.cuda.synchronize()
