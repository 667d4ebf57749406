"""Dev tool: time the fused launches (HIP events over back-to-back launches) for the development variants.
python tools/bench_step.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import ops
from rlinf_amd._lib import PPO_OUT_FLOATS
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
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
    row = torch.zeros(PPO_OUT_FLOATS, device="cuda")
    print(f"ppo_step (pack + fused + dw launches) at M={M}")
    for prec in ("bf16", "f32"):
        for slabs in (24, 16, 12, 8):
            os.environ["RLX_DW_SLABS"] = str(slabs)
            grads = torch.empty((ops.ppo_step_slabs(lay, M, bf16=prec == "bf16"), lay.n_params), device="cuda")
            ws = torch.empty(ops.ppo_step_workspace_bytes(lay, M), dtype=torch.uint8, device="cuda")
            m_, v_ = torch.zeros(lay.n_params, device="cuda"), torch.zeros(lay.n_params, device="cuda")
            p2 = pol.flat.data.clone()
            def step():
                ops.ppo_step(p2, lay, lp, mb, grads, row, ws, grad_out=1.0, bf16=prec == "bf16")
                ops.clip_adamw_step_(p2, grads, m_, v_, pol.group_ranges(3e-4, 3e-4), 1, max_grad_norm=0.5)
            t = timeit(step)
            print(f"  {prec} slabs={grads.shape[0]:3d}: {t:8.2f} us per optimizer step (pack + fused + dw + sqnorm + adamw, eager)")
    os.environ.pop("RLX_DW_SLABS", None)
    os.environ["RLX_STEP_VARIANT"] = "0"
    B = 1024
    states, eps = torch.randn(B, 42, device="cuda"), torch.randn(B, 8, device="cuda")
    fin = torch.randn(B, 42, device="cuda")
    rew, flags = torch.rand(B, 1, device="cuda"), torch.rand(B, 1, device="cuda") < 0.3
    out = (torch.empty(B, 8, device="cuda"), torch.empty(B, 8, device="cuda"), torch.empty(B, 1, device="cuda"))
    for rep in range(2):
        for pd in (2,):
            os.environ["RLX_ROLLOUT_PD"] = str(pd)
            t = timeit(lambda: ops.mlp_rollout_step(pol.flat.data, pol.tiles(), lay, states, eps, out=out,
                                                    value_jobs=(dict(states=fin, rewards=rew, flags=flags, gamma=0.8),)), iters=50)
            t2 = timeit(lambda: ops.mlp_rollout_step(pol.flat.data, pol.tiles(), lay, states[:128], eps[:128], out=tuple(o[:128] for o in out)), iters=50)
            print(f"  rep{rep} rollout_step PD={pd}: {t:7.2f} us (1024 envs + bootstrap job)   {t2:7.2f} us (128 envs, policy only)")


if __name__ == "__main__":
    main()
