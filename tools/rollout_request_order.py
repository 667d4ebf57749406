"""Dev tool (development library: RLX_LIB_TAG=dev built with -DRLX_DEV_VARIANTS): the rollout launch under every compiled request
order of its weight fragments (RLX_ROLLOUT_ORDER), 64 steps per replayed hipGraph; run under `rocprofv3 --kernel-trace` and read the
per-instantiation medians with tools/rocpd_stats.py (profiles/r06_rollout_request_order.txt).  Asserts bit-identical outputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import ops
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy
torch.manual_seed(0)
pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16).to("cuda")
lay = pol.layout
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
states, eps = torch.randn(B, 42, device="cuda"), torch.randn(B, 8, device="cuda")
fin = torch.randn(B, 42, device="cuda"); rew = torch.rand(B, 1, device="cuda"); flags = torch.rand(B, 1, device="cuda") < 0.3
base = None
orders = ["0", "1345", "1135", "1134", "33", "1355", "2345"]
side = torch.cuda.Stream()
for rep in range(2):
    for order in orders:
        os.environ["RLX_ROLLOUT_ORDER"] = order
        r = rew.clone()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = ops.mlp_rollout_step(pol.flat.data, pol.tiles(), lay, states, eps, value_jobs=(dict(states=fin, rewards=r, flags=flags, gamma=0.8),))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(64):
                    out = ops.mlp_rollout_step(pol.flat.data, pol.tiles(), lay, states, eps, value_jobs=(dict(states=fin, rewards=r, flags=flags, gamma=0.8),))
            for _ in range(5):
                g.replay()
        torch.cuda.synchronize()
        if base is None: base = [o.clone() for o in out[:3]]
        assert all(torch.equal(a, b) for a, b in zip(out[:3], base)), order
print("bit-identical across orders")
