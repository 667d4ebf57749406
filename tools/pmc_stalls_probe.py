"""Workload for the stall-counter passes (tools/pmc_stalls.sh): the mixed read/write streams that sit at 0.62-0.65 of the HBM peak,
next to a plain device-to-device copy of the same byte count (the copy ceiling's own counters) and the read-only forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import token_ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
N, V = 2048, 151936
x = (torch.randn(N, V, device=dev, generator=g) * 4).to(torch.bfloat16)
y = torch.empty_like(x)
labels = torch.randint(0, V, (N,), device=dev, generator=g)
dlp = torch.randn(N, device=dev, generator=g)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
for _ in range(3):
    lp, _, lse = token_ops.token_logprob_fwd(x, labels)
torch.cuda.synchronize()
for _ in range(3):
    token_ops.token_logprob_bwd(x, labels, lse, None, dlp, None, out=y)
torch.cuda.synchronize()
v = torch.randn(4096, 8192, device=dev, generator=g)
r = torch.randn(4096, device=dev, generator=g)
for _ in range(3):
    token_ops.gae_seq(v, r, 1.0, 0.95)
torch.cuda.synchronize()
from rlinf_amd import ops
T, B = 128, 65536
rw, vv = torch.rand(T, B, 1, device=dev, generator=g), torch.randn(T + 1, B, 1, device=dev, generator=g)
dd = torch.rand(T + 1, B, 1, device=dev, generator=g) < 0.02
a, q = torch.empty_like(rw), torch.empty_like(rw)
for _ in range(3):
    ops.gae_scan(rw, vv, dd, None, 0.99, 0.95, normalize_advantages=False, out=(a, q))
torch.cuda.synchronize()
