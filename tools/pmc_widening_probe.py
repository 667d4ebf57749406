"""Workload for the --pmc passes over the widening kernels: a device-to-device copy of known size (calibration for the
16-byte-per-lane access width), then token_logprob fwd / bwd, patch_scan, gae_seq, the three reinpp launches and copy_segments, each a few launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import _lib, token_ops
from rlinf_amd.hybrid_engines.weight_syncer.patch_syncer import _dtype_code

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
N, V = 2048, 151936
x = (torch.randn(N, V, device=dev, generator=g) * 4).to(torch.bfloat16)
y = torch.empty_like(x)
labels = torch.randint(0, V, (N,), device=dev, generator=g)
dlp = torch.randn(N, device=dev, generator=g)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)                       # calibration: reads and writes N*V*2 bytes
torch.cuda.synchronize()
for _ in range(3):
    lp, _, lse = token_ops.token_logprob_fwd(x, labels)
torch.cuda.synchronize()
for _ in range(3):
    token_ops.token_logprob_bwd(x, labels, lse, None, dlp, None, out=y)
torch.cuda.synchronize()
lib = _lib.load()
n = x.numel()
code = _dtype_code(torch.bfloat16)
wsb = lib.rlx_patch_workspace_bytes(n)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
nnz = torch.zeros(1, dtype=torch.int64, device=dev)
for _ in range(3):
    lib.rlx_patch_scan(x.data_ptr(), code, y.data_ptr(), code, n, ws.data_ptr(), wsb, nnz.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
v = torch.randn(4096, 8192, device=dev, generator=g)
r = torch.randn(4096, device=dev, generator=g)
for _ in range(3):
    token_ops.gae_seq(v, r, 1.0, 0.95)
torch.cuda.synchronize()
lp2 = -torch.rand(4096, 8192, device=dev, generator=g) * 3
rlp2 = lp2 + 0.3 * torch.randn(4096, 8192, device=dev, generator=g)
msk = torch.ones(4096, 8192, dtype=torch.bool, device=dev)
for _ in range(3):
    token_ops.reinpp_seq_adv(r, msk, lp2, rlp2, 0.001, "low_var_kl")   # reinpp_returns / _reduce / _normalize
torch.cuda.synchronize()
from rlinf_amd.hybrid_engines.weight_syncer.bucket_syncer import BucketPacker
masters = [(f"w{i}", torch.randn(4096, 8192, device=dev, generator=g), torch.bfloat16) for i in range(8)]
packer = BucketPacker(masters)
for _ in range(3):
    packer.pack(masters, torch.device(dev), None, persistent=True)      # copy_segments: 8 x 128 MiB f32 -> bf16
torch.cuda.synchronize()
print("bytes", N * V * 2, "gae_seq tokens", v.numel())
