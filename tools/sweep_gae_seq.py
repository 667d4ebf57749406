"""Dev tool: rlx_gae_seq, the LDS kernels (RLX_GAESEQ_REG=0) against gae_seq_reg_kernel with ordinary (1) / non-temporal (2)
accesses on eleven shapes; HIP-event averages and the largest difference from mode 0 (profiles/r06_gae_seq_register_kernel.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import _lib, token_ops
lib = _lib.load()
dev = torch.device("cuda:0")
def avg_us(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for bsz, seq in [(4096, 8192), (32768, 1024), (2048, 8192), (1024, 8192), (512, 16384), (8192, 2048), (65536, 512), (64, 32768), (256, 16384), (32, 8192), (512, 2048)]:
    v = torch.randn(bsz, seq, device=dev); r = torch.randn(bsz, device=dev)
    adv, ret = torch.empty_like(v), torch.empty_like(v)
    ws = torch.empty(max(256, lib.rlx_gae_seq_workspace_bytes(bsz, seq)), dtype=torch.uint8, device=dev)
    nb = bsz * seq * 12
    base = None
    for mode in ["0", "1", "2"]:
        os.environ["RLX_GAESEQ_REG"] = mode
        f = lambda: lib.rlx_gae_seq(v.data_ptr(), r.data_ptr(), adv.data_ptr(), ret.data_ptr(), bsz, seq, 1.0, 0.95, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        us = avg_us(f)
        if mode == "0": base = (adv.clone(), ret.clone())
        err = float((ret - base[1]).abs().max())
        print(f"{bsz}x{seq} mode {mode}: {us:8.1f} us  {nb/us/1e6:6.2f} TB/s frac {nb/us/1e6/8:.3f} maxdiff {err:.2e}", flush=True)
