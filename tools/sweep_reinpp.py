"""Dev tool: rlx_reinpp_seq_adv, the tile walk (RLX_REINPP_REG=0) against reinpp_returns_reg_kernel with 4 / 8 groups per wave on
seven shapes (profiles/r06_reinpp_register_kernel.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import token_ops as T
dev = torch.device("cuda:0")
def avg_us(fn, iters=20, warm=4):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
g = torch.Generator(device=dev).manual_seed(1)
for bsz, seq in [(4096, 8192), (32768, 1024), (2048, 8192), (8192, 2048), (512, 16384), (64, 32768), (512, 2048)]:
    r = torch.randn(bsz, device=dev, generator=g)
    lp = -torch.rand(bsz, seq, device=dev, generator=g); rlp = lp + 0.1 * torch.randn(bsz, seq, device=dev, generator=g)
    msk = torch.rand(bsz, seq, device=dev, generator=g) < 0.8
    msk[:, 0] = True
    nb = bsz * seq * 21
    base = None
    for mode in ["0", "4", "8", "0", "4"]:
        os.environ["RLX_REINPP_REG"] = mode
        us = avg_us(lambda: T.reinpp_seq_adv(r, msk, lp, rlp, 0.001, "low_var_kl"))
        out = T.reinpp_seq_adv(r, msk, lp, rlp, 0.001, "low_var_kl")
        if base is None: base = out.clone()
        print(f"{bsz}x{seq} mode {mode}: {us:8.1f} us  {nb/us/1e6:6.2f} TB/s frac {nb/us/1e6/8:.3f} maxdiff {float((out-base).abs().max()):.2e}", flush=True)
