"""Times of the mixed read/write HBM kernels of the widening rows for the library RLX_LIB_TAG selects (variant sweeps):

    RLX_LIB_TAG=nt python tools/bench_widening.py

gae_seq (12 B/token), reinpp_seq_adv (21 B/token, three launches), copy_segments f32 -> bf16 (6 B/element)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import _lib, token_ops  # noqa: E402
from rlinf_amd.hybrid_engines.weight_syncer.bucket_syncer import BucketPacker  # noqa: E402


def avg_us(fn, iters=20):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2]


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    tag = os.environ.get("RLX_LIB_TAG", "") or "default"
    out = []
    for bsz, seq in ((4096, 8192), (16384, 2048), (32768, 1024)):
        v = torch.randn(bsz, seq, device=dev, generator=g)
        r = torch.randn(bsz, device=dev, generator=g)
        # caller-owned outputs and workspace: nothing is allocated inside the timed calls (two torch.empty per call moved the
        # round-1 numbers by 10 % from run to run)
        outs = (torch.empty_like(v), torch.empty_like(v))
        gws = torch.empty(_lib.load().rlx_gae_seq_workspace_bytes(bsz, seq), dtype=torch.uint8, device=dev)
        mean, med = avg_us(lambda: token_ops.gae_seq(v, r, 1.0, 0.95, out=outs, workspace=gws))
        nb = v.numel() * 12
        out.append(dict(lib=tag, kernel="gae_seq" + ("(per-sequence walk)" if os.environ.get("RLX_GAESEQ_VARIANT") == "1" else ""), shape=f"{bsz}x{seq}", us=round(mean, 1), median_us=round(med, 1),
                        frac=round(nb / mean / 1e3 / 8000, 4)))
        lp = -torch.rand(bsz, seq, device=dev, generator=g) * 3
        rlp = lp + 0.3 * torch.randn(bsz, seq, device=dev, generator=g)
        msk = torch.ones(bsz, seq, dtype=torch.bool, device=dev)
        mean, med = avg_us(lambda: token_ops.reinpp_seq_adv(r, msk, lp, rlp, 0.001, "low_var_kl"))
        nb = v.numel() * 21
        out.append(dict(lib=tag, kernel="reinpp_seq_adv", shape=f"{bsz}x{seq}", us=round(mean, 1), median_us=round(med, 1),
                        frac=round(nb / mean / 1e3 / 8000, 4)))
        del v, lp, rlp, msk, outs, gws
    masters = [(f"w{i}", torch.randn(4096, 8192, device=dev, generator=g), torch.bfloat16) for i in range(16)]
    packer = BucketPacker(masters)
    mean, med = avg_us(lambda: packer.pack(masters, dev, None, persistent=True))
    nb = 16 * 4096 * 8192 * 6
    out.append(dict(lib=tag, kernel="copy_segments f32->bf16", shape="16x4096x8192", us=round(mean, 1), median_us=round(med, 1),
                    frac=round(nb / mean / 1e3 / 8000, 4)))
    for row in out:
        print(json.dumps(row))
    assert os.path.basename(_lib.LIB_PATH).startswith("librlx_hip")


if __name__ == "__main__":
    main()
