"""Roofline probe of gae_seq: [bsz, seq] reasoning GAE.  Algorithmic bytes = 12 B/token (values read, advantages and
returns written).  Also times the reference route (transposes + its Python loop over seq) on the same GPU with torch."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlinf_amd import token_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bsz", type=int, default=4096)
ap.add_argument("--seq", type=int, default=8192)
a = ap.parse_args()
dev = "cuda"
v = torch.randn(a.bsz, a.seq, device=dev)
r = torch.randn(a.bsz, device=dev)
for _ in range(3):
    token_ops.gae_seq(v, r, 1.0, 0.95)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for s, e in evs:
    s.record(); token_ops.gae_seq(v, r, 1.0, 0.95); e.record()
torch.cuda.synchronize()
t = sum(s.elapsed_time(e) for s, e in evs) / len(evs) * 1e-3
by = a.bsz * a.seq * 12
print(json.dumps({"kernel": "gae_seq", "bsz": a.bsz, "seq": a.seq, "bytes": by, "us": t * 1e6, "GBps": by / t / 1e9,
                  "frac": by / t / 8e12, "note": "times include two torch.empty allocations per call"}))
# the reference's route with torch ops on this GPU (restated inline: transposes + the sequential loop), small seq sample
seq_s = min(a.seq, 1024)
vt = torch.cat([v[:, :seq_s].transpose(0, 1), torch.zeros(1, a.bsz, device=dev)], 0)
rew = torch.zeros(seq_s, a.bsz, device=dev); rew[-1] = r
nd = torch.ones(seq_s + 1, a.bsz, device=dev, dtype=torch.bool); nd[-1] = False
torch.cuda.synchronize(); t0 = time.perf_counter()
gae = 0; ret = torch.zeros_like(rew)
for tt in reversed(range(seq_s)):
    delta = rew[tt] + 1.0 * vt[tt + 1] * nd[tt + 1] - vt[tt]
    gae = delta + 0.95 * nd[tt + 1] * gae
    ret[tt] = gae + vt[tt]
advr = (ret - vt[:-1]).transpose(0, 1).contiguous()
torch.cuda.synchronize(); tr = time.perf_counter() - t0
print(json.dumps({"kernel": "torch loop (reference route) on the same GPU", "seq_sample": seq_s, "seconds": tr,
                  "extrapolated_seconds_full_seq": tr * a.seq / seq_s, "speedup": tr * a.seq / seq_s / t}))
