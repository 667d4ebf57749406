"""Extra objects of bench.py's JSON line (round 5; everything here is measured inside the bench run itself, so the driver's record
carries it): `scaling_model`, `in_loop_kernels`, the normalised / masked roofline rows of the graded kernel.  Nothing here touches
oracle/."""

from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 peak (no sparsity)
HID = 256


def gae_normalised_rows(device, iters: int = 30) -> list:
    """The reference's DEFAULT advantage path at the HBM-sized shape (65536 envs x 128 steps): scan + safe_normalize
    (algorithms/advantages.py:24-86 + utils.py:397-404: 25 B per element -- the scan's 17 plus one more read and write of the
    advantages) and the same with a loss mask (26 B).  Same method as `roofline`: HIP events around back-to-back calls on the
    launch stream, rotating over buffer sets larger than the Infinity Cache; a "launch" here is the scan AND the standardize
    kernel behind it (two launches; the scan's epilogue leaves the moments)."""
    from rlinf_amd import ops
    T, B, nbuf = 128, 65536, 5
    g = torch.Generator().manual_seed(0)
    bufs = []
    for _ in range(nbuf):
        r = torch.rand(T, B, 1, generator=g).to(device)
        v = torch.randn(T + 1, B, 1, generator=g).to(device)
        d = (torch.rand(T + 1, B, 1, generator=g) < 0.02).to(device)
        d[0] = False
        m, _ = ops.done_prefix_mask(d)
        bufs.append((r, v, d, m, torch.empty_like(r), torch.empty_like(r)))
    rows = []
    for label, masked, per_elem in (("gae_scan + standardize (normalize_advantages: the reference default)", False, 25),
                                    ("gae_scan + standardize with a loss mask", True, 26)):
        def launch(i):
            r, v, d, m, a, q = bufs[i % nbuf]
            ops.gae_scan(r, v, d, m if masked else None, 0.99, 0.95, normalize_advantages=True, out=(a, q))
        for i in range(5):
            launch(i)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            launch(i)
        e1.record()
        torch.cuda.synchronize(device)
        us = e0.elapsed_time(e1) * 1e3 / iters
        nb = per_elem * T * B
        rows.append({"kernel": label, "bound": "hbm", "achieved": round(nb / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_us_per_call": round(us, 2), "algorithmic_bytes": nb,
                     "bytes_per_element": per_elem, "shape": f"{B} envs x {T} steps", "launches_per_call": 2})
    del bufs
    torch.cuda.empty_cache()
    return rows


# ---- per-kernel algorithmic work of the five hot launches of the loop --------------------------------------------------------
def _flops_per_row(obs: int, act: int, val: int = 1):
    """(forward, backward-data, weight-gradient) FLOPs per sample row, both networks (571 904 forward)."""
    fwd = bwd = dw = 0
    for o in (val, act):
        fwd += 2 * (obs * HID + 2 * HID * HID + HID * o)
        bwd += 2 * (HID * o + 2 * HID * HID)
        dw += 2 * (obs * HID + 2 * HID * HID + HID * o)
    return fwd, bwd, dw


def in_loop_kernels(precision: str, *, rows: int, envs: int, slabs: int, n_params: int, obs: int = 42, act: int = 8,
                    timeout_s: float = 240.0):
    """The hot launches of the timed loop, from a rocprofv3 kernel trace of a short run of THIS bench command (3 iterations, taken by
    bench.py itself so the driver's record holds it): median duration, achieved TFLOP/s or GB/s from the kernel's algorithmic work,
    fraction of the bound that applies (dense bf16 MFMA peak 2.5 PF for the three matrix kernels -- f32 mode is priced against the
    same pipe, its products run there as bf16 splits --, 8 TB/s HBM for the two streaming ones)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    with tempfile.TemporaryDirectory(prefix="rlx_ilk_", dir=env["TMPDIR"]) as wd:
        cmd = [exe, "--kernel-trace", "-d", wd, "-o", "ilk", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3",
               "--warmup", "2", "--precision", precision, "--no-cpu-baseline", "--no-roofline", "--no-variants", "--no-extras"]
        subprocess.run(cmd, env=env, cwd=wd, check=True, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(wd) for f in fs if f.endswith(".db")]
        if not dbs:
            return None
        db = sqlite3.connect(dbs[0])
        rws = db.execute("select name, duration from kernels").fetchall()
    fwd, bwd, dw = _flops_per_row(obs, act)
    value_fwd = 2 * (obs * HID + 2 * HID * HID + HID * 1)  # the bootstrap-value job rides in the same grid: value network only
    want = (("rollout step", ("rollout_step",), "mfma", (fwd + value_fwd) * envs, f"{envs} policy rows (both networks) + {envs} bootstrap-value rows"),
            ("fused forward + loss + backward-data", ("ppo_step_fused",), "mfma", (fwd + bwd) * rows, f"{rows} rows"),
            ("weight gradients", ("ppo_step_dw",), "mfma", dw * rows, f"{rows} rows, {slabs} split-K slabs"),
            # the optimizer step is ONE launch (rlx_adamw_params.sync_words); the two-launch form's kernels when that is switched off
            ("slab sum + norm + clip + AdamW (+ weight-image refresh), one launch", ("reduce_clip_adamw_one_launch",), "hbm",
             (slabs * 4 + 28) * n_params, f"{slabs} slabs x {n_params} f32 + 28 B per parameter"),
            ("slab sum + squared norm", ("grad_reduce_sqnorm",), "hbm", (slabs + 1) * n_params * 4, f"{slabs} slabs x {n_params} f32 -> 1"),
            ("clip + AdamW (+ weight-tile refresh)", ("clip_adamw_kernel",), "hbm", 28 * n_params, "28 B per parameter"))
    out = []
    for label, keys, bound, work, what in want:
        d = sorted(dur / 1e3 for name, dur in rws if any(k in name for k in keys))
        if not d:
            continue
        med = d[len(d) // 2]
        if bound == "mfma":
            ach = work / med / 1e6  # TFLOP/s
            out.append({"launch": label, "calls": len(d), "median_us": round(med, 2), "bound": "mfma", "achieved": round(ach, 1),
                        "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TF, "frac": round(ach / MFMA_BF16_PEAK_TF, 4),
                        "algorithmic_flops": int(work), "of": what})
        else:
            ach = work / med / 1e3  # GB/s
            out.append({"launch": label, "calls": len(d), "median_us": round(med, 2), "bound": "hbm", "achieved": round(ach, 1),
                        "unit": "GB/s", "peak": HBM_PEAK_GBPS, "frac": round(ach / HBM_PEAK_GBPS, 4), "algorithmic_bytes": int(work),
                        "of": what})
    step = [r for r in out if r["launch"] != "rollout step"]
    return {"source": "rocprofv3 --kernel-trace over `bench.py --steps 3 --warmup 2` spawned by this run (medians over all launches)",
            "optimizer_step_us": round(sum(r["median_us"] for r in step), 2) if len(step) in (3, 4) else None, "kernels": out,
            "note": "the matrix kernels are latency-bound chains on small tiles (SURVEY.md 8d: graded on time, not on MFMA fraction); "
                    "the fractions are here so the record states them"}


# ---- scaling model: measured per-rank compute share + measured exchange launch cost --------------------------------------------
def per_rank_share_ms(device, precision: str, share: int, steps: int = 8) -> float:
    """ONE GPU running what a rank of a `share`-GPU strong-scaling job runs (1024 / share envs, 8192 / share minibatch rows, no
    exchange): ms per iteration of the run-ahead loop."""
    import bench
    from rlinf_amd.scheduler import DistContext
    ctx = DistContext(0, 0, 1, torch.device(device))
    runner = bench.build_runner(bench.build_cfg(1, True, precision, total_envs=bench.ENVS // share, global_batch=bench.GLOBAL_BATCH // share), ctx)
    try:
        for _ in range(4):
            runner.run_step()
        torch.cuda.synchronize(device)
        # the FASTEST of three windows: a window that catches a clock ramp or a host hiccup (one 12-step window read 10.3 ms for the
        # 1/2 share on a box where the others read 6.6) must not become the model's per-rank share
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            pending = None
            for _ in range(steps):
                step = runner.run_step(defer=True)
                if pending is not None:
                    pending.result()
                pending = step
            pending.result()
            torch.cuda.synchronize(device)
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        return best
    finally:
        runner.close()
        del runner
        torch.cuda.empty_cache()


def scaling_model(device, precision: str, n_params: int, slabs_by_share: dict | None = None, opt_steps: int = 128) -> dict:
    """What a SCALE record should be read against.  Per N in (1, 2, 4, 8): the per-rank compute share measured on this GPU (above) and
    the exchange's own launches measured on this GPU (tools/exchange_self.py: peers aliased to the device, so link latency and peer
    skew are NOT in it -- a lower bound) => predicted strong- and weak-scaling iteration times.  Strong scaling of this loop is
    bounded by the per-launch latency floor: a rank's share of the rows costs almost as much as all of them."""
    from tools.exchange_self import measure as exchange_self
    shares = {}
    for w in (1, 2, 4, 8):
        shares[w] = round(per_rank_share_ms(device, precision, w), 3)
    ex = exchange_self(device, n_params, (slabs_by_share or {}).get(1, 10), steps=300)
    strong, weak = {}, {}
    for w in (1, 2, 4, 8):
        extra_ms = 0.0 if w == 1 else max(0.0, ex[f"w{w}"]["extra_us_per_step"]) * opt_steps / 1e3
        s_ms, w_ms = shares[w] + extra_ms, shares[1] + extra_ms
        strong[str(w)] = {"ms_per_iteration": round(s_ms, 3), "speedup_vs_1": round(shares[1] / s_ms, 3)}
        weak[str(w)] = {"ms_per_iteration": round(w_ms, 3), "throughput_x_vs_1": round(w * shares[1] / w_ms, 3),
                        "efficiency": round(shares[1] / w_ms, 3)}
    return {"per_rank_share_ms_no_exchange": {str(k): v for k, v in shares.items()},
            "exchange_launches_on_one_device": ex, "optimizer_steps_per_iteration": opt_steps,
            "predicted_strong": strong, "predicted_weak": weak,
            "reading": "predicted = measured per-rank share + 128 x the measured extra of the exchange (inside the optimizer launch: "
                       "pushed self-validating words) per optimizer step; a LOWER bound of the real iteration time (no link latency, no "
                       "peer skew, no RCCL).  Strong scaling at N = 8 is expected near "
                       f"{strong['8']['speedup_vs_1']} x (north_star asks >= 6 x: that assumes a throughput-bound step; this loop is "
                       "bound by the latency of three dependent launches per optimizer step -- an eighth of the rows costs three quarters "
                       f"of the time), weak scaling near {weak['8']['throughput_x_vs_1']} x at N = 8."}
