"""Roofline probes of bench.py (moved out of it in round 4: bench.py keeps the contract -- configuration, timed region, N > 1
launch, the CPU-baseline legs -- and imports these): the graded gae_scan launch timed with HIP events at the HBM-sized shape with
its PMC traffic passes, and the kernels of the widening rows timed the same way.  Nothing here touches oracle/."""

from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS, HORIZON = 1024, 128            # BASELINE.json configs[1] (bench.py)
GAMMA, LAMBDA = 0.8, 0.9             # examples/embodiment/config/maniskill_ppo_mlp.yaml:63-64
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _lib_handle():
    from rlinf_amd import _lib
    return _lib.load()


def _pmc_pass(counter: str, workdir: str):
    """One rocprofv3 counter pass (own run, --kernel-trace only: the gpurun rules) over tools/gae_traffic_probe.py ->
    {kernel-name-prefix: mean counter value per launch} or None when rocprofv3 is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    out = os.path.join(workdir, counter)
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "pmc", "-f", "csv", "--", sys.executable,
                        os.path.join(ROOT, "tools", "gae_traffic_probe.py")], env=env, cwd=workdir, check=True, timeout=240,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return None
    acc = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = "copy" if "dev_stream_copy" in name else ("gae" if "gae_scan" in name else None)
            if key:
                tot, n = acc.get(key, (0.0, 0))
                acc[key] = (tot + float(row["Counter_Value"]), n + 1)
    return {k: t / n for k, (t, n) in acc.items()} if acc else None


def gae_traffic(algo_read: int, algo_write: int):
    """HBM bytes per gae_scan launch from the TCC_EA-derived counters, one --pmc pass per counter, calibrated as the
    microarch guide prescribes: FETCH_SIZE / WRITE_SIZE are in KiB-class units whose scale depends on the access width on
    gfx950, so each is scaled by (known bytes / counter) of a copy kernel with gae_scan's own access pattern (one dword
    per lane, time-major rows)."""
    import tempfile
    cal = 4 * HORIZON * 65536  # the calibration copy reads and writes exactly this many bytes per launch
    with tempfile.TemporaryDirectory(prefix="rlx_pmc_", dir=os.environ.get("TMPDIR", "/tmp")) as wd:
        fetch = _pmc_pass("FETCH_SIZE", wd)
        write = _pmc_pass("WRITE_SIZE", wd)
    if not fetch or not write or "copy" not in fetch or "gae" not in fetch or "copy" not in write or "gae" not in write:
        return None, None
    rd = fetch["gae"] * (cal / fetch["copy"])
    wr = write["gae"] * (cal / write["copy"])
    detail = {"read_bytes": round(rd), "write_bytes": round(wr), "raw_FETCH_SIZE": round(fetch["gae"], 1),
              "raw_WRITE_SIZE": round(write["gae"], 1), "calibration": "dword-per-lane copy of 33.5 MB: "
              f"FETCH_SIZE {fetch['copy']:.1f}, WRITE_SIZE {write['copy']:.1f} per launch",
              "algorithmic_read_bytes": algo_read, "algorithmic_write_bytes": algo_write}
    return round(rd + wr), detail


def gae_roofline(device, iters: int = 30, with_traffic: bool = True):
    """gae_scan (un-normalised, the variant the auto heuristic picks) on 65536 x 128, rotating over buffer sets larger than
    the 256 MiB Infinity Cache so every launch streams from HBM.  `achieved` = algorithmic bytes / average launch duration,
    the average taken with HIP events around `iters` back-to-back launches on the launch stream (per-launch event pairs
    add ~2 us of event overhead to a 28 us kernel; the per-launch numbers are reported next to it)."""
    from rlinf_amd import ops
    T, B, nbuf = HORIZON, 65536, 5
    g = torch.Generator().manual_seed(0)
    bufs = []
    for _ in range(nbuf):
        r = torch.rand(T, B, 1, generator=g).to(device)
        v = torch.randn(T + 1, B, 1, generator=g).to(device)
        d = (torch.rand(T + 1, B, 1, generator=g) < 0.02).to(device)
        bufs.append((r, v, d, torch.empty_like(r), torch.empty_like(r)))

    def launch(i):
        r, v, d, a, q = bufs[i % nbuf]
        ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=False, out=(a, q))

    for i in range(5):
        launch(i)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()  # torch's current stream == the stream the C ABI launches on
    for i in range(iters):
        launch(i)
    e1.record()
    torch.cuda.synchronize(device)
    avg_us = e0.elapsed_time(e1) * 1e3 / iters
    evs = []
    for i in range(iters):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        launch(i)
        a1.record()
        evs.append((a0, a1))
    torch.cuda.synchronize(device)
    single = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
    algo_read, algo_write = 9 * T * B + 5 * B, 8 * T * B  # r 4 + V 4 + done 1 (+ the extra V / done row); adv 4 + ret 4
    algo_bytes = 17 * T * B
    achieved = algo_bytes / avg_us / 1e3  # GB/s
    # contract shape, for reference (launch/latency bound: 2.2 MB)
    r, v, d = (t[:, :ENVS].contiguous() for t in bufs[0][:3])
    a, q = torch.empty_like(r), torch.empty_like(r)
    for _ in range(3):
        ops.gae_scan(r, v, d, None, GAMMA, LAMBDA, out=(a, q))
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(20):
        ops.gae_scan(r, v, d, None, GAMMA, LAMBDA, out=(a, q))
    c1.record()
    torch.cuda.synchronize(device)
    out = {"bound": "hbm", "kernel": "gae_scan_c1<1,1,64,nt> (streaming scan, 65536 envs x 128 steps: the HBM-sized buffer SURVEY.md 8d "
                                      "prescribes for the roofline; at the 1024 x 128 contract shape the loop runs gae_scan_c1<1,8,8> + "
                                      "standardize_kernel out of L2, see contract_shape_us_per_call_incl_normalise)",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
           "traffic": None, "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": round(avg_us, 2),
           "launches_timed": iters, "single_launch_event_pair_us": {"median": round(single[len(single) // 2], 2), "min": round(single[0], 2)},
           "frac_of_measured_copy_ceiling_6.29TBps": round(achieved / 6290.0, 4),
           "contract_shape_us_per_call_incl_normalise": round(c0.elapsed_time(c1) * 1e3 / 20, 2)}
    del bufs
    torch.cuda.empty_cache()
    if with_traffic:
        traffic, detail = gae_traffic(algo_read, algo_write)
        out["traffic"] = traffic
        if detail:
            out["traffic_detail"] = detail
    return out


def token_tier_roofline(device, tokens: int = 4096, vocab: int = 151936, iters: int = 10, with_codec: bool = False):
    """SURVEY.md 8f item 1 (the widening after rows a-e): the reasoning learner's logits -> log-prob/entropy kernel and
    its backward at a Qwen-size vocabulary, bf16 logits, timed with HIP events on the launch stream.  Algorithmic bytes:
    forward = tokens * vocab * 2 (one read); backward = twice that (one read + one write)."""
    from rlinf_amd import token_ops

    g = torch.Generator(device=device).manual_seed(1)
    x = torch.empty(tokens, vocab, dtype=torch.bfloat16, device=device)
    for i in range(0, tokens, 1024):
        x[i:i + 1024] = (torch.randn(min(1024, tokens - i), vocab, device=device, generator=g) * 4).to(torch.bfloat16)
    labels = torch.randint(0, vocab, (tokens,), device=device, generator=g)
    dlp = torch.randn(tokens, device=device, generator=g)
    out = torch.empty_like(x)
    _, _, lse = token_ops.token_logprob_fwd(x, labels)

    def avg_us(fn):
        for _ in range(2):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize(device)
        return sum(a.elapsed_time(b) for a, b in evs) / iters * 1e3

    rows = []
    by = tokens * vocab * 2
    for name, fn, nbytes in (("token_logprob_fwd", lambda: token_ops.token_logprob_fwd(x, labels), by),
                             ("token_logprob_bwd", lambda: token_ops.token_logprob_bwd(x, labels, lse, None, dlp, None, out=out),
                              2 * by)):
        us = avg_us(fn)
        gbps = nbytes / us / 1e3
        rows.append({"kernel": name, "bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(gbps / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1), "algorithmic_bytes": nbytes,
                     "shape": f"{tokens} tokens x {vocab} vocab, bf16 logits"})
    # the packed form (round 5): the same pass with the reference's unpack fused into its stores -- rows without a destination
    # (prompt rows outside the response window, eos fill) are not read, so the algorithmic bytes are those of the rows that are
    try:
        from rlinf_amd.hybrid_engines.fsdp.utils import unpack_index_maps
        seq, prompt = 2048, 256
        starts = [0] * (tokens // seq)   # (valid columns of every sequence in the unpacked [bsz, seq] matrix: all of them)
        ends = [seq] * (tokens // seq)
        lp_dst, ent_dst = unpack_index_maps(starts, ends, tokens, seq, seq - prompt, device)
        read_rows = int(((lp_dst >= 0) | (ent_dst >= 0)).sum())
        with torch.no_grad():
            us = avg_us(lambda: token_ops._PackedLogprobFn.apply(x, labels, lp_dst, ent_dst, len(starts), seq - prompt, 1.0, True, False, False))
        nb = read_rows * vocab * 2
        rows.append({"kernel": "token_logprob_fwd_packed (+ entropy, unpack fused)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1),
                     "algorithmic_bytes": nb, "shape": f"{len(starts)} packed sequences x ({prompt} prompt + {seq - prompt} response) tokens x "
                     f"{vocab} vocab, bf16 logits: {read_rows} of {tokens} rows have a destination (host-side output allocation included)"})
    except Exception as e:  # noqa: BLE001 -- a probe must not cost the rows above
        rows.append({"kernel": "token_logprob_fwd_packed", "error": f"{type(e).__name__}: {e}"[:200]})
    # reasoning GAE along the contiguous axis (12 B/token) and the weight-patch scan (tensor + snapshot read once)
    v = torch.randn(4096, 8192, device=device, generator=g)
    r = torch.randn(4096, device=device, generator=g)
    adv, ret = torch.empty_like(v), torch.empty_like(v)
    lib = _lib_handle()
    st = torch.cuda.current_stream(device).cuda_stream
    gws = torch.empty(lib.rlx_gae_seq_workspace_bytes(4096, 8192), dtype=torch.uint8, device=device)
    us = avg_us(lambda: lib.rlx_gae_seq(v.data_ptr(), r.data_ptr(), adv.data_ptr(), ret.data_ptr(), 4096, 8192, 1.0, 0.95,
                                        gws.data_ptr(), gws.numel(), st))
    nb = v.numel() * 12
    rows.append({"kernel": "gae_seq", "bound": "hbm", "achieved": round(nb / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1), "algorithmic_bytes": nb,
                 "shape": "4096 sequences x 8192 tokens, f32 values"})
    n = x.numel()
    wsb = lib.rlx_patch_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=device)
    nnz = torch.zeros(1, dtype=torch.int64, device=device)
    us = avg_us(lambda: lib.rlx_patch_scan(x.data_ptr(), 1, out.data_ptr(), 1, n, ws.data_ptr(), wsb, nnz.data_ptr(), st))
    nb = 2 * n * 2
    rows.append({"kernel": "patch_scan (+ offsets)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1), "peak": HBM_PEAK_GBPS,
                 "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1),
                 "algorithmic_bytes": nb, "shape": f"{n} bf16 elements against their snapshot"})
    # Reinforce++ advantages (returns pass 13 B/token + normalisation 8 B/token; three launches timed together)
    from rlinf_amd import token_ops as _t
    lp = -torch.rand(4096, 8192, device=device, generator=g) * 3
    rlp = lp + 0.3 * torch.randn(4096, 8192, device=device, generator=g)
    msk = torch.ones(4096, 8192, dtype=torch.bool, device=device)
    us = avg_us(lambda: _t.reinpp_seq_adv(r, msk, lp, rlp, 0.001, "low_var_kl"))
    nb = lp.numel() * 21
    rows.append({"kernel": "reinpp_seq_adv (returns + reduce + normalize)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1),
                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1),
                 "algorithmic_bytes": nb, "shape": "4096 sequences x 8192 tokens, k3 KL penalty"})
    # the same bytes as short rows (one 64-lane workgroup per sequence, four tiles each): the shape round 1 was furthest off at
    try:
        r2 = torch.randn(32768, device=device, generator=g)
        lp2, rlp2, msk2 = lp.view(32768, 1024), rlp.view(32768, 1024), msk.view(32768, 1024)
        us = avg_us(lambda: _t.reinpp_seq_adv(r2, msk2, lp2, rlp2, 0.001, "low_var_kl"))
        rows.append({"kernel": "reinpp_seq_adv (returns + reduce + normalize)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4),
                     "avg_launch_us": round(us, 1), "algorithmic_bytes": nb, "shape": "32768 sequences x 1024 tokens, k3 KL penalty"})
        del lp2, rlp2, msk2
    except Exception as e:  # noqa: BLE001 -- an extra row must never cost the others
        rows.append({"kernel": "reinpp_seq_adv", "shape": "32768 sequences x 1024 tokens", "error": f"{type(e).__name__}: {e}"[:200]})
    del lp, rlp, msk
    # bucket weight sync: 16 f32 masters of 4096 x 8192 -> one flat bf16 transport buffer (6 B per element, one launch)
    from rlinf_amd.hybrid_engines.weight_syncer.bucket_syncer import BucketPacker
    masters = [(f"w{i}", torch.randn(4096, 8192, device=device, generator=g), torch.bfloat16) for i in range(16)]
    packer = BucketPacker(masters)
    dev_t = torch.device(device)
    us = avg_us(lambda: packer.pack(masters, dev_t, None, persistent=True))
    nb = 16 * 4096 * 8192 * 6
    rows.append({"kernel": "copy_segments (bucket pack f32 -> bf16)", "bound": "hbm", "achieved": round(nb / us / 1e3, 1),
                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1),
                 "algorithmic_bytes": nb, "shape": "16 tensors x 4096 x 8192 f32 into one bf16 bucket"})
    del masters, packer
    # compressed patch transport (csrc/zplane_codec.hip): the two index streams of a sparse weight patch, 100 MB each.
    # Algorithmic bytes: compress reads the stream twice (measure, pack) and writes the compressed form; decompress reads the
    # compressed form and writes the stream.  NOT part of the default set since round 5 (bench.py --with-codec): `compression: none`
    # is the default and the only setting a reference peer can decode; the codec (this build's own "RLXZ" format, 0.35-0.60 of the
    # HBM peak: its compressor reads its input twice) is an opt-in for homogeneous deployments.
    if not with_codec:
        return rows
    try:
        from rlinf_amd import _lib
        from rlinf_amd.ops import _stream_ptr
        lib = _lib.load()
        n = 100_000_000
        streams = (("rows: uint8 deltas, 1 in 800 nonzero", (torch.rand(n, device=device, generator=g) < 1 / 800).to(torch.uint8)),
                   ("cols: int32 gaps, mean 800", torch.randint(1, 1600, (n // 4,), device=device, generator=g, dtype=torch.int32)))
        for label, t in streams:
            es, ne = t.element_size(), t.numel()
            out = torch.empty(lib.rlx_zplane_bound_bytes(ne, es), dtype=torch.uint8, device=device)
            ws = torch.empty(lib.rlx_zplane_workspace_bytes(ne, es), dtype=torch.uint8, device=device)
            length = torch.zeros(1, dtype=torch.int64, device=device)
            back, status = torch.empty_like(t), torch.zeros(1, dtype=torch.int32, device=device)
            st = _stream_ptr(dev_t)
            comp = lambda: _lib.check(lib.rlx_zplane_compress(t.data_ptr(), ne, es, out.data_ptr(), out.numel(), length.data_ptr(),  # noqa: E731
                                                              ws.data_ptr(), ws.numel(), st), "rlx_zplane_compress")
            t_c = avg_us(comp)
            clen = int(length.item())
            dec = lambda: _lib.check(lib.rlx_zplane_decompress(out.data_ptr(), clen, back.data_ptr(), ne, es, status.data_ptr(), st),  # noqa: E731
                                     "rlx_zplane_decompress")
            t_d = avg_us(dec)
            ok = bool(torch.equal(back, t)) and int(status.item()) == 0
            raw = ne * es
            for kname, us, nb in (("zplane compress (measure + scan + pack)", t_c, 2 * raw + clen), ("zplane decompress", t_d, raw + clen)):
                rows.append({"kernel": kname, "bound": "hbm", "achieved": round(nb / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(nb / us / 1e3 / HBM_PEAK_GBPS, 4), "avg_launch_us": round(us, 1), "algorithmic_bytes": nb,
                             "shape": f"{label}; {raw} -> {clen} bytes ({raw / clen:.1f} x), round trip {'ok' if ok else 'MISMATCH'}"})
            del out, ws, back
    except Exception as e:  # noqa: BLE001
        rows.append({"kernel": "zplane codec", "error": f"{type(e).__name__}: {e}"[:200]})
    return rows
