"""Dev tool (GPU box): two processes on the visible GPU(s) -- which memory kinds can be exported over HIP IPC, does the peer-read
all-reduce give the right sums, and what does one all-reduce of the gradient (287504 floats) cost.
    python tools/xgmi_probe.py            (parent: spawns 2 ranks)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    import torch.distributed as dist
    from rlinf_amd.scheduler import init_distributed
    from rlinf_amd.scheduler.xgmi import XgmiAllReduce, _attempt
    ctx = init_distributed()
    n = 287504
    # RLX_XGMI_PROBE_LIGHT (the one-GPU unit test): the first coherent memory kind only, a handful of launches -- two processes
    # time-slicing ONE GPU can starve each other's spin waits for seconds per hand-off
    light = os.environ.get("RLX_XGMI_PROBE_LIGHT") == "1"
    n_eager, n_graph, n_replay = (6, 4, 2) if light else (200, 20, 10)
    for kind in ((0, 1) if light else (0, 1, 2)):
        comm, why = _attempt(ctx, n, kind, 2 if light else 4)
        if ctx.rank == 0:
            print(f"mem_kind {kind}: {'OK' if comm is not None else 'FAILED: ' + why}", flush=True)
        if comm is None:
            continue
        if ctx.rank == 0:
            print(f"   shared device: {comm.shared_device}; wait_mode {comm.wait_mode}", flush=True)
        x = torch.randn(24, n, device=ctx.device)
        out = torch.empty(n, device=ctx.device)
        for algo in ("direct", "rsag"):
            comm.configure(algo=algo)
            for _ in range(5):
                comm.all_reduce(x, out, 0.5)
            torch.cuda.synchronize()
            comm.check_status()
            dist.barrier()
            t0 = time.perf_counter()
            iters = n_eager
            for _ in range(iters):
                comm.all_reduce(x, out, 0.5)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters * 1e6
            comm.check_status()
            # the same inside a captured graph
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(n_graph):
                    comm.all_reduce(x, out, 0.5)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(n_replay):
                g.replay()
            torch.cuda.synchronize()
            dg = (time.perf_counter() - t0) / (n_graph * n_replay) * 1e6
            comm.check_status()
            want = x.sum(0)
            dist.all_reduce(want)
            ok = torch.allclose(out * 2, want, rtol=1e-5, atol=1e-5)
            if ctx.rank == 0:
                print(f"   {algo}: all-reduce {dt:.1f} us eager, {dg:.1f} us in a replayed graph; {algo}: result ok after graph: {ok}", flush=True)
            del g
        comm.close()
        if light:
            break
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if os.environ.get("RANK") is not None:
        os.environ.setdefault("RLX_XGMI_TIMEOUT_MS", "20000")
        child()
    else:
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29733",
                       RLX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
        rc = 0
        for p in procs:
            try:
                rc |= p.wait(timeout=240)
            except subprocess.TimeoutExpired:
                p.kill()
                rc = 1
        sys.exit(rc)
