"""Dev tool: the optimizer step of several variant builds of the library (librlx_hip_<tag>.so) timed in ONE process, alternating,
so that a box's clock state cancels.  Each timed unit is a replayed hipGraph of REPS x [rewrite the gradient slabs (a torch
kernel: leaves them the way the weight-gradient launch does, written by every XCD) + rlx_clip_adamw_step].
   python tools/ab_opt_libs.py <tag> <tag> ... [--f32] [--rounds N]        ("" or "product" = librlx_hip.so)"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import _lib
from rlinf_amd._lib import AdamwParams, AdamwGroup
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy

ap = argparse.ArgumentParser()
ap.add_argument("tags", nargs="+")
ap.add_argument("--f32", action="store_true")
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--reps", type=int, default=64)
ap.add_argument("--slabs", type=int, default=10)
ap.add_argument("--two-launch", action="store_true", help="also time the first tag without sync words (the two-launch form)")
args = ap.parse_args()
here = os.path.dirname(_lib.LIB_PATH)


def bind(tag):
    path = os.path.join(here, "librlx_hip" + ("" if tag in ("", "product") else "_" + tag) + ".so")
    lib = ctypes.CDLL(path)
    for name in ("rlx_clip_adamw_step", "rlx_adamw_workspace_bytes", "rlx_adamw_sync_words", "rlx_mlp_pack_tiles", "rlx_mlp_pack_tiles_bf16",
                 "rlx_mlp_tiles_bytes_for", "rlx_last_error"):
        restype, argtypes = _lib.PROTOTYPES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


class Unit:
    def __init__(self, tag, sync=True):
        self.name = tag + ("" if sync else " (two launches)")
        lib = self.lib = bind(tag)
        dt = torch.float32 if args.f32 else torch.bfloat16
        pol = self.pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=dt).to("cuda")
        n, lay = pol.n_params, pol.layout
        nb = lib.rlx_mlp_tiles_bytes_for(ctypes.byref(lay), 0 if args.f32 else 1)
        self.tiles = torch.empty(nb // (4 if args.f32 else 2), dtype=dt, device="cuda")
        (lib.rlx_mlp_pack_tiles if args.f32 else lib.rlx_mlp_pack_tiles_bf16)(pol.flat.data_ptr(), ctypes.byref(lay), self.tiles.data_ptr(), None)
        self.m, self.v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        self.state = torch.zeros(2, dtype=torch.int32, device="cuda")
        self.stats = torch.zeros(2, device="cuda")
        self.grads = torch.randn(args.slabs, n, device="cuda") * 0.01
        self.ws = torch.empty(lib.rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        self.sync = torch.zeros(lib.rlx_adamw_sync_words(n), dtype=torch.int64, device="cuda") if sync else None
        p = self.p = AdamwParams()
        p.beta1, p.beta2, p.eps, p.weight_decay, p.max_grad_norm = 0.9, 0.999, 1e-8, 0.01, 0.5
        groups = pol.group_ranges(3e-4, 1e-3)
        p.step, p.n_groups, p.grad_partials, p.grad_scale = 0, len(groups), args.slabs, 1.0
        for k, (b, e, lr) in enumerate(groups):
            p.groups[k] = AdamwGroup(int(b), int(e), float(lr))
        p.tile_layout, p.tiles, p.tiles_bf16 = ctypes.pointer(lay), self.tiles.data_ptr(), 0 if args.f32 else 1
        if sync:
            p.sync_words = self.sync.data_ptr()
        self.n = n
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            self.call(side.cuda_stream)
            side.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                for _ in range(args.reps):
                    self.grads.mul_(1.0)
                    self.call(side.cuda_stream)
        side.synchronize()

    def call(self, stream):
        rc = self.lib.rlx_clip_adamw_step(self.pol.flat.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.n,
                                          ctypes.byref(self.p), self.stats.data_ptr(), self.state.data_ptr(), self.ws.data_ptr(),
                                          self.ws.numel(), stream)
        if rc:
            raise RuntimeError(self.lib.rlx_last_error())

    def time(self):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.graph.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1000.0 / args.reps


units = [Unit(t) for t in args.tags]
if args.two_launch:
    units.append(Unit(args.tags[0], sync=False))
# the slab rewrite alone, for reference
g = units[0].grads
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    g.mul_(1.0)
    side.synchronize()
    only = torch.cuda.CUDAGraph()
    with torch.cuda.graph(only, stream=side):
        for _ in range(args.reps):
            g.mul_(1.0)
side.synchronize()
for u in units:
    u.time()
res = {u.name: [] for u in units}
base = []
for r in range(args.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); only.replay(); e1.record(); e1.synchronize()
    base.append(e0.elapsed_time(e1) * 1000.0 / args.reps)
    for u in (units if r % 2 == 0 else units[::-1]):
        res[u.name].append(u.time())
med = lambda xs: sorted(xs)[len(xs) // 2]
b = med(base)
print(f"slab rewrite alone: {b:.2f} us per launch (median of {args.rounds})")
for name, xs in res.items():
    print(f"{name:28s} rewrite + step: median {med(xs):6.2f}  min {min(xs):6.2f}  max {max(xs):6.2f}   -> step ~ {med(xs) - b:5.2f} us")
for u in units:
    assert float(u.stats[1]) == 1.0 and int(u.state[0]) > 0
