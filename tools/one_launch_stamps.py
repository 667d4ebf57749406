"""Dev tool: per-block wall-clock stamps of the one-launch optimizer step (library built with -DRLX_ONE_LAUNCH_STAMPS,
RLX_LIB_TAG=stamps).  Prints, over all blocks, when each phase boundary is reached relative to the earliest block start."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import ops
from rlinf_amd.models.embodiment.mlp_policy import MLPPolicy

bf16 = (sys.argv[1] if len(sys.argv) > 1 else "bf16") == "bf16"
slabs = 10
pol = MLPPolicy(42, 8, 1, True, False, compute_dtype=torch.bfloat16 if bf16 else torch.float32).to("cuda")
n, lay = pol.n_params, pol.layout
tiles = pol.tiles()
m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
state = torch.zeros(2, dtype=torch.int32, device="cuda")
stats = torch.zeros(2, device="cuda")
grads = torch.randn(slabs, n, device="cuda") * 0.01
ws = torch.empty(ops._lib.load().rlx_adamw_workspace_bytes(n), dtype=torch.uint8, device="cuda")
sync = ops.adamw_sync_words(n, "cuda")
step = ops.PreparedAdamw(pol.flat.data, grads, m, v, pol.group_ranges(3e-4, 1e-3), betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
                         max_grad_norm=0.5, grad_scale=1.0, stats=stats, step_state=state, workspace=ws, tile_layout=lay, tiles=tiles, sync=sync)
names = ["start", "slab sum ready", "partial formed", "all partials seen", "coef known", "p/m/v/g stored", "image stored"]
big = torch.empty(64 << 20, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for rep in range(6):
    grads.normal_().mul_(0.01)
    big.zero_()  # push the slabs out of the caches, like the weight-gradient launch of another XCD would leave them
    torch.cuda.synchronize()
    step(stream)
    torch.cuda.synchronize()
    st = sync[-8 * 512:].view(-1, 8).cpu()
    live = st[:, 0] != 0
    st = st[live].double()
    t0 = st[:, 0].min()
    rel = (st - t0) / 100.0  # us
    print(f"rep {rep}: {int(live.sum())} blocks; start skew max {rel[:, 0].max():.2f} us")
    for k in range(1, 7):
        col = rel[:, k][st[:, k] != 0]
        if len(col):
            print(f"   {names[k]:20s} min {col.min():6.2f}  median {col.median():6.2f}  max {col.max():6.2f}")
