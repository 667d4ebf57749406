"""Workload for the rocprofv3 --pmc passes behind bench.py's roofline.traffic: a calibration copy with a known byte
count in gae_scan's access pattern (one dword per lane, rows 4*B bytes apart), then gae_scan at the roofline shape
(65536 envs x 128 steps) rotating over buffer sets larger than the Infinity Cache."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rlinf_amd import _lib, ops

T, B, nbuf = 128, 65536, 5
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.rlx_dev_stream_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
g = torch.Generator().manual_seed(0)
bufs = []
for _ in range(nbuf):
    r = torch.rand(T, B, 1, generator=g).cuda()
    v = torch.randn(T + 1, B, 1, generator=g).cuda()
    d = (torch.rand(T + 1, B, 1, generator=g) < 0.02).cuda()
    bufs.append((r, v, d, torch.empty_like(r), torch.empty_like(r)))
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for i in range(nbuf):  # calibration: T x B floats copied, reads and writes 4*T*B bytes each
    raw.rlx_dev_stream_copy(bufs[i][0].data_ptr(), bufs[i][3].data_ptr(), T, B, st)
torch.cuda.synchronize()
for i in range(2 * nbuf):
    r, v, d, a, q = bufs[i % nbuf]
    ops.gae_scan(r, v, d, None, 0.99, 0.95, normalize_advantages=False, out=(a, q))
torch.cuda.synchronize()
