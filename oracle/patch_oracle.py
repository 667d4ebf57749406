"""CPU restatement of the reference's sparse weight-patch wire format.  TEST INFRASTRUCTURE ONLY.

SURVEY.md 8f item 3 / Appendix B: rlinf/hybrid_engines/weight_syncer/patch_syncer.py -- ``as_coo_2d_view`` (:60-95),
``downscale_nonnegative_indices`` (:35-57), ``PatchBuilder.delta_encode / delta_decode`` (:290-370), the same-device
builder ``GPUSnapshotPatchBuilder.create_patch`` (:648-774) and the receiver loop of ``PatchWeightSyncer.apply``
(:1040-1137).  Integer / byte work: everything here is compared bit for bit -- against the real reference classes in
``tests/test_oracle_vs_reference.py`` and against ``tests/golden/weight_patch.pt`` everywhere.

Nothing under ``rlinf_amd/`` imports this file.
"""

from __future__ import annotations

import torch


def coo_2d_view(t: torch.Tensor) -> torch.Tensor:
    if t.ndim == 0:
        return t.unsqueeze(0).unsqueeze(0)
    if t.ndim == 1:
        return t.unsqueeze(0)
    if t.ndim == 2:
        return t
    return t.view(t.shape[0], -1)


def downscale(idx: torch.Tensor) -> torch.Tensor:
    if idx.numel() == 0:
        return idx.to(torch.uint8)
    top = int(idx.max().item())
    if top <= 255:
        return idx.to(torch.uint8)
    if top <= torch.iinfo(torch.int32).max:
        return idx.to(torch.int32)
    return idx.to(torch.int64)


def delta_encode(rows, cols):
    if rows.numel() == 1:
        return rows, cols
    dr, dc = torch.empty_like(rows), torch.empty_like(cols)
    dr[0], dc[0] = rows[0], cols[0]
    dr[1:] = rows[1:] - rows[:-1]
    dc[1:] = torch.where(rows[1:] == rows[:-1], cols[1:] - cols[:-1], cols[1:])
    return dr, dc


def delta_decode(dr, dc):
    rows = torch.cumsum(dr, dim=0, dtype=torch.int64)
    start = torch.zeros_like(dr, dtype=torch.bool)
    start[0] = True
    start[1:] = dr[1:] != 0
    idx = torch.arange(dr.numel(), dtype=torch.int64)
    seg = torch.cummax(torch.where(start, idx, torch.zeros_like(idx)), dim=0).values
    cum = torch.cumsum(dc, dim=0, dtype=torch.int64)
    return rows, cum - (cum - dc)[seg]


def create_patch(state: dict, snapshot: dict, ordered_keys: list, sync_names: list, version: int, delta: bool):
    """-> None (nothing changed) or dict(version, ordinals i32, nnz_per_tensor i32, rows, cols, values u8).
    ``snapshot`` (2-D views, receiver dtypes) is updated in place, as the reference does."""
    ordinal_of = {k: i for i, k in enumerate(ordered_keys)}
    ords, nnzs, rws, cls, vals = [], [], [], [], []
    for name in sync_names:
        snap = snapshot[name]
        cur = coo_2d_view(state[name]).to(dtype=snap.dtype)
        r, c = cur.ne(snap).nonzero(as_tuple=True)
        if r.numel() == 0:
            continue
        v = cur[r, c]
        snap[r, c] = v
        if delta:
            r, c = delta_encode(r, c)
        ords.append(ordinal_of[name])
        nnzs.append(v.numel())
        rws.append(r.contiguous())
        cls.append(c.contiguous())
        vals.append(v.contiguous().view(torch.uint8))
    if not rws:
        return None
    return dict(version=torch.tensor(version, dtype=torch.int64), ordinals=torch.tensor(ords, dtype=torch.int32),
                nnz_per_tensor=torch.tensor(nnzs, dtype=torch.int32), rows=downscale(torch.cat(rws)),
                cols=downscale(torch.cat(cls)), values=torch.cat(vals))


def apply_patch(state: dict, ordered_keys: list, patch: dict, delta: bool) -> int:
    off = voff = 0
    for i in range(patch["ordinals"].numel()):
        tgt = coo_2d_view(state[ordered_keys[int(patch["ordinals"][i])]])
        nnz = int(patch["nnz_per_tensor"][i])
        r, c = patch["rows"][off:off + nnz].clone(), patch["cols"][off:off + nnz].clone()
        off += nnz
        nbytes = nnz * tgt.element_size()
        vb = patch["values"][voff:voff + nbytes]
        voff += nbytes
        if delta:
            r, c = delta_decode(r, c)
        else:
            r, c = r.to(torch.int64), c.to(torch.int64)
        tgt[r, c] = vb.clone().view(tgt.dtype)
    return int(patch["version"])
