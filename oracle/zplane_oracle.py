"""numpy restatement of the "RLXZ" v1 patch-stream format (TEST INFRASTRUCTURE; csrc/zplane_codec.hip is the product).

The reference compresses the rows / cols / value-byte fields of a WeightPatch with nvCOMP's LZ4 codec
(rlinf/hybrid_engines/weight_syncer/compressor.py:148-199).  nvCOMP is NVIDIA-only and not installed anywhere this repo runs, and
its container is not a public format: there is no reference OUTPUT to pin against -- "parity unpinned" for the payload bytes.  What
is pinned instead: (1) this independent, loop-level restatement of the format produces byte-identical streams to the HIP codec,
(2) decode(encode(x)) == x for both, in both directions (HIP stream -> numpy decoder, numpy stream -> HIP decoder), (3) the
transport contract around it (CompressedWeightPatch fields, dtype codes, patch_syncer.py:205-250, compressor.py:35-70) against the
reference's own tables, in tests/test_weight_patch_host.py.

Format: see the header of csrc/zplane_codec.hip.
"""

from __future__ import annotations

import struct

import numpy as np

MAGIC, BLOCK, HEADER = 0x5A584C52, 4096, 24


def _pad8(n: int) -> int:
    return (n + 7) & ~7


def compress(data: np.ndarray, elem_size: int) -> np.ndarray:
    """data: uint8 [n_elems * elem_size] -> the stream as uint8."""
    assert data.dtype == np.uint8 and data.size % elem_size == 0 and elem_size in (1, 2, 4, 8)
    n = data.size // elem_size
    nb = (n + BLOCK - 1) // BLOCK
    elems = data.reshape(n, elem_size)
    entries, chunks, off = [], [], 0
    for b in range(nb):
        blk = elems[b * BLOCK:(b + 1) * BLOCK]
        for p in range(elem_size):
            plane = np.zeros(BLOCK, np.uint8)
            plane[:blk.shape[0]] = blk[:, p]
            nz = int((plane != 0).sum())
            if nz == 0:
                entries.append((0, off))
                continue
            xplane = plane ^ np.concatenate([np.zeros(1, np.uint8), plane[:-1]])  # b[i] ^ b[i-1], b[-1] = 0

            def masked_form(pl):
                nzmask = pl != 0
                groups = nzmask.reshape(64, 64)
                top, gmasks = 0, []
                for g in range(64):
                    if groups[g].any():
                        top |= 1 << g
                        gmasks.append(int(sum(1 << int(i) for i in np.nonzero(groups[g])[0])))
                body = struct.pack("<Q", top) + b"".join(struct.pack("<Q", m) for m in gmasks) + pl[nzmask].tobytes()
                return body + b"\0" * (_pad8(int(nzmask.sum())) - int(nzmask.sum()))

            plain, xored, raw = masked_form(plane), masked_form(xplane), _pad8(blk.shape[0])
            if len(plain) <= len(xored) and len(plain) < raw:
                mode, body = 1, plain
            elif len(xored) < raw:
                mode, body = 3, xored
            else:
                mode, body = 2, plane[:raw].tobytes()
            entries.append((mode, off))
            chunks.append(body)
            off += len(body)
    head = struct.pack("<IBBHQQ", MAGIC, 1, elem_size, 12, n, off)
    directory = b"".join(struct.pack("<Q", (m << 62) | o) for m, o in entries)
    return np.frombuffer(head + directory + b"".join(chunks), dtype=np.uint8).copy()


def decompress(stream: np.ndarray) -> tuple[np.ndarray, int]:
    """-> (data uint8 [n_elems * elem_size], elem_size)."""
    raw = stream.tobytes()
    magic, ver, es, blog, n, payload_bytes = struct.unpack_from("<IBBHQQ", raw, 0)
    assert magic == MAGIC and ver == 1 and blog == 12 and es in (1, 2, 4, 8)
    nb = (n + BLOCK - 1) // BLOCK
    p0 = HEADER + 8 * nb * es
    assert len(raw) == p0 + payload_bytes
    out = np.zeros((nb * BLOCK, es), np.uint8)
    for b in range(nb):
        in_block = min(BLOCK, n - b * BLOCK)
        for p in range(es):
            (e,) = struct.unpack_from("<Q", raw, HEADER + 8 * (b * es + p))
            mode, off = e >> 62, p0 + (e & ((1 << 62) - 1))
            plane = np.zeros(BLOCK, np.uint8)
            if mode == 2:
                plane[:_pad8(in_block)] = np.frombuffer(raw, np.uint8, _pad8(in_block), off)
            elif mode in (1, 3):
                (top,) = struct.unpack_from("<Q", raw, off)
                ng = bin(top).count("1")
                pos, k = off + 8 + 8 * ng, 0
                for g in range(64):
                    if (top >> g) & 1:
                        (m,) = struct.unpack_from("<Q", raw, off + 8 + 8 * k)
                        k += 1
                        for i in range(64):
                            if (m >> i) & 1:
                                plane[g * 64 + i] = raw[pos]
                                pos += 1
                if mode == 3:
                    plane = np.bitwise_xor.accumulate(plane)
            else:
                assert mode == 0
            out[b * BLOCK:(b + 1) * BLOCK, p] = plane
    return out[:n].reshape(-1).copy(), es
