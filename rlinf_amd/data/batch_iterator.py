"""The reasoning learner's streaming batch source in pipeline mode: rollout results arrive in pieces, training consumes fixed-size
micro-batches.  Same public surface and call order as the reference's ``BatchResizingIterator``
(rlinf/data/schema/reasoning_results.py:1424-1600) and the helpers it leans on -- ``RolloutResult.merge_batches`` (:692-712) and
``get_iterator_k_split`` (rlinf/utils/data_iter_utils.py:129-262):

    received batch --(get-batch handler: advantages)--> [topped up to one global batch when it is smaller and a global-batch
    handler is registered] --(seeded shuffle, equal split)--> global batches --(global-batch handler: advantage
    normalisation)--(seeded shuffle, equal split)--> micro-batches

Host logic only; the tensors stay where they are (the row gathers of the two shuffles run on the tensors' own device, one index
tensor per shuffle instead of a Python list of indices).
"""

from __future__ import annotations

from typing import Callable, Iterator, Optional

import torch


def merge_batches(batches: list) -> dict:
    """Row-wise concatenation of same-keyed batches (tensors along dim 0, lists extended); one batch is returned as it is."""
    if len(batches) == 0:
        return {}
    if len(batches) == 1:
        return batches[0]
    keys = batches[0].keys()
    assert all(b.keys() == keys for b in batches[1:]), "All batches must have the same keys"
    out = {}
    for k in keys:
        first = batches[0][k]
        if torch.is_tensor(first):
            out[k] = torch.cat([b[k] for b in batches], dim=0)
        elif isinstance(first, list):
            out[k] = [x for b in batches for x in b[k]]
        else:
            raise ValueError(f"Unsupported batch key type: {type(first)}")
    return out


def batch_rows(batch: dict, key: str = "input_ids") -> int:
    return int(batch[key].shape[0])


def k_split(batch: dict, num_splits: int, shuffle: bool = False, shuffle_seed: Optional[int] = None) -> Iterator[dict]:
    """``num_splits`` equal row ranges of the tensors / lists of ``batch`` (other entries are dropped), after one seeded row
    permutation when ``shuffle`` -- a generator re-seeded on every call, so equal sizes get equal permutations."""
    items = {k: v for k, v in batch.items() if isinstance(v, (torch.Tensor, list))}
    tensors = [v for v in items.values() if isinstance(v, torch.Tensor)]
    if not items:
        raise ValueError("Batch contains no tensors or lists to determine batch size.")
    n = int(tensors[0].shape[0]) if tensors else len(next(iter(items.values())))
    assert num_splits > 0 and n % num_splits == 0, "Issue with batch size configuration!"
    if shuffle:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(int(shuffle_seed)))
        by_device, host = {}, order.tolist()

        def rows(v):
            if isinstance(v, list):
                return [v[i] for i in host]
            if v.device not in by_device:
                by_device[v.device] = order.to(v.device)
            return v[by_device[v.device]]

        items = {k: rows(v) for k, v in items.items()}
    per = n // num_splits
    for i in range(num_splits):
        yield {k: v[i * per:(i + 1) * per] for k, v in items.items()}


class BatchResizingIterator:
    """``next(it)`` is the next micro-batch of the stream described in the module docstring.

    ``get_batch_fn() -> (batch, result)``: ``result.num_sequence`` (or ``result`` itself when it is an int) is the number of
    sequences received.  ``total_batch_size`` / ``num_global_batches`` = sequences per global batch (one optimizer step);
    the iterator checks that micro-batches add up to exactly that."""

    def __init__(self, cfg, get_batch_fn: Callable, micro_batch_size: int, total_batch_size: int, num_global_batches: int,
                 forward_only: bool, batch_tensor_key: str = "input_ids"):
        self.cfg = cfg
        self.get_batch_fn = get_batch_fn
        self.micro_batch_size = int(micro_batch_size)
        self.num_global_batches = int(num_global_batches)
        self.forward_only = forward_only
        self.batch_tensor_key = batch_tensor_key
        self.reset_total_batch_size(total_batch_size)
        algo = cfg["algorithm"] if isinstance(cfg, dict) else cfg.algorithm
        actor = cfg["actor"] if isinstance(cfg, dict) else cfg.actor
        shuffle = algo.get("shuffle_rollout", True)
        self._shuffle = True if shuffle is None else bool(shuffle)
        self._seed = actor["seed"] if isinstance(actor, dict) else actor.seed
        self.consumed_batch_size = 0
        self.global_batch_done = False
        self.prefetch_micro_batch = None
        self.batches: list = []
        self.get_batch_fn_handler: Optional[Callable] = None
        self.global_batch_handler: Optional[Callable] = None
        self._stream = self._micro_batches()

    # ---- configuration ------------------------------------------------------------------------------------------------------
    def register_get_batch_handler(self, handler: Callable):
        """Applied to every received batch right after ``get_batch_fn`` (the learner: advantages and returns)."""
        self.get_batch_fn_handler = handler

    def register_global_batch_handler(self, handler: Callable):
        """Applied to every global batch before it is cut into micro-batches (the learner: advantage normalisation).  With one
        registered, a received batch smaller than a global batch is topped up to a whole one first."""
        self.global_batch_handler = handler

    def reset_total_batch_size(self, total_batch_size: int):
        self.total_batch_size = int(total_batch_size)
        self.global_batch_size = self.total_batch_size // self.num_global_batches

    # ---- the stream -----------------------------------------------------------------------------------------------------------
    def _receive(self):
        batch, result = self.get_batch_fn()
        if self.get_batch_fn_handler is not None:
            batch = self.get_batch_fn_handler(batch)
        return batch, int(getattr(result, "num_sequence", result))

    def _global_batches(self) -> Iterator[dict]:
        batch, n = self._receive()
        gbs = self.global_batch_size
        if n % gbs != 0:
            if self.global_batch_handler is None:
                yield batch  # smaller than a global batch and nobody needs whole ones: handed on as it is
                return
            n = batch_rows(batch, self.batch_tensor_key)
            while n < gbs and n % gbs != 0:
                more, _ = self._receive()
                batch = merge_batches([batch, more])
                n = batch_rows(batch, self.batch_tensor_key)
        yield from k_split(batch, n // gbs, self._shuffle, self._seed)

    def _micro_batches(self) -> Iterator[dict]:
        while True:
            for global_batch in self._global_batches():
                if self.global_batch_handler is not None:
                    global_batch = self.global_batch_handler(global_batch)
                    assert global_batch is not None, f"global batch handler {self.global_batch_handler} must not return None."
                n = batch_rows(global_batch, self.batch_tensor_key)
                for micro_batch in k_split(global_batch, n // self.micro_batch_size, self._shuffle, self._seed):
                    self.global_batch_done = False
                    self.consumed_batch_size += batch_rows(micro_batch, self.batch_tensor_key)
                    self.batches.append(micro_batch)
                    if self.consumed_batch_size == self.global_batch_size:
                        self.consumed_batch_size = 0
                        self.global_batch_done = True
                    else:
                        assert self.consumed_batch_size < self.global_batch_size, (
                            f"Received batches with a total size of {self.consumed_batch_size}, which exceeds the global batch size per dp "
                            f"{self.global_batch_size}. This suggests that the configured global batch size cannot be divided by the actual "
                            f"batch size.")
                    yield micro_batch

    def __iter__(self):
        return self

    def __next__(self) -> dict:
        if self.prefetch_micro_batch is not None:
            micro_batch, self.prefetch_micro_batch = self.prefetch_micro_batch, None
            return micro_batch
        return next(self._stream)

    def prefetch_one_batch(self) -> dict:
        """Look at the next micro-batch without consuming it."""
        if self.prefetch_micro_batch is None:
            self.prefetch_micro_batch = next(self)
        return self.prefetch_micro_batch

    # ---- bookkeeping -----------------------------------------------------------------------------------------------------------
    def check_finished_global_batch(self):
        assert self.global_batch_done, (
            f"Batch iterator has not finished for this global batch, only consumed {self.consumed_batch_size} sequences, expected "
            f"{self.global_batch_size}")

    def get_all_batches(self) -> dict:
        """Everything handed out since the last call, merged (the iteration's rollout metrics are computed on it)."""
        batch = merge_batches(self.batches)
        self.batches = []
        return batch
