"""Host-side pieces of the loop that need no accelerator: metrics on their way from the device (utils/pending.py, CPU form), the
runner's one-iteration-late generator with stand-in workers, and the streaming batch iterator's bookkeeping on plain tensors."""

import pytest
import torch

from rlinf_amd.utils.pending import PendingMetrics, resolve


def test_pending_metrics_finish_once_and_passthrough():
    calls = []

    def finish(host):
        calls.append(list(host))
        return {"a": host[0] + host[1]}

    p = PendingMetrics(torch.tensor([1.5, 2.0], dtype=torch.float64), finish)
    assert not calls                      # nothing is read until somebody asks
    assert resolve(p) == {"a": 3.5} and resolve(p) == {"a": 3.5} and len(calls) == 1
    assert resolve({"b": 1}) == {"b": 1}  # a plain dict is its own result


class _Handle:
    def __init__(self, value):
        self.value = value

    def wait(self):
        return [self.value]


class _Actor:
    """The calls EmbodiedRunner.run_step makes on its actor group, recorded; metrics come back pending when asked to."""

    def __init__(self, log):
        self.log, self.worker, self.step = log, self, 0
        self.device, self.defer_host_reads = None, False

    def set_global_step(self, s):
        self.step = s
        return _Handle(None)

    def recv_rollout_trajectories(self, input_channel=None):
        return _Handle(None)

    def compute_advantages_and_returns(self):
        s = self.step
        self.log.append(("queued", s))
        fin = lambda host, s=s: (self.log.append(("read", s)), {"rewards": float(s)})[1]  # noqa: E731
        return _Handle(PendingMetrics(torch.zeros(1), fin) if self.defer_host_reads else fin([0.0]))

    def run_training(self):
        return _Handle({"actor/total_loss": 10.0 + self.step})

    def sync_model_to_rollout(self):
        return _Handle(None)


class _Quiet:
    def __getattr__(self, name):
        return lambda *a, **k: _Handle(None)


@pytest.mark.parametrize("defer", [True, False])
def test_runner_reads_metrics_one_iteration_late_and_in_order(defer):
    from rlinf_amd.config import DictConfig
    from rlinf_amd.runners import EmbodiedRunner
    log = []
    cfg = DictConfig(dict(runner=dict(max_epochs=4, max_steps=-1, defer_metrics=defer),
                          env=dict(train=dict(total_num_envs=8, max_steps_per_rollout_epoch=4, rollout_epoch=1))))
    runner = EmbodiedRunner(cfg, _Actor(log), _Quiet(), _Quiet())
    seen = [m["rollout/rewards"] for m in runner.iter_steps()]
    assert seen == [0.0, 1.0, 2.0, 3.0] and [m["train/actor/total_loss"] for m in runner.metrics_history] == [10.0, 11.0, 12.0, 13.0]
    if defer:   # step i's numbers are read after step i + 1 has been queued, never later than that
        assert log == [("queued", 0), ("queued", 1), ("read", 0), ("queued", 2), ("read", 1), ("queued", 3), ("read", 2), ("read", 3)]
    else:
        assert log == [("queued", 0), ("read", 0), ("queued", 1), ("read", 1), ("queued", 2), ("read", 2), ("queued", 3), ("read", 3)]


def test_batch_iterator_tops_up_small_pieces_only_when_a_global_handler_needs_whole_batches():
    from rlinf_amd.data.batch_iterator import BatchResizingIterator, k_split, merge_batches
    cfg = dict(algorithm=dict(shuffle_rollout=False), actor=dict(seed=7))
    rows = torch.arange(16).reshape(16, 1)

    def feed_of(sizes):
        pieces, lo = [], 0
        for n in sizes:
            pieces.append({"input_ids": rows[lo:lo + n].clone()})
            lo += n
        return lambda: (lambda p: (p, p["input_ids"].shape[0]))(pieces.pop(0))

    it = BatchResizingIterator(cfg, feed_of([2, 2, 4, 8]), micro_batch_size=2, total_batch_size=16, num_global_batches=2, forward_only=False)
    seen = []
    it.register_global_batch_handler(lambda b: (seen.append(b["input_ids"].shape[0]), b)[1])
    got = [next(it)["input_ids"].flatten().tolist() for _ in range(8)]
    assert got == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11], [12, 13], [14, 15]] and seen == [8, 8]
    it.check_finished_global_batch()
    assert it.get_all_batches()["input_ids"].flatten().tolist() == list(range(16)) and it.get_all_batches() == {}
    # without a handler the small pieces are handed on as they are
    it = BatchResizingIterator(cfg, feed_of([2, 2, 4, 8]), micro_batch_size=2, total_batch_size=16, num_global_batches=2, forward_only=False)
    assert next(it)["input_ids"].flatten().tolist() == [0, 1] and not it.global_batch_done
    with pytest.raises(AssertionError, match="All batches must have the same keys"):
        merge_batches([{"a": rows}, {"b": rows}])
    with pytest.raises(AssertionError, match="Issue with batch size configuration"):
        list(k_split({"input_ids": rows}, 3))


def test_ranks_share_a_device_and_sync_words_switch(monkeypatch):
    """The one-launch optimizer step needs all its workgroups resident together: a job whose ranks share a GPU does not get it
    (scheduler.ranks_share_a_device: collective, False for one rank), and RLX_ADAMW_ONE_LAUNCH=0 switches it off everywhere."""
    from rlinf_amd import ops
    from rlinf_amd.scheduler import DistContext, ranks_share_a_device
    assert ranks_share_a_device(DistContext(0, 0, 1, torch.device("cpu"))) is False
    monkeypatch.setenv("RLX_ADAMW_ONE_LAUNCH", "0")
    assert ops.adamw_sync_words(1024, "cpu") is None
    # the sticky word of an expired exchange: a warning and "continue on two launches" (True), not an exception
    with pytest.warns(RuntimeWarning, match="one-launch optimizer step timed out"):
        assert ops.check_adamw_sync(torch.tensor([3, 1, 0], dtype=torch.int64), float("nan")) is True
    assert ops.check_adamw_sync(torch.tensor([3, 0, 0], dtype=torch.int64), float("nan")) is False   # a real non-finite norm: the step was skipped
    assert ops.check_adamw_sync(torch.tensor([3, 1, 0], dtype=torch.int64), 0.25) is False           # finite norm: nothing is read
    assert ops.check_adamw_sync(None, float("inf")) is False
