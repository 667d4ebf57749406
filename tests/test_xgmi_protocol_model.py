"""A schedule-exploring model of the xGMI all-reduce hand-shake (csrc/xgmi_allreduce.hip, csrc/adamw_clip.hip::peer_handshake).

The transport itself needs one GPU per rank to be exercised (tests/test_distributed.py); what can be checked anywhere is the
PROTOCOL argument DESIGN.md section 6 makes: with two staging slots and one monotonic flag per (reader, writer) pair, no rank
ever reads a slot that is being rewritten or that holds another all-reduce's gradient -- under ANY interleaving of the ranks'
kernel chains.  Each rank runs, per all-reduce s = 1, 2, ...:

    stage(s)   write own slot (s & 1), several non-atomic steps             (xgmi_stage_kernel)
    reduce(s)  publish flag[peer][me] = s to every peer                      (peer_handshake: block 0, release store)
               wait until flag[me][peer] >= s for every peer                 (acquire loads, bounded spin)
               read slot (s & 1) of every peer, several non-atomic steps     (grad_reduce_sqnorm: peer loads)

in stream order.  A random scheduler advances one rank by one micro-step at a time; the model asserts that every read sees
version s of the slot, complete, with no write in progress.  The negative control (ONE slot) must trip the same assertion: the
model has teeth."""

import random

import pytest

WRITE_STEPS = 3   # a slot write / read is not atomic: it spans several scheduler steps
READ_STEPS = 3


class Slot:
    def __init__(self):
        self.version = 0       # all-reduce whose gradient the slot holds (complete)
        self.writing = None    # all-reduce being written right now, or None
        self.readers = 0       # peers currently reading it


def _rank_program(me, world, n_allreduce, slots, flags, n_slots):
    """Generator: one micro-step per `yield`.  Raises AssertionError on a protocol violation."""
    for s in range(1, n_allreduce + 1):
        mine = slots[me][s % n_slots]
        # ---- stage(s) ----
        assert mine.readers == 0, f"rank {me} starts overwriting its slot for all-reduce {s} while a peer still reads it"
        mine.writing = s
        for _ in range(WRITE_STEPS):
            yield
            assert mine.readers == 0, f"a peer started reading rank {me}'s slot while all-reduce {s} was being staged"
        mine.version, mine.writing = s, None
        yield
        # ---- reduce(s): publish, wait, read ----
        for peer in range(world):
            flags[peer][me] = s
        yield
        while any(flags[me][peer] < s for peer in range(world)):
            yield
        for peer in range(world):
            if peer == me:
                continue
            theirs = slots[peer][s % n_slots]
            theirs.readers += 1
            for _ in range(READ_STEPS):
                assert theirs.writing is None and theirs.version == s, (
                    f"rank {me}, all-reduce {s}: peer {peer}'s slot holds version {theirs.version}, writing={theirs.writing}")
                yield
            theirs.readers -= 1
        yield  # clip + AdamW, the sequence number advances


def _explore(world, n_allreduce, n_slots, seed, bias=None):
    rng = random.Random(seed)
    slots = [[Slot() for _ in range(n_slots)] for _ in range(world)]
    flags = [[0] * world for _ in range(world)]
    progs = {r: _rank_program(r, world, n_allreduce, slots, flags, n_slots) for r in range(world)}
    steps = 0
    while progs:
        ranks = sorted(progs)
        # biased schedulers: let one rank race ahead / lag behind as far as the protocol allows
        r = ranks[0] if (bias == "first" and rng.random() < 0.9) else (ranks[-1] if (bias == "last" and rng.random() < 0.9) else rng.choice(ranks))
        try:
            next(progs[r])
        except StopIteration:
            del progs[r]
        steps += 1
        assert steps < 200_000, "the model did not terminate: a rank waits for a flag nobody will publish (deadlock)"
    return steps


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_two_slots_are_safe_under_random_interleavings(world):
    for seed in range(150):
        for bias in (None, "first", "last"):
            _explore(world, n_allreduce=6, n_slots=2, seed=seed, bias=bias)


def test_one_slot_is_caught_by_the_model():
    """Negative control: with a single staging slot a fast rank restages while a slow peer still reads (or reads the next
    all-reduce's gradient) -- some interleaving must trip the model's assertions."""
    tripped = 0
    for seed in range(150):
        for bias in (None, "first", "last"):
            try:
                _explore(3, n_allreduce=6, n_slots=1, seed=seed, bias=bias)
            except AssertionError as e:
                assert "deadlock" not in str(e)
                tripped += 1
    assert tripped > 0
