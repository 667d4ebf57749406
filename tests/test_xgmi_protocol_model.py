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


# ---------------------------------------------------------------------------------------------------------------------------------
# The exchange INSIDE the one-launch optimizer step (csrc/adamw_clip.hip: XchgPeers, reduce_clip_adamw_one_launch<.., XCHG = true>):
# no flags at all -- every value is PUSHED into the consumer's buffer as a self-validating word (epoch << 32 | payload, one
# single-copy-atomic store) and the consumer polls the payload itself.  Per launch, block b of rank r:
#
#     e1 = this rank's epoch + 1
#     owner(b) = b * W // nblk
#     non-owner:  push its K words into the owner's inbox row [r]                     (K separate stores, any order in time)
#     owner:      poll rows p != r of its own inbox (word by word), reduced = sum over ranks IN RANK ORDER,
#                 push `reduced` into every other rank's gather words, publish the block's partial to every rank's slot b
#     non-owner:  poll its own gather words
#     everybody:  poll all nblk partial slots of its own buffer; block 0 then stores the rank's epoch = e1
#
# and a rank's launch L + 1 starts when its launch L has ended (stream order) -- the RANKS are not synchronised with each other: a
# fast rank is already pushing launch L + 1 while a slow one still polls launch L.  Claims: under any interleaving every block of
# every rank ends launch L with exactly launch L's rank-ordered sum and total, nothing stale, nothing mixed, nobody waits forever.
# Negative controls: a word whose two halves are stored separately under ONE tag (torn word), and an epoch that belongs to the
# CALLER instead of the communicator (two parameter sets alternating on one communicator reuse tags).
K_WORDS = 2


class ExchangeDeadlock(AssertionError):
    pass


def _poll(cell, e1, who):
    spins = 0
    while True:
        tag, val = cell["w"]
        if tag == e1:
            return val
        spins += 1
        if spins > 600:
            raise ExchangeDeadlock(f"{who} polls for tag {e1} forever (word holds tag {tag})")
        yield None


def _store(cell, e1, val, variant):
    if variant == "torn" and isinstance(val, tuple):  # negative control: two halves under one tag, stored one after the other
        old = cell["w"][1] if isinstance(cell["w"][1], tuple) else (0, 0)
        cell["w"] = (e1, (val[0], old[1]))
        yield None
        cell["w"] = (e1, val)
    else:
        cell["w"] = (e1, val)
    yield None


def _xblock(r, b, W, nblk, mem, truth, expect, e1, variant):
    me = mem[r]
    owner = b * W // nblk
    g = truth[r][b]
    if owner != r:
        for k in range(K_WORDS):
            yield from _store(mem[owner]["inbox"][r][b][k], e1, g[k], variant)
        reduced = []
        for k in range(K_WORDS):
            v = yield from _poll(me["gather"][b][k], e1, f"rank {r} block {b} (gather word {k})")
            reduced.append(v)
    else:
        reduced = []
        for k in range(K_WORDS):
            acc = None
            for p in range(W):  # rank order, own contribution at position r
                if p == r:
                    t = g[k]
                else:
                    t = yield from _poll(me["inbox"][p][b][k], e1, f"rank {r} block {b} (inbox row {p} word {k})")
                acc = t if acc is None else (acc[0] + t[0], acc[1] + t[1])
            reduced.append(acc)
        for p in range(W):
            if p != r:
                for k in range(K_WORDS):
                    yield from _store(mem[p]["gather"][b][k], e1, reduced[k], variant)
        for p in range(W):
            yield from _store(mem[p]["parts"][b], e1, sum(x[0] + x[1] for x in reduced), "product")
    assert reduced == expect["reduced"][b], f"rank {r} block {b}: {reduced} != {expect['reduced'][b]}"
    total = 0
    for j in range(nblk):
        v = yield from _poll(me["parts"][j], e1, f"rank {r} block {b} (partial {j})")
        total += v
    assert total == expect["total"], f"rank {r} block {b}: total {total} != {expect['total']}"
    if b == 0:
        me["epoch_done"] = e1
    yield None


def _run_exchange(W, nblk, n_launches, seed, variant="product", bias=None):
    rng = random.Random(seed)
    cell = lambda: {"w": (0, (0, 0))}  # noqa: E731
    mem = [{"epoch_done": 0, "inbox": [[[cell() for _ in range(K_WORDS)] for _ in range(nblk)] for _ in range(W)],
            "gather": [[cell() for _ in range(K_WORDS)] for _ in range(nblk)], "parts": [{"w": (0, 0)} for _ in range(nblk)]}
           for _ in range(W)]
    truths, expects = [], []
    for L in range(1, n_launches + 1):
        truth = [[[(rng.getrandbits(12) + (L << 16), rng.getrandbits(12) + (L << 16)) for _ in range(K_WORDS)] for _ in range(nblk)]
                 for _ in range(W)]
        red = [[(sum(truth[p][b][k][0] for p in range(W)), sum(truth[p][b][k][1] for p in range(W))) for k in range(K_WORDS)]
               for b in range(nblk)]
        truths.append(truth)
        expects.append({"reduced": red, "total": sum(x[0] + x[1] for b in range(nblk) for x in red[b])})
    launch_of = [0] * W          # launch a rank is in (1-based), 0 = not started
    progs = {}                   # (rank, block) -> generator

    def start(r):
        launch_of[r] += 1
        L = launch_of[r]
        # the communicator's epoch counts ITS exchanges; negative control "epoch_per_caller": two callers alternate on one
        # communicator, each counting its own launches -> launches 1 and 2 both use tag 1
        e1 = (L + 1) // 2 if variant == "epoch_per_caller" else L
        for b in range(nblk):
            progs[(r, b)] = _xblock(r, b, W, nblk, mem, truths[L - 1], expects[L - 1], e1, variant)

    for r in range(W):
        start(r)
    steps = 0
    while progs:
        keys = sorted(progs)
        if bias == "rank0_fast" and rng.random() < 0.8 and any(k[0] == 0 for k in keys):
            key = rng.choice([k for k in keys if k[0] == 0])
        elif bias == "last_rank_slow" and rng.random() < 0.9 and any(k[0] != W - 1 for k in keys):
            key = rng.choice([k for k in keys if k[0] != W - 1])
        else:
            key = rng.choice(keys)
        try:
            next(progs[key])
        except StopIteration:
            del progs[key]
            r = key[0]
            if not any(k[0] == r for k in progs) and launch_of[r] < n_launches:
                start(r)  # stream order: the rank's next launch starts when this one has ended
        steps += 1
        assert steps < 2_000_000
    assert all(launch_of[r] == n_launches for r in range(W))
    return True


@pytest.mark.parametrize("bias", [None, "rank0_fast", "last_rank_slow"])
@pytest.mark.parametrize("W,nblk", [(2, 2), (2, 5), (3, 6), (4, 4)])
def test_pushed_word_exchange_is_correct_under_any_interleaving(W, nblk, bias):
    for seed in range(25):
        assert _run_exchange(W, nblk, 4, seed, bias=bias)


def test_negative_control_torn_word():
    """A 16-byte payload under ONE tag whose halves land one after the other: a poller accepts the tag with one half still the
    previous launch's.  (The kernel tags every 64-bit word: what is single-copy atomic is exactly what carries a tag.)"""
    caught = 0
    for seed in range(40):
        try:
            _run_exchange(2, 3, 4, seed, variant="torn", bias="rank0_fast")
        except ExchangeDeadlock:
            raise
        except AssertionError:
            caught += 1
    assert caught >= 20, caught


def test_negative_control_epoch_owned_by_the_caller():
    """Two parameter sets (the learner's and the start-up validation's) taking turns on one communicator, each with its own epoch
    counter: the second set's launch polls for a tag the first set's launch left in every word -- stale values pass.  (The
    communicator owns the epoch: rlx_xgmi_comm::xsync.)"""
    caught = 0
    for seed in range(40):
        try:
            _run_exchange(2, 3, 4, seed, variant="epoch_per_caller")
        except (AssertionError, ExchangeDeadlock):
            caught += 1
    assert caught >= 30, caught
