"""A schedule-exploring model of the exchange inside the one-launch optimizer step (csrc/adamw_clip.hip::reduce_clip_adamw_one_launch).

The kernel needs a GPU (tests/test_gpu_losses.py::test_one_launch_optimizer_step); what can be checked anywhere is the PROTOCOL
argument DESIGN.md section 4.2 makes.  Per launch every block b

    reads the epoch word E and sets e1 = E + 1                               (relaxed atomic load, before anything is published)
    publishes its f64 norm partial as TWO 64-bit words (e1 << 32 | half)     (two relaxed atomic stores, in either order in time)
    polls every block's two words until BOTH carry tag e1, takes the halves  (relaxed atomic loads; a word is never torn)
    block 0 only, after ITS poll completed: stores the epoch word = e1
    applies the update

and the next launch starts when the previous one has ended (stream order).  Claims: under ANY interleaving of the blocks'
micro-steps (a) every block assembles exactly the partial every other block published in THIS launch -- never a stale half of
an earlier launch, never a mix --, (b) every block of a launch uses the same e1, (c) nothing waits forever.  No fence orders data
against a flag anywhere: each word carries its own validity.  Negative controls: the epoch advanced before block 0's poll (a late
block then computes a different e1 and the launch dead-locks), and a separate flag word in front of untagged data words (a reader
can then pair a fresh flag with stale data): the model must catch both."""

import random

import pytest


class Deadlock(AssertionError):
    pass


def _block(b, n_blocks, mem, truth, variant, log):
    """One block of one launch as a generator: a `yield` is a point where the scheduler may run another block."""
    e1 = mem["epoch"] + 1
    log[b] = e1
    yield
    lo, hi = truth[b]
    slot = mem["slots"][b]
    if variant == "flag_then_data":          # negative control: one flag word, data words without tags
        slot["flag"] = e1
        yield
        slot["lo"], slot["hi"] = (None, lo), (None, hi)
        yield
    else:
        slot["lo"] = (e1, lo)
        yield
        slot["hi"] = (e1, hi)
        yield
    if variant == "epoch_early" and b == 0:  # negative control: the epoch moves before every block has read it
        mem["epoch"] = e1
        yield
    got = {}
    pending = set(range(n_blocks))
    spins = 0
    while pending:
        for j in sorted(pending):
            s = mem["slots"][j]
            if variant == "flag_then_data":
                ok = s["flag"] == e1
                yield
                w0, w1 = s["lo"], s["hi"]
            else:
                w0 = s["lo"]                  # two separate loads: another block may store between them
                yield
                w1 = s["hi"]
                ok = w0[0] == e1 and w1[0] == e1
            if ok:
                got[j] = (w0[1], w1[1])
                pending.discard(j)
            yield
        spins += 1
        if spins > 400:
            raise Deadlock(f"block {b} polls for tag {e1} forever: {sorted(pending)} never published it")
    for j in range(n_blocks):
        assert got[j] == truth[j], f"block {b} assembled {got[j]} for block {j}, which published {truth[j]} in this launch"
    if b == 0 and variant != "epoch_early":
        mem["epoch"] = e1
    yield


def _run(n_blocks, n_launches, seed, variant="product", bias=None):
    rng = random.Random(seed)
    mem = {"epoch": 0, "slots": [{"flag": 0, "lo": (0, 0), "hi": (0, 0)} for _ in range(n_blocks)]}
    for launch in range(1, n_launches + 1):
        truth = {b: (rng.getrandbits(16) | (launch << 20), rng.getrandbits(16) | (launch << 20)) for b in range(n_blocks)}
        log = {}
        progs = {b: _block(b, n_blocks, mem, truth, variant, log) for b in range(n_blocks)}
        while progs:
            alive = sorted(progs)
            if bias == "zero_first" and 0 in progs and rng.random() < 0.85:
                b = 0
            elif bias == "zero_last" and len(alive) > 1 and rng.random() < 0.85:
                b = rng.choice([x for x in alive if x != 0])
            elif bias == "one_late" and len(alive) > 1 and rng.random() < 0.9:
                b = rng.choice([x for x in alive if x != alive[-1]])
            else:
                b = rng.choice(alive)
            try:
                next(progs[b])
            except StopIteration:
                del progs[b]
        assert len(set(log.values())) == 1, f"launch {launch}: blocks disagree on the epoch: {log}"
        assert mem["epoch"] == launch
    return True


@pytest.mark.parametrize("bias", [None, "zero_first", "zero_last", "one_late"])
@pytest.mark.parametrize("n_blocks", [2, 5, 9])
def test_exchange_is_correct_under_any_interleaving(n_blocks, bias):
    for seed in range(60):
        assert _run(n_blocks, 6, seed, bias=bias)


def test_negative_control_epoch_advanced_before_the_poll():
    """Block 0 bumping the epoch as soon as it has published: a block that starts late reads the NEW epoch, publishes and polls for
    a tag nobody else uses -- the launch never completes.  (The kernel advances the epoch behind block 0's poll for this reason.)"""
    caught = 0
    for seed in range(40):
        try:
            _run(5, 4, seed, variant="epoch_early", bias="zero_first")
        except (Deadlock, AssertionError):
            caught += 1
    assert caught >= 30, caught


def test_negative_control_flag_in_front_of_untagged_data():
    """A flag word published BEFORE tag-less data words (the classic pattern that needs a release / acquire pair, which the kernel
    avoids by tagging every word): a reader pairs a fresh flag with the previous launch's data."""
    caught = 0
    for seed in range(40):
        try:
            _run(4, 4, seed, variant="flag_then_data")
        except AssertionError:
            caught += 1
    assert caught >= 30, caught
