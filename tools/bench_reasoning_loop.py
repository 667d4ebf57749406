#!/usr/bin/env python
"""End-to-end timing of the REASONING learner's iteration (the driver loop BASELINE.json configs[2] plugs a model into):
rlinf_amd.workers.actor.fsdp_actor_worker.FSDPActor.run_training -- advantages (GRPO), masked normalisation, seeded shuffle,
n_minibatches x micro-batches of  model(...).logits -> TokenLearnerStep (logits -> log-prob / entropy -> GRPO loss -> d_logits)
-> backward -> flat-buffer clip + AdamW  -- around a STAND-IN language model (token + position embedding, one tanh layer, an
untied vocabulary projection): the transformer is the caller's, this package's path starts at the logits.  With a model this thin
the iteration is dominated by what this package owns (the [tokens, vocab] streams) plus the stand-in's own head GEMM.

    python tools/bench_reasoning_loop.py [--seqs 32] [--prompt 256] [--response 1792] [--vocab 151936] [--micro 4] [--iters 3]
Prints one JSON line: ms per iteration, response tokens per second, the share of the iteration spent in this package's kernels
is left to a kernel trace (rocprofv3 --kernel-trace -- python tools/bench_reasoning_loop.py)."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class StandInLM(torch.nn.Module):
    """(input_ids, position_ids) -> .logits [bsz, seq, vocab] that depend on its parameters; nothing more."""

    def __init__(self, vocab: int, dim: int, max_len: int):
        super().__init__()
        self.tok = torch.nn.Embedding(vocab, dim)
        self.pos = torch.nn.Embedding(max_len, dim)
        self.mix = torch.nn.Linear(dim, dim)
        self.head = torch.nn.Linear(dim, vocab, bias=False)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, use_cache=False, **_):
        h = torch.tanh(self.mix(self.tok(input_ids) + self.pos(position_ids)))
        # (a plain namespace: a throw-away CLASS per call would sit in a reference cycle and keep its 5 GB of logits -- and their
        #  autograd graph -- alive until the cyclic collector runs)
        return SimpleNamespace(logits=self.head(h))


def rollout_batch(seed: int, bsz: int, prompt_len: int, response_len: int, vocab: int, device) -> dict:
    """One rank's rollout batch in the reference's field names (RolloutResult.to_actor_batch): left-padded prompts, right-padded
    responses, response_mask over the real response tokens, per-sequence rewards, rollout log-probs."""
    g = torch.Generator().manual_seed(seed)
    S = prompt_len + response_len
    input_ids = torch.randint(1, vocab, (bsz, S), generator=g)
    plen = torch.randint(max(2, prompt_len // 2), prompt_len + 1, (bsz,), generator=g)
    rlen = torch.randint(max(1, response_len // 2), response_len + 1, (bsz,), generator=g)
    pos = torch.arange(S).unsqueeze(0)
    attn = (pos >= (prompt_len - plen).unsqueeze(1)) & (pos < (prompt_len + rlen).unsqueeze(1))
    response_mask = (pos >= prompt_len) & (pos < (prompt_len + rlen).unsqueeze(1))
    input_ids = torch.where(attn, input_ids, torch.zeros_like(input_ids))
    position_ids = (attn.long().cumsum(dim=1) - 1).clamp(min=0)
    rewards = torch.randint(0, 2, (bsz,), generator=g).float() * 5.0 - 2.5 + torch.randn(bsz, generator=g) * 0.1
    batch = dict(input_ids=input_ids, attention_mask=attn, position_ids=position_ids, response_mask=response_mask, rewards=rewards,
                 rollout_logprobs=-torch.rand(bsz, response_len, generator=g) * 2.0, prompt_lengths=plen, response_lengths=rlen,
                 is_end=(rlen < response_len))
    return {k: v.to(device) for k, v in batch.items()}


def measure(seqs: int = 32, prompt: int = 256, response: int = 1792, vocab: int = 151936, micro: int = 4, n_mini: int = 2,
            group: int = 4, dim: int = 64, iters: int = 3, warmup: int = 1, inplace_grad: bool = True) -> dict:
    from rlinf_amd.scheduler import init_distributed
    from rlinf_amd.workers.actor.fsdp_actor_worker import FSDPActor
    cfg = dict(
        runner=dict(task_type="reasoning"),
        algorithm=dict(adv_type="grpo", group_size=group, n_minibatches=n_mini, normalize_advantages=True, shuffle_rollout=True,
                       loss_type="actor", loss_agg_func="token-mean", ratio_clip_eps=0.2, clip_ratio_high=0.28,
                       sampling_params=dict(temperature=1.0), calculate_entropy=True, entropy_bonus=0.001, kl_beta=0.0,
                       kl_penalty_type="low_var_kl", logprob_forward_micro_batch_size=micro),
        actor=dict(seed=1234, micro_batch_size=micro, global_batch_size=seqs // n_mini, model=dict(encoder_seq_length=prompt + response),
                   inplace_logits_grad=inplace_grad,
                   optim=dict(lr=1e-5, adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, weight_decay=0.01, clip_grad=1.0)),
        data=dict(rollout_batch_size=seqs // group, max_prompt_length=prompt))
    ctx = init_distributed()
    torch.manual_seed(5)
    actor = FSDPActor(cfg, ctx, model=StandInLM(vocab, dim, prompt + response).to(ctx.device))
    times, last = [], None
    for it in range(warmup + iters):
        batch = rollout_batch(7 + it, seqs, prompt, response, vocab, ctx.device)
        torch.cuda.synchronize(ctx.device)
        t0 = time.perf_counter()
        _, last = actor.run_training([batch])
        torch.cuda.synchronize(ctx.device)
        if it >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    tokens = seqs * response  # positions of the response window the token kernels stream per iteration (masked ones included)
    return {"workload": f"reasoning GRPO learner iteration: {seqs} sequences x ({prompt} + {response}) tokens, vocab {vocab}, f32 logits, "
                        f"{n_mini} optimizer steps x {seqs // n_mini // micro} micro-batches of {micro} sequences "
                        f"({micro * response} x {vocab} logits = {micro * response * vocab * 4 / 1e9:.2f} GB per micro-batch), "
                        "stand-in language model (embedding + one tanh layer + vocabulary projection); d_logits "
                        + ("written into the logits buffer (actor.inplace_logits_grad)" if inplace_grad else "in a buffer of its own"),
            "ms_per_iteration": round(med * 1e3, 2), "ms_per_iteration_min_max": [round(times[0] * 1e3, 2), round(times[-1] * 1e3, 2)],
            "response_tokens_per_sec": round(tokens / med, 1), "iterations_timed": len(times), "optimizer_steps": actor.optimizer_steps,
            "final_loss_last_step": float(last[-1]["actor/final_loss"]) if last else None}


def main():
    ap = argparse.ArgumentParser()
    for name, dflt in (("seqs", 32), ("prompt", 256), ("response", 1792), ("vocab", 151936), ("micro", 4), ("iters", 3)):
        ap.add_argument(f"--{name}", type=int, default=dflt)
    ap.add_argument("--no-inplace-grad", action="store_true")
    a = ap.parse_args()
    print(json.dumps(measure(seqs=a.seqs, prompt=a.prompt, response=a.response, vocab=a.vocab, micro=a.micro, iters=a.iters,
                             inplace_grad=not a.no_inplace_grad)), flush=True)


if __name__ == "__main__":
    main()
