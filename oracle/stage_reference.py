"""Stage the reference's pure-torch hot-path files under oracle/_ref/ so that they travel to the GPU box (TEST INFRASTRUCTURE).

    python -m oracle.stage_reference            (also run by __graft_entry__.build() when /root/reference is present)

/root/reference does not exist on the GPU box, but `oracle/_ref/` does (git-ignored build output that gpurun ships like the
built .so).  The files are COPIED THERE BY THIS RECIPE, never committed: the registries (rlinf/algorithms/{registry,utils,
advantages,losses}.py), their two utility modules, the nested-dict helpers and the MLP policy with its modules -- all pure
torch, loadable behind the stub packages of oracle/reference_loader.py -- plus four files that are only ever PARSED (single
functions / classes compiled on their own: the model registry, the learner's run_training / train_micro_batch, the FSDP
manager's optimizer_step / build_optimizer).  Used for (1) the `-m gpu` test that runs
rlinf_amd.ext.register() against the REAL registry and calls the reference's own policy_loss / calculate_adv_and_returns, and
(2) bench.py's cpu_baseline with kind "reference"."""

from __future__ import annotations

import os
import shutil

from . import reference_loader as RL

SOURCE = "/root/reference"
DEST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def stage(verbose: bool = True) -> int:
    if not os.path.isfile(os.path.join(SOURCE, "rlinf/algorithms/advantages.py")):
        if verbose:
            print(f"[stage_reference] {SOURCE} not present: nothing staged (prebuilt oracle/_ref is used as it is)")
        return 0
    n = 0
    for rel in [rel for _, rel in RL._FILES_ALGO + RL._FILES_POLICY] + RL._FILES_SOURCE_ONLY:
        dst = os.path.join(DEST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(SOURCE, rel), dst)
        n += 1
    with open(os.path.join(DEST, "README"), "w") as f:
        f.write("Build output of oracle/stage_reference.py: unmodified copies of reference files, staged so that they travel to the\n"
                "GPU box. Git-ignored; never edit, never commit.\n")
    if verbose:
        print(f"[stage_reference] staged {n} reference files under {DEST}")
    return n


if __name__ == "__main__":
    stage()
