"""Runs the reference's OWN examples/embodiment/train_embodied_agent.py (argv[1]), unmodified, against rlinf_amd through the
``rlinf`` import alias -- in a clean interpreter (tests/test_reference_entry_point.py launches this file; other tests of the
suite load real reference modules under the name ``rlinf`` and must not share sys.modules with the alias).  Prints one JSON line."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from rlinf_amd import compat  # noqa: E402
from rlinf_amd._lib import RlxError  # noqa: E402
from rlinf_amd.config import load_config  # noqa: E402

entry, log_path = sys.argv[1], sys.argv[2]
compat.install_config_shims()
compat.install_alias()
import rlinf.workers.env.env_worker as aliased  # noqa: E402
import rlinf_amd.workers.env.env_worker as real  # noqa: E402

res = {"alias_is_same_module": aliased is real}
ns = {"__name__": "reference_train_embodied_agent", "__file__": entry}
exec(compile(open(entry).read(), entry, "exec"), ns)  # module body: imports, mp.set_start_method, the hydra-decorated main
main = ns["main"]
res["hydra_config_name"] = main.config_name
cfg_dir = os.path.join(ROOT, "examples", "embodiment", "config")
cfg = load_config(os.path.join(cfg_dir, "maniskill_ppo_mlp.yaml"), search_paths=[cfg_dir], overrides=[
    "env.train.total_num_envs=16", "env.train.max_steps_per_rollout_epoch=8", "actor.global_batch_size=64",
    "actor.micro_batch_size=64", "algorithm.update_epoch=1", f"runner.logger.log_path={log_path}",
    "runner.logger.experiment_name=t", "runner.max_epochs=2"])
buf = io.StringIO()
try:
    with redirect_stdout(buf):
        main(cfg)
    res["outcome"] = "ran"
except RlxError as e:  # without a GPU: the first kernel entry of runner.run() refuses, loudly
    res["outcome"], res["error"] = "rlx_error", str(e)[:300]
from rlinf_amd.workers.common import peer  # noqa: E402

res["config_dumped"] = '"loss_type": "actor_critic"' in buf.getvalue()
res["workers_initialised"] = bool(peer("actor").model is not None and peer("env").buffer is not None and peer("rollout") is not None)
res["cuda"] = torch.cuda.is_available()
res["actor_steps"] = int(peer("actor").optimizer_steps)
print(json.dumps(res))
