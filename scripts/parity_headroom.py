"""How far the HIP path sits from the reference's golden vectors (tests/golden/steps_cnn.npz: down-sized NatureCNN, Atari
preset, 2 steps) with the bf16 matrix-core contractions and with the float32-MFMA kernels: max |parameter difference|
after the last step (test bound 2e-5) and the relative error of the loss terms (bound 1e-5)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
if mode == "f32":
    os.environ["SRL_MFMA"] = "f32"
    os.environ["SRL_OBS_BF16"] = "0"
import srl_amd
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_trainer import CASES

srl_amd.register_all()
for tag in ("cnn", "c1", "multi", "lstm"):
    pargs, targs, skw, n_steps, fname = CASES[tag]
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", fname))
    tr = trainer_api.make(config.Trainer("mappo", args=targs), config.Policy("actor-critic", args=pargs))
    names = list(g[f"{tag}_stat_names"])
    worst_stat = 0.0
    for step in range(n_steps):
        res = tr.step(synthetic.to_sample_batch(synthetic.make_sample_arrays(seed=100 + step, **skw)))
        ref = dict(zip(names, g[f"{tag}_step{step}_stats"]))
        for k in ("policy_loss", "value_loss", "entropy"):
            worst_stat = max(worst_stat, abs(res.stats[k] - ref[k]) / max(abs(ref[k]), 1e-2))
    sd = tr.policy.get_checkpoint()["state_dict"]
    worst = 0.0
    for key in g.files:
        pre_full, pre_s = f"{tag}_step{n_steps - 1}_param:", f"{tag}_step{n_steps - 1}_param_s97:"
        if key.startswith(pre_full):
            got = sd[key[len(pre_full):]].numpy()
        elif key.startswith(pre_s):
            got = sd[key[len(pre_s):]].numpy().reshape(-1)[::97]
        else:
            continue
        worst = max(worst, float(np.abs(got - g[key]).max()))
    print(f"[{mode}] {tag:6s}: max |param - golden| = {worst:.2e} (bound 2e-5); loss terms rel err {worst_stat:.2e} (bound 1e-5)")
