"""Actor-side trajectory GAE, registered as trajectory post-processor ``"gae"``.

Mirror of ``TrajGAE`` (reference ``legacy/algorithm/modules/gae.py:100-142``).  This runs inside the actor
worker (a host Python process stepping environments, out of scope to accelerate: SURVEY.md section 2) on one
finished episode at a time, a list of per-step ``SampleBatch`` with ``[1]``-shaped leaves, so it is plain
numpy on the host like the reference; the batched device-side scan is ``srl_gae_scan``.
"""
import numpy as np

from srl_amd.api import trainer as trainer_api


class TrajGAE(trainer_api.TrajPostprocessor):

    def __init__(self, gamma, lmbda):
        self.gamma = gamma
        self.lmbda = lmbda

    def process(self, memory):
        last = memory[-1]
        assert np.logical_or(last.done, last.truncated).all()
        # bootstrap from the final observation's value only if the episode was cut by a time limit (:121-127)
        if last.analyzed_result is None:
            next_value = 0
        else:
            next_value = last.analyzed_result.value * last.truncated
        running = np.zeros_like(memory[0].reward)
        for step in reversed(memory[:-1]):
            value = step.analyzed_result.value
            delta = step.reward + self.gamma * next_value - value
            running = self.gamma * self.lmbda * running + delta
            step.analyzed_result.adv = running
            step.analyzed_result.ret = running + value
            next_value = value
        return memory


trainer_api.register_traj_postprocessor('gae', TrajGAE)
