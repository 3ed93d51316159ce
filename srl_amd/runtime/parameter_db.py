"""Checkpoint / parameter store on a filesystem, on-disk compatible with the reference's
``PytorchFilesystemParameterDB`` (``distributed/system/parameter_db.py:39-176`` interface, ``:176-324`` filesystem
implementation), plus the sample-staleness rule of the trainer worker.

Layout: ``<root>/<user_namespace>/<experiment>/<trial>/<policy_name>/<version>`` is a ``torch.save``'d checkpoint
``{"steps", "state_dict", "optimizer_state_dict"}`` (``mappo.py:58-66``); a tag is a symlink in the same
directory whose target is the version's file name; ``latest`` is re-pointed on every push.  Because this
package's checkpoints carry the reference's parameter names and shapes, a stock SRL policy worker or eval
manager pointed at the same directory loads what this trainer pushes, and vice versa.

Inside one job the trainer -> inference-GPU path does not go through here: parameters travel as one flat
buffer by RCCL broadcast (``ActorCriticPolicy.broadcast_parameters``).  This store is for resume, evaluation
and for mixing with stock SRL workers.
"""
import os
import shutil
import time
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch


class ParameterDBClient:
    """Same surface as the reference's abstract client (``parameter_db.py:39-176``)."""

    def __init__(self, experiment_name, trial_name):
        self.experiment_name = experiment_name
        self.trial_name = trial_name
        self.namespace = self.experiment_name + "/" + self.trial_name

    def list_names(self):
        raise NotImplementedError()

    def clear(self, name=None):
        raise NotImplementedError()

    def gc(self, name, max_untagged_version_count=None, max_untagged_version_ttl=None):
        raise NotImplementedError()

    def push(self, name, checkpoint, version: str, tags: Union[None, str, List[str]] = None,
             metadata: Dict[str, Any] = None) -> str:
        raise NotImplementedError()

    def tag(self, name, identifier, new_tag):
        raise NotImplementedError()

    def get(self, name, identifier="latest", block: bool = False, retry_times=60, mode="pytorch") -> Any:
        raise NotImplementedError()

    def list_versions(self, name) -> List[str]:
        raise NotImplementedError()

    def list_tags(self, name) -> List[Tuple[str, str]]:
        raise NotImplementedError()

    def has_tag(self, name, tag) -> bool:
        raise NotImplementedError()

    def version_of(self, name, identifier) -> int:
        raise NotImplementedError()


class FilesystemParameterDB(ParameterDBClient):

    def __init__(self, experiment_name, trial_name, root: str, user_namespace: str = "default"):
        super().__init__(experiment_name, trial_name)
        self.root = root
        self._workdir = os.path.join(root, user_namespace, experiment_name, trial_name)
        os.makedirs(self._workdir, exist_ok=True, mode=0o775)

    @staticmethod
    def purge(experiment_name, trial_name, root: str, user_namespace: str = "default"):
        d = os.path.join(root, user_namespace, experiment_name, trial_name)
        if os.path.exists(d):
            shutil.rmtree(d)

    # ------------------------------------------------------------------ paths
    def _path_of(self, name, identifier):
        return os.path.join(self._workdir, name, str(identifier))

    def _is_tag(self, name, tag):
        return os.path.islink(self._path_of(name, tag))

    def _list_all(self, name):
        d = os.path.join(self._workdir, name)
        return [f for f in os.listdir(d) if not f.endswith(".tmp")] if os.path.isdir(d) else []

    # ------------------------------------------------------------------ writes
    def push(self, name, checkpoint, version: str, tags=None, metadata=None):
        assert metadata is None, "metadata queries need the reference's MongoDB-backed store"
        version = str(version)
        path = self._path_of(name, version)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + ".tmp"
        torch.save(checkpoint, tmp)
        os.replace(tmp, path)  # readers never see a half-written file
        self.tag(name, version, "latest")
        for t in ([tags] if isinstance(tags, str) else (tags or [])):
            self.tag(name, version, t)
        return version

    def tag(self, name, identifier, new_tag):
        identifier = str(identifier)
        if self._is_tag(name, identifier):
            identifier = os.readlink(self._path_of(name, identifier))
        if not os.path.exists(self._path_of(name, identifier)):
            raise FileNotFoundError(f"no version `{identifier}` of policy `{name}`")
        tmp_path, new_path = self._path_of(name, new_tag + ".tmp"), self._path_of(name, new_tag)
        if os.path.lexists(tmp_path):
            os.remove(tmp_path)
        os.symlink(identifier, tmp_path)
        os.replace(tmp_path, new_path)  # atomic re-point

    def clear(self, name=None):
        shutil.rmtree(self._workdir if name is None else os.path.join(self._workdir, name))

    def gc(self, name, max_untagged_version_count=None, max_untagged_version_ttl=None):
        if max_untagged_version_ttl is not None:
            raise NotImplementedError()
        tagged = set(v for _, v in self.list_tags(name))
        untagged = [v for v in self.list_versions(name) if v not in tagged]
        doomed = untagged[:-max_untagged_version_count] if max_untagged_version_count else []
        for v in doomed:
            os.remove(self._path_of(name, v))
        return len(doomed)

    # ------------------------------------------------------------------ reads
    def _read(self, path, mode):
        if mode == "pytorch":
            return torch.load(path, map_location="cpu", weights_only=False)
        if mode == "bytes":
            with open(path, "rb") as f:
                return f.read()
        raise NotImplementedError(mode)

    def get(self, name, identifier="latest", block=False, retry_times=60, mode="pytorch"):
        """A checkpoint by version or tag.  Two things can stand between the caller and the file: it is not there yet
        (``block=True`` polls once a second, a writer may still be on its way) and it is being replaced under the reader
        (the writer renames a finished file over the tag: a read that loses that race fails with an OSError and is simply
        tried again a few milliseconds later).  Both draw on one budget of ``retry_times`` attempts."""
        path = self._path_of(name, identifier)
        missing = FileNotFoundError(f"Read checkpoint failed {name} {identifier}.")
        for _ in range(retry_times + 1):
            if os.path.lexists(path):
                try:
                    return self._read(path, mode)
                except FileNotFoundError:  # the tag points at a version that was collected: waiting will not bring it back
                    raise
                except OSError:
                    time.sleep(0.005)
            elif block:
                time.sleep(1)
            else:
                raise missing
        raise missing  # the budget is spent: the file never appeared (``block``), or every read lost its race

    def list_names(self):
        return [n for n in os.listdir(self._workdir) if len(self.list_tags(n)) > 0]

    def list_versions(self, name):
        rs = [v for v in self._list_all(name) if not self._is_tag(name, v)]
        rs.sort(key=lambda x: int(x))  # a version is a number of trainer steps
        return rs

    def list_tags(self, name):
        return [(t, os.readlink(self._path_of(name, t))) for t in self._list_all(name) if self._is_tag(name, t)]

    def has_tag(self, name, identifier):
        return os.path.lexists(self._path_of(name, identifier))

    def version_of(self, name, identifier) -> int:
        return self.get(name, identifier).get("steps", -1)


def sample_staleness(sample, policy_version: int, preemption_steps: float = float("inf")) -> Optional[Dict[str, float]]:
    """The trainer worker's admission rule (``trainer_worker.py:144-160``): ``None`` means "drop the sample"
    (no valid ``policy_version_steps``, or its oldest row is more than ``preemption_steps`` versions behind the
    policy); otherwise the numbers it logs next to the step statistics."""
    pv = sample.policy_version_steps
    if pv is None:
        return None
    pv = pv.cpu().numpy() if isinstance(pv, torch.Tensor) else np.asarray(pv)
    valid = pv[pv >= 0]
    if valid.size == 0:
        return None
    vmin, vavg = float(valid.min()), float(valid.mean())
    diff = policy_version - vmin
    if diff > preemption_steps:
        return None
    return dict(sample_min_policy_version=vmin, sample_version_difference=diff, staleness=policy_version - vavg)
