import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


# The kernel instantiations ONE 16 384-row chunk of the benchmarked update launches (srl_dispatch_tiles labels -> launches;
# heads and bias sums ride on the skinny family, which has no tile key).  tests/test_gpu_fullsize.py asserts a full-size step
# launches exactly these; tests/test_gpu_trainer.py asserts the 256-row step that is compared with the float32 / float64 oracles
# runs the same instantiations (same kernels and ring depths; only split factors may follow the row count).
BENCH_CHUNK_TILES = {
    "obs_fwd_bf16:k256:h2blk:split250": 1, "obs_bwd_bf16:k256:h2blk:split5": 1,
    "h2:conv:0:s3": 1, "h2:conv:1:s3": 1, "h2:conv:2:s2": 1, "h2:conv:3:s2": 1, "h2:wgrad:0:s2": 1, "h2:wgrad:1:s3": 1,
    "h2:gemm:4:s3": 1, "h2:gemm:8:s2": 1, "h2:tn:8:s4": 1,
}


def tile_kinds(tiles):
    """Tile labels without their split factors (``:k5`` / ``:split8``): the instantiation, whatever the row count."""
    import re
    return sorted({re.sub(r":(k|split)\d+(?=:|$)", "", k) if not k.startswith("obs") else re.sub(r":split\d+", "", k) for k in tiles})
