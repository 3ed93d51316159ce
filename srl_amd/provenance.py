"""What ties a recording under ``profiles/`` to the code it was taken from.  The GPU box has no ``.git`` (``gpurun`` ships a
snapshot), so a commit hash cannot be read where the counters are collected; a digest of the kernel sources can: the
recording scripts store it next to their tables and ``bench.py`` attaches a recording to its line only while the digest of
the sources it runs on is the same."""
import glob
import hashlib
import os


def kernel_sources_digest(root: str = None) -> str:
    """sha256 over every kernel source, header and the C ABI header (paths and contents, sorted), first 16 hex digits."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "srl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "srl_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
