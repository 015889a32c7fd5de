"""SHA-256 over the library's device/host sources (zkdl_amd/csrc/*.hip, *.hpp, include/zkdl_amd.h), in sorted order.
profiles/pmc_<round>.json records the fingerprint of the tree its counters were collected from; bench.py compares it with the
tree it runs from and flags `traffic_stale` when they differ (the GPU box has no .git to ask for ancestry)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_fingerprint():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "zkdl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "zkdl_amd", "csrc", "*.hpp")))
    files.append(os.path.join(ROOT, "include", "zkdl_amd.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_fingerprint())
