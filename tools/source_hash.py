"""sha256 over the sources that determine what the GPU executes (kernels, C ABI, host sequencing, bench.py).
Profiles under profiles/ carry this hash; bench.py only quotes a counter file whose hash matches the tree it runs from
(the GPU box has no .git, so a commit id cannot be checked there)."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ("aldi_amd/csrc/*.hip", "aldi_amd/csrc/*.h", "aldi_amd/csrc/*.cpp", "aldi_amd/csrc/Makefile", "include/*.h", "aldi_amd/*.py",
            "aldi_amd/detr/*.py", "bench.py")


def source_hash(root: str = ROOT) -> str:
    h = hashlib.sha256()
    for pat in PATTERNS:
        for f in sorted(glob.glob(os.path.join(root, pat))):
            h.update(os.path.relpath(f, root).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    sys.stdout.write(source_hash() + "\n")
