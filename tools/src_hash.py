"""Hash of the kernel + engine sources (what decides which kernels run and how): a committed rocprofv3 summary under profiles/
records the hash it was taken at, and bench.py flags (`profile_stale`) numbers read from a summary whose hash is not the tree's.
The C-ABI shim (api.cpp) and the host-side code (csrc/host, csrc/cli) launch nothing and are not part of it."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha(root=ROOT):
    h = hashlib.sha256()
    files = []
    for pat in ("supertonic_amd/csrc/*.hip", "supertonic_amd/csrc/*.inc", "supertonic_amd/csrc/*.hpp", "supertonic_amd/csrc/engine*.cpp", "Makefile"):
        files += glob.glob(os.path.join(root, pat))
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha())
