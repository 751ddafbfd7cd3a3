"""Identity of the kernel sources a measurement belongs to: sha256 over the HIP/C++ sources of libmhx (sorted by name).
The PMC traffic file under profiles/ records the id of the tree it was collected on; bench.py only reports
`roofline.traffic` when the running tree has the same id (a counter measurement of other kernels is not evidence)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_id():
    """sha256 of the built library itself (what actually runs); None when it is not built"""
    try:
        h = hashlib.sha256()
        with open(os.path.join(ROOT, "megahit_amd", "libmhx.so"), "rb") as fh:
            for chunk in iter(lambda: fh.read(1 << 20), b""):
                h.update(chunk)
        return h.hexdigest()[:16]
    except OSError:
        return None


def build_id():
    h = hashlib.sha256()
    src = os.path.join(ROOT, "megahit_amd", "csrc")
    files = sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
