"""Identity of the kernel sources a profile was taken on: sha256 over ml-hugs_amd/csrc/* and include/*.h (first 16 hex
digits).  The PMC summaries record it and bench.py attaches their figures to its JSON line only when it matches the
sources of the running build (otherwise: traffic = null + a 'stale' note)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "ml-hugs_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha16())
