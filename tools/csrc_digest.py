#!/usr/bin/env python3
"""sha256 over the kernel sources (cuahn_vio_amd/csrc/*.h, *.hip, Makefile), 16 hex digits: names the BUILD a profile belongs to.
tools/profile_round.sh stores it next to the counter passes, tools/summarize_profile.py writes it into the tracked CSVs, and bench.py
quotes a committed traffic figure only when it carries the digest of the sources it is running (bench.py committed_traffic)."""
import glob
import hashlib
import os


def csrc_digest(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "cuahn_vio_amd", "csrc", "*"))):
        if f.endswith((".h", ".hip")) or os.path.basename(f) == "Makefile":
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_digest())
