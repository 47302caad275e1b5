#!/usr/bin/env python3
"""sha256 over the kernel sources (inferbiomechanics_amd/csrc/*.hip, *.h, Makefile), in sorted file order.

profiles/traffic.json stores the hash of the build its counters were collected on (tools/profile_round.sh writes it
next to the collection on the GPU box); bench.py recomputes it and says `traffic_stale` when the kernels have changed since.
"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, "inferbiomechanics_amd", "csrc")
    files = sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + [os.path.join(d, "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
