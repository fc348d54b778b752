#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the sources a profile depends on: the device and host code of libaugx.so and the Makefile.
Printed by profiles/run_profile.sh / run_pmc.sh into what they write; bench.py quotes a profile only when the hash of the tree it
runs in is the same (there is no .git on the GPU box)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha():
    h = hashlib.sha256()
    files = []
    for d in ("augustus_amd/csrc", "augustus_amd/csrc/device", "include"):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith((".h", ".hip", ".cc")):
                files.append(os.path.join(d, f))
    for f in files + ["Makefile"]:
        h.update(f.encode())
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha())
