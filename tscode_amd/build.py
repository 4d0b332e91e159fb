"""Build libtscode_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m tscode_amd.build [--force] [--verbose]
"""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtscode_hip.so")
SOURCES = ["tscode_hip.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "tscode_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    # the digest of the sources the binary was built from is kept beside it (and inside it: tsc_build_digest); file times alone
    # say nothing after a checkout or a copy to another machine
    try:
        if open(OUT + ".digest").read().strip() != csrc_digest():
            return True
    except OSError:
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def csrc_digest():
    """SHA-256 (16 hex digits) over what the binary is made of -- the translation units, the headers they include (the public
    include/tscode_hip.h among them) and the compiler flags, names included: baked into the library (tsc_build_digest) so that a
    measurement can say which kernels the BINARY it ran was built from, not only which sources lie next to it.  Exactly these files:
    an editor's backup or a stray directory under csrc/ changes nothing."""
    import hashlib
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        path = os.path.join(CSRC, name)
        if not os.path.isfile(path):
            continue
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc()] + FLAGS + [f'-DTSC_CSRC_DIGEST="{csrc_digest()}"', "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(OUT + ".digest", "w") as f:
        f.write(csrc_digest() + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(OUT)
