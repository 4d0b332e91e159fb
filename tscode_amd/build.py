"""Build libtscode_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m tscode_amd.build [--force] [--verbose]
"""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")           # object files and their dependency lists (git-ignored; travels to the GPU box, unused there)
OUT = os.path.join(HERE, "libtscode_hip.so")
# One translation unit per kernel family: an edit rebuilds the units that include what changed (hipcc -MD), side by side.
SOURCES = ["ctx.hip", "embed.hip", "prune.hip", "pairs_tile.hip", "pairs_sieve.hip", "pairs_sieve_plain.hip", "pairs_sorted.hip", "pairs_mm.hip", "adjacent.hip", "pipeline.hip", "xchg.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "tscode_hip.h")]
DIGEST_UNIT = "ctx.hip"                     # the unit that bakes the digest in (tsc_build_digest): rebuilt whenever anything changes


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
        return open(OUT + ".digest").read().strip() != csrc_digest()
    except OSError:
        return True


CFLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
          "-Wno-cuda-compat"]   # (-Wcuda-compat: "inline" on a kernel -- which is what lets a header's kernels be included by several units)
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared"]
FLAGS = CFLAGS + LDFLAGS


def _sha(paths, extra=""):
    import hashlib
    h = hashlib.sha256()
    for path in paths:
        if not os.path.isfile(path):
            continue
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    h.update(extra.encode())
    return h.hexdigest()[:16]


def csrc_digest():
    """SHA-256 (16 hex digits) over what the binary is made of -- the translation units, the headers they include (the public
    include/tscode_hip.h among them) and the compiler flags, names included: baked into the library (tsc_build_digest) so that a
    measurement can say which kernels the BINARY it ran was built from, not only which sources lie next to it.  Exactly these files:
    an editor's backup or a stray directory under csrc/ changes nothing."""
    return _sha([os.path.join(CSRC, name) for name in SOURCES + HEADERS], " ".join(FLAGS))


def _unit_deps(src):
    """The files a unit was last compiled from (hipcc -MD): its source and the project headers it includes, transitively."""
    dep = os.path.join(OBJ, src + ".d")
    try:
        words = open(dep).read().replace("\\\n", " ").split()
    except OSError:
        return None
    root = os.path.dirname(HERE)
    return sorted({os.path.normpath(w) for w in words[1:] if os.path.normpath(w).startswith(root)})


def _unit_stale(src, digest):
    obj, stamp = os.path.join(OBJ, src + ".o"), os.path.join(OBJ, src + ".sha")
    deps = _unit_deps(src)
    if deps is None or not os.path.exists(obj):
        return True
    want = _sha(deps, " ".join(CFLAGS) + (digest if src == DIGEST_UNIT else ""))
    try:
        return open(stamp).read().strip() != want
    except OSError:
        return True


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    digest = csrc_digest()
    stale = [s for s in SOURCES if force or _unit_stale(s, digest)]
    procs = []
    for src in stale:
        cmd = [_hipcc()] + CFLAGS + ["-c", "-MD", "-MF", os.path.join(OBJ, src + ".d"), "-o", os.path.join(OBJ, src + ".o"), os.path.join(CSRC, src)]
        if src == DIGEST_UNIT:
            cmd.insert(1, f'-DTSC_CSRC_DIGEST="{digest}"')
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd)))     # (nine units: side by side)
    failed = [src for src, pr in procs if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, "hipcc -c " + " ".join(failed))
    for src in stale:
        with open(os.path.join(OBJ, src + ".sha"), "w") as f:
            f.write(_sha(_unit_deps(src), " ".join(CFLAGS) + (digest if src == DIGEST_UNIT else "")) + "\n")
    subprocess.run([_hipcc()] + LDFLAGS + ["-o", OUT] + [os.path.join(OBJ, s + ".o") for s in SOURCES], check=True)
    with open(OUT + ".digest", "w") as f:
        f.write(digest + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(OUT)
