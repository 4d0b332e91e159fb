"""G13: what tscode_amd's embed drop-ins read from the REFERENCE'S OWN objects (tests/dropin_reads.py has the recorder).

BUILD CONTAINER ONLY (listed in .gpurunignore): imports the reference through the stand-ins of tests/golden/_reference.py and the
helpers of gen_golden_embeds.py, builds the molecules of G11 case 0 (tests/string.txt: CH3Cl + HCOOH) and G12 case 0
(tests/cyclical.txt: 2 x C2H4) as real Hypermolecule objects, runs tscode_amd.embeds.string_embed / cyclical_embed on them with the
GPU call replaced by a stand-in, and writes which attributes were read and what they held to tests/golden/G13_dropin_reads.json.

Usage:  python -B tests/golden/gen_dropin_reads.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen_golden_embeds as G          # noqa: E402  (installs the stand-ins, imports the reference)
import dropin_reads                    # noqa: E402

import tscode_amd.embeds as E          # noqa: E402

T = G.TESTS
out = {}
# string embed: tests/string.txt
mols = [G._molecule(os.path.join(T, "CH3Cl.xyz"), [0], 2.5), G._molecule(os.path.join(T, "HCOOH.xyz"), [3], 2.5)]
e = G._embedder(mols, "string", clash_thresh=1.5)
e.systematic_angles = [n * 360 / 36 for n in range(36)]
e.candidates = 72
out["string_embed"] = dropin_reads.record(E.string_embed, e)
# cyclical embed: tests/cyclical.txt
C2H4 = os.path.join(T, "C2H4.xyz")
mols = [G._molecule(C2H4, [0, 3], 2.2), G._molecule(C2H4, [0, 3], 2.2)]
e = G._embedder(mols, "cyclical", clash_thresh=1.5, rigid=True)
for m in mols:
    G.ref_embedder.Embedder._set_pivots(e, m)
e.systematic_angles = G.ref_utils.cartesian_product(*[range(6) for _ in mols]) * 2 * 45 / 5 - 45
e.candidates = 0
out["cyclical_embed"] = dropin_reads.record(E.cyclical_embed, e)
path = os.path.join(HERE, "G13_dropin_reads.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path)
for k, v in out.items():
    print(k, {kind: sorted(names) for kind, names in v.items()})
