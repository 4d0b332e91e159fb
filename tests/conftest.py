import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


LARGE_PRUNE_CASES = ("G16a", "G16b", "G17")
_large = {}


def load_large_prune(name):
    """G16a / G16b / G17 (tests/golden/gen_golden_large.py): the reference's own prune_conformers_rmsd on 40 023 / 41 999 / 104 999
    structures.  The fixture stores the generator's parameters and the digest of the heavy-atom array they produce, not the
    array: it is regenerated here (tscode_amd.synthetic, deterministic) and checked against the digest first -- a mismatch means the
    inputs differ from what the reference saw and nothing below it would be a parity statement.
    Returns a namespace: structures, atomnos, heavy, thr, and the recorded g (mask_bits, ks, pass_mask_bits, pass_nkeys, keys)."""
    if name not in _large:
        import hashlib
        import types

        from tscode_amd.synthetic import make_ensemble
        g = load_golden(name + "_prune_large")
        ens = make_ensemble(int(g["n"]), tuple(int(a) for a in g["atoms_per_frag"]), seed=int(g["seed"]), children=int(g["children"]),
                            local_spread=float(g["local_spread"]))
        structures = ens.poses()
        heavy = np.ascontiguousarray(structures[:, ens.atomnos != 1])
        digest = hashlib.sha256(heavy.tobytes()).digest()
        assert digest == g["heavy_sha256"].tobytes(), (f"{name}: the regenerated ensemble is not the one the reference ran on "
                                                       f"(sha256 {digest.hex()[:16]} vs recorded {g['heavy_sha256'].tobytes().hex()[:16]})")
        _large[name] = types.SimpleNamespace(structures=structures, atomnos=ens.atomnos, heavy=heavy, thr=float(g["thr"]), n=int(g["n"]), g=g)
    return _large[name]


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    # no more threads than CPUs this process may run on (an OpenMP runtime that sizes its team by the machine, on a box that
    # grants 16 of 256 hardware threads, turns every small parallel region into milliseconds)
    orc.set_num_threads(max(1, min(orc.num_threads(), len(os.sched_getaffinity(0)), 32)))
    return orc
