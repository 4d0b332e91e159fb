"""G14: which modules of the reference bind which hot-path names -- what tscode_amd.install() has to patch.
G15: the reference's rotate_dihedral with FRACTIONAL angles (tscode/utils.py:389-414 as tscode/torsion_module.py:984-1005 calls it).

BUILD CONTAINER ONLY (listed in .gpurunignore): imports every module of the reference that the stand-ins of
tests/golden/_reference.py make importable, then, for every name in tscode_amd.install._PATCHES (+ rotate_dihedral), records the
modules whose namespace holds that name bound to the SAME object as the defining module's -- TSCoDe binds the hot-path functions
by name at import time (`from tscode.rmsd_pruning import prune_conformers_rmsd`), so each of these is a site a replacement has
to be set on.  tests/test_abi_and_host.py compares install.py's hand-written table with the fixture.

Usage:  python -B tests/golden/gen_install_sites.py
"""
import importlib
import json
import os
import pkgutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _reference as R  # noqa: E402

R.install_standins(full=True)
import networkx as nx  # noqa: E402
if not hasattr(nx, "from_numpy_matrix"):          # (renamed in networkx 3: the reference calls the old name at import time nowhere, at run time in graph code)
    nx.from_numpy_matrix = nx.from_numpy_array

import tscode  # noqa: E402

sys.path.insert(0, R.REPO)
from tscode_amd.install import _PATCHES  # noqa: E402

imported, failed = {}, {}
for info in pkgutil.iter_modules(tscode.__path__):
    name = f"tscode.{info.name}"
    if info.name in ("__main__", "tests", "run_tests"):    # (entry points: importing them runs the program)
        continue
    try:
        imported[name] = importlib.import_module(name)
    except BaseException as e:  # noqa: BLE001  (SystemExit from modules that parse argv, ImportError from off-path packages)
        failed[name] = f"{type(e).__name__}: {e}"[:200]

names = sorted(set(_PATCHES) | {"rotate_dihedral"})
sites = {}
for attr in names:
    owners = [m for m in imported.values() if getattr(getattr(m, attr, None), "__module__", None) == m.__name__]
    if not owners:
        sites[attr] = {"defined_in": None, "bound_in": []}
        continue
    obj = getattr(owners[0], attr)
    sites[attr] = {"defined_in": owners[0].__name__,
                   "bound_in": sorted(n for n, m in imported.items() if getattr(m, attr, None) is obj)}
out = {"modules_imported": sorted(imported), "modules_not_importable_here": failed, "sites": sites}
path = os.path.join(HERE, "G14_install_sites.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path)
for a, s in sites.items():
    print(f"  {a:28s} defined in {s['defined_in']}; bound in {s['bound_in']}")
print("not importable here:", failed)

# ---- G15: rotate_dihedral with fractional angles, and the rotate / undo sequence of torsion_module.py:984-1005
ref_utils = imported["tscode.utils"]
rng = np.random.default_rng(1405)
coords = rng.normal(size=(14, 3)) * 2.0
dihedral = np.array([3, 5, 6, 11])
mask = np.zeros(14, bool)
mask[[0, 1, 2, 3, 9]] = True
angles = np.array([12.5, -0.37, 123.456, 359.9, 1e-3, -179.99, 0.5, 90.25])
data = {"coords": coords, "dihedral": dihedral, "mask": mask, "angles": angles,
        "out_mask": np.array([ref_utils.rotate_dihedral(coords.copy(), list(dihedral), a, mask=mask) for a in angles]),
        "out_first": np.array([ref_utils.rotate_dihedral(coords.copy(), list(dihedral), a) for a in angles])}
seq = coords.copy()
trail = []
for a in angles[:4]:      # rotate, look, rotate back: the correction search of torsion_module.py:984-1000
    seq = ref_utils.rotate_dihedral(seq, list(dihedral), a, mask=mask)
    trail.append(seq.copy())
    seq = ref_utils.rotate_dihedral(seq, list(dihedral), -a, mask=mask)
    trail.append(seq.copy())
data["trail"] = np.array(trail)
p15 = os.path.join(HERE, "G15_rotate_dihedral_fractional.npz")
np.savez_compressed(p15, **data)
print("wrote", p15)
