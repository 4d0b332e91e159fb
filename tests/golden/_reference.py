"""Loader of the reference's own Python for the golden-vector generators.  BUILD CONTAINER ONLY.

Nothing here travels to the GPU box as something to run (`.gpurunignore` lists the generators), nothing is written
next to the reference (no bytecode either), and no reference source is copied: the modules are imported from
/root/reference and executed as plain NumPy.

Off-path packages the reference imports at module level but that this image lacks are registered in
``sys.modules`` as NAME-ONLY stand-ins before the import (the technique SURVEY.md 8c / Appendix D blessed for numba):

* ``numba``            njit / jit = identity decorator, prange = range, typed.List = list;
* ``rmsd``             kabsch* names, never called on the hot path;
* ``_tkinter``, ``cclib``, ``ase``, ``sella``, ``prettytable``, ``openbabel``   every attribute is an inert object;
* ``periodictable``    the reference builds ``tscode.pt.pt`` from it at import time.  The stand-in hands back an element
                       table holding DATA for the few elements the fixtures use (symbol, covalent radius, mass: the values
                       periodictable ships, Cordero 2008 radii / IUPAC 2005 masses, typed in below -- not checkable in this
                       image).  They only shape INPUTS of the hot path (bond graphs, orbital lengths, masses) and every such
                       input is recorded in the fixture next to the reference's outputs, so a wrong radius would change the
                       recorded inputs, never the pinned input -> output relation.
"""

from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

# Z: (symbol, covalent radius / A, mass / u)
ELEMENTS = {
    1: ("H", 0.31, 1.00794), 6: ("C", 0.76, 12.0107), 7: ("N", 0.71, 14.0067), 8: ("O", 0.66, 15.9994),
    9: ("F", 0.57, 18.9984032), 16: ("S", 1.05, 32.065), 17: ("Cl", 1.02, 35.453),
}


class _Inert:
    """An object every use of which is a no-op: attribute, call, index, iteration."""

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self

    def __getitem__(self, item):
        return self

    def __iter__(self):
        return iter(())


class _NameOnly(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


class _Element:
    def __init__(self, z):
        self.number = z
        self.symbol, self.covalent_radius, self.mass = ELEMENTS[z]


class _Table:
    def __init__(self, **_):
        self._e = {z: _Element(z) for z in ELEMENTS}

    def __getitem__(self, z):
        return self._e[int(z)]


def _name_only(name):
    m = _NameOnly(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install_standins(full=False):
    """Register the stand-ins and put the reference on sys.path.  ``full`` adds what tscode.utils / embeds / embedder /
    torsion_module / optimization_methods / hypermolecule_class need on top of the three hot-path modules."""
    sys.dont_write_bytecode = True

    class _Sig:
        def __getitem__(self, item):
            return self

        def __call__(self, *a, **k):
            return self

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs and not isinstance(args[0], _Sig):
            return args[0]
        return lambda f: f

    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.njit = nb.jit = njit
        nb.prange = range
        nb.float32, nb.float64, nb.boolean = np.float32, np.float64, np.bool_
        nb.int32 = _Sig()
        typed = types.ModuleType("numba.typed")

        class List(list):
            """numba.typed.List as a plain list.  Membership (`key in cache`, rmsd_pruning.py:65) is answered from a set kept
            beside the items -- the same answers as the list scan, without its O(len) cost, which is what lets the reference's
            own code run ensembles of 40 000 - 105 000 structures here (G16 / G17).  A List of Lists (unhashable items,
            rmsd_pruning.py:128) falls back to the scan."""

            def __init__(self, it=()):
                super().__init__(it)
                try:
                    self._set = set(self)
                except TypeError:
                    self._set = None

            def append(self, x):
                super().append(x)
                if self._set is not None:
                    try:
                        self._set.add(x)
                    except TypeError:
                        self._set = None

            def extend(self, it):
                it = list(it)
                super().extend(it)
                if self._set is not None:
                    try:
                        self._set.update(it)
                    except TypeError:
                        self._set = None

            def __contains__(self, x):
                if self._set is not None:
                    try:
                        return x in self._set
                    except TypeError:
                        pass
                return super().__contains__(x)

            def __setitem__(self, i, v):           # computed_pairs[chunk] = ...: item assignment invalidates the set
                super().__setitem__(i, v)
                self._set = None

        typed.List = List
        nb.typed = typed
        rmsd = types.ModuleType("rmsd")
        rmsd.kabsch_rotate = rmsd.kabsch = rmsd.kabsch_rmsd = None   # off-path names only
        sys.modules.update({"numba": nb, "numba.typed": typed, "rmsd": rmsd})
    if full and "periodictable" not in sys.modules:
        for name in ("_tkinter", "cclib", "cclib.io", "ase", "ase.calculators", "ase.calculators.calculator",
                     "ase.calculators.orca", "ase.calculators.gaussian", "ase.calculators.mopac", "ase.constraints",
                     "ase.dyneb", "ase.optimize", "ase.vibrations", "ase.neb", "ase.units", "ase.io", "ase.visualize",
                     "ase.gui", "ase.gui.gui", "ase.gui.images", "sella", "prettytable", "openbabel"):
            _name_only(name)
        sys.modules["_tkinter"].TclError = type("TclError", (Exception,), {})
        pt = _name_only("periodictable")
        core = _name_only("periodictable.core")
        core.PeriodicTable = _Table
        pt.core = core
    for p in (REFERENCE, REPO):
        if p not in sys.path:
            sys.path.insert(0, p)


def read_xyz_data(path):
    """An .xyz file read as data: (atomnos int[n], coords f64[n_frames, n, 3]) -- what the reference gets from cclib's
    ccread (Hypermolecule.__init__, tscode/hypermolecule_class.py:163-170)."""
    sym = {s: z for z, (s, _, _) in ELEMENTS.items()}
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines()]
    frames, atomnos, i = [], None, 0
    while i < len(lines) and lines[i].strip():
        n = int(lines[i].split()[0])
        rows = [ln.split() for ln in lines[i + 2:i + 2 + n]]
        atomnos = np.array([sym[r[0]] for r in rows])
        frames.append(np.array([[float(v) for v in r[1:4]] for r in rows]))
        i += 2 + n
    return atomnos, np.array(frames)


def ccread_like(path):
    """The two attributes of a cclib ccData object that Hypermolecule reads."""
    atomnos, coords = read_xyz_data(path)
    return types.SimpleNamespace(atomnos=atomnos, atomcoords=coords)
