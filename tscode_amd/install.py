"""Patch a live TSCoDe so that its hot path runs on the MI355X engine.

TSCoDe binds the hot-path functions by name at import time (``from tscode.rmsd_pruning import
prune_conformers_rmsd`` in embedder.py:56, operators.py:39, optimization_methods.py:32,
atropisomer_module.py:33; ``compenetration_check`` in embedder.py:48, embeds.py:28; ...), so the
replacement has to be set on every importing module, not only on the defining one.
This module never imports tscode itself: it only touches modules already in sys.modules.
"""

from __future__ import annotations

import sys

from . import algebra, embeds, numba_functions, optimization_methods, rmsd_pruning, torsion_module

# attribute -> (replacement, modules that bind it)
_PATCHES = {
    "prune_conformers_rmsd": (rmsd_pruning.prune_conformers_rmsd,
                              ("tscode.rmsd_pruning", "tscode.embedder", "tscode.operators", "tscode.optimization_methods",
                               "tscode.atropisomer_module")),
    "rmsd_and_max_numba": (rmsd_pruning.rmsd_and_max_numba, ("tscode.rmsd_pruning", "tscode.automep")),
    "_rmsd_similarity": (rmsd_pruning._rmsd_similarity, ("tscode.rmsd_pruning", "tscode.embeds")),
    "compenetration_check": (numba_functions.compenetration_check, ("tscode.numba_functions", "tscode.embedder", "tscode.embeds")),
    "count_clashes": (numba_functions.count_clashes, ("tscode.numba_functions", "tscode.embedder")),
    "get_embed": (embeds.get_embed, ("tscode.embeds",)),
    "all_dists": (algebra.all_dists, ("tscode.algebra", "tscode.numba_functions", "tscode.graph_manipulations")),
    "transform_coords": (algebra.transform_coords, ("tscode.algebra",)),
    # (any real angle: tscode/torsion_module.py:984-1005 calls it with fractional corrections; the candidate loops of the
    # conformational search belong on tscode_amd.csearch_rotate, which takes their integer tables whole)
    "rotate_dihedral": (torsion_module.rotate_dihedral, ("tscode.utils", "tscode.torsion_module")),
    "torsion_comp_check": (torsion_module.torsion_comp_check, ("tscode.numba_functions", "tscode.torsion_module")),
    "get_moi_similarity_matches": (optimization_methods.get_moi_similarity_matches, ("tscode.algebra", "tscode.optimization_methods")),
    "_score_embed_poses": (optimization_methods._score_embed_poses, ("tscode.numba_functions",)),
    "fitness_check": (optimization_methods.fitness_check, ("tscode.optimization_methods", "tscode.embedder")),
    "prune_conformers_tfd": (numba_functions.prune_conformers_tfd,
                             ("tscode.numba_functions", "tscode.embedder", "tscode.operators", "tscode.torsion_module")),
    # the embed loops themselves (tscode/embedder.py:39-40 binds both names; generate_candidates looks them up at call time, :1141-1154)
    "string_embed": (embeds.string_embed, ("tscode.embeds", "tscode.embedder")),
    "cyclical_embed": (embeds.cyclical_embed, ("tscode.embeds", "tscode.embedder")),
}

# Functions that take a whole ensemble per call: what install() replaces by default.  The others are called by TSCoDe once per
# pose / pair from Python loops; a GPU call (upload, launch, download, synchronise) takes 35-185 us against the few microseconds of
# the reference's jitted function (tools/dropin_latency.py), so replacing them would SLOW those loops down -- they are drop-in
# equivalents for checking and for callers that move to the batched forms, patched only on request.
_WHOLE_ENSEMBLE = ("prune_conformers_rmsd", "prune_conformers_tfd", "get_moi_similarity_matches", "_score_embed_poses", "string_embed",
                   "cyclical_embed")

_saved = {}


def install(modules=None, per_item=False):
    """Replace the hot-path functions in every already-imported tscode module: by default those that work on a whole ensemble
    per call (prune_conformers_rmsd, prune_conformers_tfd, get_moi_similarity_matches, _score_embed_poses) and the two embed
    loops (string_embed, cyclical_embed: one GPU call each instead of one Python iteration per pose); with
    ``per_item=True`` also the per-pose / per-pair ones (compenetration_check, get_embed, rmsd_and_max_numba, ...), which are
    equivalent but slower than the reference's jitted code when called one item at a time.
    Returns the list of (module, attribute) pairs that were patched."""
    mods = sys.modules if modules is None else modules
    done = []
    for attr, (fn, names) in _PATCHES.items():
        if not per_item and attr not in _WHOLE_ENSEMBLE:
            continue
        for name in names:
            mod = mods.get(name)
            if mod is not None and hasattr(mod, attr):
                _saved.setdefault((name, attr), getattr(mod, attr))
                if attr in ("string_embed", "cyclical_embed") and name == "tscode.embeds":
                    embeds._originals[attr] = _saved[(name, attr)]      # what the drop-in hands the cases it does not cover to
                setattr(mod, attr, fn)
                done.append((name, attr))
    return done


def uninstall(modules=None):
    mods = sys.modules if modules is None else modules
    for (name, attr), fn in list(_saved.items()):
        mod = mods.get(name)
        if mod is not None:
            setattr(mod, attr, fn)
        del _saved[(name, attr)]
    embeds._originals.clear()
