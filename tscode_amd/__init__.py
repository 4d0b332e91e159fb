"""tscode_amd -- MI355X (gfx950) engine for TSCoDe's geometry hot path.

Drop-in mirrors of the reference's functions (same names and signatures):

    from tscode_amd import prune_conformers_rmsd, compenetration_check, get_embed
    import tscode_amd; tscode_amd.install()      # patch an already imported tscode

Everything numeric runs in libtscode_hip.so (hand-written HIP kernels behind the C ABI of
include/tscode_hip.h).  There is no CPU fallback; importing this package does not touch the GPU.
"""

from .algebra import (align_vec_pair, all_dists, norm, norm_of, quaternion_to_rotation_matrix,  # noqa: F401
                      rot_mat_from_pointer, rotation_matrix_from_vectors, transform_coords, vec_angle)
from .embeds import (EmbedTrace, cyclical_embed_batch, cyclical_embed_params, embed_batch, filter_angular_groups, get_embed,  # noqa: F401
                     string_embed_batch, string_embed_params, string_embed_poses)
from .utils import TriangleError, cartesian_product, polygonize  # noqa: F401
from .engine import Engine, FragmentSet, device_count, get_engine  # noqa: F401
from .install import install, uninstall  # noqa: F401
from .numba_functions import (_get_tf_mat, compenetration_check, compenetration_mask, count_clashes, get_torsion_fingerprint,  # noqa: F401
                              prune_conformers_tfd, tfd_similarity)
from .optimization_methods import (_score_embed_poses, fitness_check, fitness_mask, get_inertia_moments,  # noqa: F401
                                   get_moi_similarity_matches, prune_by_moment_of_inertia)
from .torsion_module import csearch_candidates, csearch_rotate, rotate_dihedral, rotate_dihedral_batch, torsion_comp_check  # noqa: F401
from .rmsd_pruning import _rmsd_similarity, last_prune_stats, prune_conformers_rmsd, rmsd_and_max_numba  # noqa: F401

__version__ = "0.1.0"
