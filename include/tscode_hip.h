/*
 * tscode_hip.h -- C ABI of libtscode_hip.so, the MI355X (gfx950) engine for TSCoDe's geometry hot path.
 *
 * The reference (ntampellini/TSCoDe v0.4.16) is pure Python + Numba: the path has no FFI or plugin
 * table, callers bind plain Python functions by name.  Each entry point below states the
 * reference function (file:line under the reference root) whose work it takes over; the Python
 * mirror that keeps the reference's call signatures is tscode_amd/ (see INTEGRATION.md for the
 * binding a TSCoDe maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types cross the boundary.
 *   - every function returns 0 on success and a negative tsc_status on failure; tsc_last_error()
 *     returns a thread-local, human-readable message for the last failure on the calling thread.
 *   - there is NO CPU path in this library: without a usable HIP device every call fails with
 *     TSC_ERR_NO_DEVICE.
 *   - all coordinates are C-contiguous float64 (Angstrom), index arrays int32 unless stated.
 *   - "host" entry points take host pointers, copy in/out and synchronise before returning;
 *     "_dev" entry points take device pointers valid on the context's device, enqueue on the
 *     context's stream and return without synchronising unless stated.
 *   - the caller owns every buffer it passes; the library keeps no pointer after return except
 *     inside a tsc_prune object, which borrows `heavy` until tsc_prune_destroy.
 *   - MULTI-GPU, a deliberate deviation from SURVEY.md 8(b): that sketch has "multi-GPU variants take a device list / an RCCL
 *     communicator held in tsc_ctx".  This library opens no communicator and spawns no process.  The path shards as one
 *     process per GPU, each with a context of its own, and the exchange steps (one all-gather of the surviving heavy-atom
 *     shards -- or none, when every rank embeds all poses itself: the host times both forms on its node -- and one
 *     all-reduce(MIN) over best[] per sharded pass) belong to the HOST that owns the process group: in this
 *     repository torch.distributed over RCCL (tscode_amd/pipeline.py::sharded_step), in a C host ncclAllGather /
 *     ncclAllReduce on the same device pointers.  What the C ABI provides for it is the part only the library can do: a
 *     rank's block of poses (tsc_embed_clash_compact_dev), the stepping form of the prune with the row tiles of a pass dealt to
 *     (rank, world_size) and best[] in a caller-owned buffer the collective can run on (tsc_prune_create ..
 *     tsc_prune_pass_local(rank, world_size) .. tsc_prune_use_best_buffer .. tsc_prune_pass_finish), and
 *     tsc_ctx_set_stream, so that kernels and collectives are ordered on one stream.  Linking RCCL into the library would tie
 *     it to one launcher and one communicator lifetime for no kernel's benefit.
 *   - a context (tsc_ctx: one device, its streams, its scratch cache) is for ONE thread at a time, like the reference's
 *     callers (single-threaded Python; multiembed.py uses processes): threads that want to work concurrently create a
 *     context each.  Any number of processes may use a device at once (tests/: three processes stepping one GPU).
 */
#ifndef TSCODE_HIP_H
#define TSCODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSC_VERSION 100 /* 0.1.0 */

typedef enum {
    TSC_OK = 0,
    TSC_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, unsupported shape) */
    TSC_ERR_NO_DEVICE = -2, /* no HIP device / HIP runtime failure at context creation */
    TSC_ERR_HIP = -3,       /* a HIP call failed; message has the HIP error string */
    TSC_ERR_NOMEM = -4,     /* device or host allocation failed */
    TSC_ERR_STATE = -5      /* call sequence violated (prune stepping API) */
} tsc_status;

typedef struct tsc_ctx tsc_ctx;     /* one per (process, device); owns a stream and scratch memory */
typedef struct tsc_prune tsc_prune; /* state of one prune_conformers_rmsd run (stepping API) */

/* ---- library / context ------------------------------------------------------------------ */
int tsc_version(void);
const char *tsc_last_error(void);
/* SHA-256 (16 hex digits) of the kernel sources this binary was built from (tscode_amd/build.py passes it to the compiler;
 * "unrecorded" for a build by other means): lets a measurement tie its numbers to the binary that ran, not to the sources beside it. */
const char *tsc_build_digest(void);
int tsc_device_count(void); /* >= 0, or a negative tsc_status */
int tsc_ctx_create(int device, tsc_ctx **out);
int tsc_ctx_destroy(tsc_ctx *ctx);
/* Run on a caller-provided hipStream_t; NULL = the library's own (non-blocking) stream.  Note that PyTorch's DEFAULT stream has
 * the handle 0 = NULL: to order this library's kernels with torch work (copies, RCCL collectives) make an explicit
 * torch.cuda.Stream current and pass its .cuda_stream here (tscode_amd/pipeline.py does). */
int tsc_ctx_set_stream(tsc_ctx *ctx, void *hip_stream);
int tsc_ctx_synchronize(tsc_ctx *ctx);
/* Tunables.  "prune_algo": 0 = automatic (default), 1 = register-tiled all-pairs kernel (<= 32 heavy atoms),
 * 2 = descriptor sieve (any size);  "seg_cols": columns per pair-kernel work item (multiple of 256, at most 4096; 0 = automatic);
 * "drain_min": queued pairs that trigger an evaluation batch in the sieve kernel (1..64, default 32);
 * "sieve_cpl": columns per lane of the sieve kernel's screen, 1, 2 (default) or 4 -- register footprint against occupancy;
 * "early_basis": 1 (default) lets tsc_pipeline_dev estimate the descriptor basis of the prune from a sample of the unfiltered
 * poses on a side stream while the clash kernel runs (the choice of basis never changes a verdict); 0 = from the filtered
 * structures, on the main stream.  "fuse_descriptors": 1 (default) then lets the kernel that embeds the passing poses write their
 * descriptors as well (poses of up to about 80 heavy atoms; otherwise and with 0 a separate launch reads the coordinates back).
 * "pca_min_n": ensembles smaller than this (default 6000) take the identity basis for their descriptors instead of estimated
 * principal axes (three launches and about 45 us less per run; any basis gives the same verdicts).
 * "local_pass": 1 (default) lets passes whose longest chunk has at most "local_max_chunk" (default 384, up to 2048) structures
 * run in the one-launch chunk-local kernel; "sieve_trim": 1 (default) = the screen's shorter instruction sequence;
 * "sieve_mm": the pair kernels with the descriptor screen on the matrix cores and 64 rows per work item (csrc/mm.hpp, cull_mm.hpp):
 * 0 never, 1 (default) in runs of at least "mm_min_n" structures (default 100000), 2 always; "sieve_mm16": 1 (default) = smaller runs
 * take the matrix-core screen on 16-row work items (k_rmsd_sieve_mm16), 0 = the packed-fp32 screen; "mm_seg_cols": columns per work
 * item of the walked passes' 64-row kernel (a multiple of 64 up to 1024; 0 = automatic);
 * "fused_apply": 1 (default) lets the sieve kernel of a single-rank pass apply a row tile's verdicts itself when the tile's last
 * work item finishes and close the pass (two launches per pass); 0 = tsc_prune_pass_finish launches k_apply_pass (always so for the
 * register-tiled kernel and for passes searched by several ranks);  "open_lds_blocks": scan blocks (2048 structures each) up to which
 * the per-row kernel stages their prefix in LDS (default: its capacity, 2048; 0 = always read it from memory; tests);  "clash_fp32": 1 (default)
 * decides verdict-only clash masks by a packed-fp32 minimum with fp64 fallback;
 * "cull": 1 (default) lets the large passes of the sieve (at least "cull_min_pairs" = n (n / k) / 2 pairs, default 2e9, fewer than 64
 * chunks) lay their active structures out along a Morton curve of the descriptors and skip the tile pairs whose bounding boxes lie
 * beyond the screen's limit, where the rows' ranges are long enough for that to pay (decided per pass on the device, one
 * synchronisation); 0 = never, 2 = every such pass (tests).  "deterministic_basis": 1 = the descriptor basis from fixed-order sums, so
 * that every rank of a sharded run derives bit-identical descriptors and hence the same layout (default 0: atomics, 35 us faster).
 * A run remembers which of the two it was created under: only a run created under 1 culls a pass whose ROW TILES are dealt to several
 * ranks (tsc_prune_pass_local / tsc_prune_pass_rows with world_size > 1 -- the ranks deal the tiles of ONE sorted layout); a run created
 * under 0 walks such a pass in index order, and refuses it (TSC_ERR_STATE) when it had itself chosen the all-pairs kernel from its own
 * basis estimate ("prune_algo" 0), a choice that ranks with different estimates could make differently;
 * "cull_tile_block" (256): a culled pass dealt to several ranks by row tiles (tsc_prune_pass_local(rank, world)) gives a rank runs of this many
 * consecutive tiles of the sorted layout -- neighbours on the curve share their columns, and a row's early exit knows more of what was found;
 * "stage1_f32": the pair kernels' first look at a pair that passed the screen (H = p^T q and the quartic tests) reads a float32 copy
 * of the coordinates with the rounding bound that goes with it, the float64 coordinates only for what that leaves undecided: 0 = never,
 * 1 (default) = in runs with 128 MB of heavy-atom coordinates or more (where the gathers come from HBM), 2 = always;
 * "pass_timing": HIP events for tsc_pass_stats.gpu_ms / tile_ms and the pipeline's stage timings: 0 = none (default; an
 * event record in the stream costs about 4 us on MI355X), 1 = the pair kernel's own start/stop events (tile_ms; passes run by the
 * chunk-local kernel carry theirs at level 2 only), 2 = also around every whole pass (gpu_ms) and the stages of tsc_pipeline_dev. */
int tsc_ctx_set_option(tsc_ctx *ctx, const char *name, double value);
/* Device memory helpers for hosts that do not bring their own allocator (tests, C callers). */
int tsc_malloc(tsc_ctx *ctx, size_t bytes, void **dptr);
int tsc_free(tsc_ctx *ctx, void *dptr);
int tsc_memcpy_h2d(tsc_ctx *ctx, void *dst, const void *src, size_t bytes); /* synchronous */
int tsc_memcpy_d2h(tsc_ctx *ctx, void *dst, const void *src, size_t bytes); /* synchronous */
/* HIP-event timing of everything enqueued on the context's stream between begin and end (ms). */
int tsc_timer_begin(tsc_ctx *ctx);
int tsc_timer_end(tsc_ctx *ctx, float *elapsed_ms); /* synchronises */

/* ---- K1: batched rigid-body embedding ------------------------------------------------------
 * Replaces get_embed (tscode/embeds.py:961-969) and transform_coords (tscode/algebra.py:390-400),
 * batched over poses:  out[s] = concat_m ( rot[s,m] @ X_m[conf_idx[s,m]].T ).T + pos[s,m].
 *   frags      f64, all fragments back to back; fragment m is [n_conf[m], n_atoms[m], 3] at frags + frag_off[m]
 *   frag_off   i64[n_mols] offsets into frags, in doubles
 *   conf_idx   i32[n_poses, n_mols];  rot f64[n_poses, n_mols, 3, 3];  pos f64[n_poses, n_mols, 3]
 *   out        f64[n_poses, sum(n_atoms), 3]
 * n_mols <= 8. */
int tsc_transform_batch(tsc_ctx *ctx, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                        const int32_t *n_conf, int n_mols, const int32_t *conf_idx, const double *rot,
                        const double *pos, int64_t n_poses, double *out);
int tsc_transform_batch_dev(tsc_ctx *ctx, const double *frags, const int64_t *frag_off_host, const int32_t *n_atoms_host,
                            const int32_t *n_conf_host, int n_mols, const int32_t *conf_idx, const double *rot,
                            const double *pos, int64_t n_poses, double *out);

/* ---- K2: compenetration (clash) mask ---------------------------------------------------------
 * Replaces compenetration_check (tscode/numba_functions.py:59-105), count_clashes (:49-56) and the
 * all_dists it calls (tscode/algebra.py:98-157), batched as in compenetration_refining
 * (tscode/embedder.py:1243-1248):  mask[s] = compenetration_check(coords[s], ids, thresh, max_clashes).
 *   coords f64[n_poses, n_atoms, 3]; ids i32[n_ids] fragment lengths (contiguous ranges), n_ids in {0,2,3};
 *   n_ids == 0 is ids=None: count_clashes (ordered self pairs with 0 < d < 0.5; thresh is ignored).
 *   mask u8[n_poses] (1 = passes); counts (optional, may be NULL) i32[n_poses] = total pair count
 *   (for 3 fragments the count over all three fragment pairs: the reference's early exits do not
 *   change the verdict, total <= max_clashes). */
int tsc_clash_mask(tsc_ctx *ctx, const double *coords, int64_t n_poses, int n_atoms, const int32_t *ids, int n_ids,
                   double thresh, int64_t max_clashes, uint8_t *mask, int32_t *counts);
int tsc_clash_mask_dev(tsc_ctx *ctx, const double *coords, int64_t n_poses, int n_atoms, const int32_t *ids_host, int n_ids,
                       double thresh, int64_t max_clashes, uint8_t *mask, int32_t *counts);
/* Fused K1+K2: the clash verdict of every pose straight from its (rot, pos), no pose materialised
 * (the embed loops of tscode/embeds.py:116-118 and :713-714 do get_embed then compenetration_check). */
int tsc_embed_clash_mask_dev(tsc_ctx *ctx, const double *frags, const int64_t *frag_off_host, const int32_t *n_atoms_host,
                             const int32_t *n_conf_host, int n_mols, const int32_t *conf_idx, const double *rot,
                             const double *pos, int64_t n_poses, double thresh, int64_t max_clashes, uint8_t *mask,
                             int32_t *counts);
/* All-distances matrix of one pair of point sets (tscode/algebra.py:98-157), for value parity tests. */
int tsc_all_dists(tsc_ctx *ctx, const double *a, int na, const double *b, int nb, double *out);

/* ---- ordered compaction helpers (device) --------------------------------------------------------
 * n_kept = count_nonzero(mask); dst[rank(s)] = src[s] for mask[s] != 0, order preserved (NumPy's
 * structures[mask], tscode/rmsd_pruning.py:206, tscode/embedder.py:1250-1251).  row_bytes % 8 == 0.
 * tsc_gather_heavy_dev additionally keeps only the listed atoms: dst[rank(s), a] = src[s, heavy_idx[a]]
 * (structures[:, atomnos != 1], tscode/rmsd_pruning.py:178-179); mask may be NULL (keep all). */
int tsc_compact_rows_dev(tsc_ctx *ctx, const void *src, const uint8_t *mask, int64_t n_rows, int64_t row_bytes, void *dst,
                         int64_t *n_kept_host);
int tsc_gather_heavy_dev(tsc_ctx *ctx, const double *coords, const uint8_t *mask, int64_t n_poses, int n_atoms,
                         const int32_t *heavy_idx_host, int n_heavy, double *heavy_out, int64_t *n_kept_host);
/* The first half of tsc_pipeline_dev on its own (one rank's block of the pose axis in the sharded protocol): fused
 * embed + clash verdicts, ordered compaction, then the passing poses embedded straight into `structures` (all atoms,
 * may be NULL) and `heavy` (their heavy atoms, f64[n_pass, n_heavy, 3]) -- rejected poses are never materialised.
 * n_pass_host receives the count (the call synchronises for it while the embed runs).  With "early_basis" (default) the call also
 * estimates a descriptor basis from a sample of these poses on a side stream; the next tsc_prune_create on this context with
 * the same heavy-atom count uses it (once) instead of estimating its own -- a choice that never changes a verdict. */
int tsc_embed_clash_compact_dev(tsc_ctx *ctx, const double *frags, const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf,
                                int n_mols, const int32_t *conf_idx, const double *rot, const double *pos, int64_t n_poses,
                                const int32_t *heavy_idx, int n_heavy, double clash_thresh, int64_t max_clashes, uint8_t *clash_mask,
                                double *structures, double *heavy, int64_t *n_pass_host);

/* The two halves of the above as calls of their own, for a front half that is spread over ranks differently (tscode_amd/pipeline.py,
 * front = "hybrid": every rank takes the clash verdicts of ITS block of poses -- tsc_embed_clash_mask_dev --, the verdicts are
 * summed over the ranks, one byte per pose, and every rank then embeds the heavy atoms of ALL passing poses itself: recomputing a
 * pose from its 100 bytes of parameters costs less than moving its 24 n_heavy bytes of coordinates over xGMI).
 * tsc_basis_from_poses_dev: forks the estimate of the prune's descriptor basis from a sample of these poses onto the context's side
 *   stream (about 50 us that whatever is enqueued next on the main stream hides); consumed once, by tsc_embed_masked_dev or
 *   tsc_prune_create on this context.
 * tsc_embed_masked_dev: the poses selected by mask u8[n_poses] (device), embedded in order: structures f64[n_sel, n_atoms, 3]
 *   and / or heavy f64[n_sel, n_heavy, 3] (either may be NULL).  With `heavy` and a pending basis the kernel writes the prune's
 *   descriptors as well, and the next tsc_prune_create on this context over the same `heavy` takes them instead of reading the
 *   coordinates back.  n_sel_host (optional): count of selected poses (the call then synchronises for it while the embed runs).
 *   That run BORROWS the context's descriptor buffers until tsc_prune_destroy: a later tsc_embed_masked_dev on the same context that
 *   would have to regrow them while it lives fails with TSC_ERR_STATE instead of pulling them from under it. */
int tsc_basis_from_poses_dev(tsc_ctx *ctx, const double *frags, const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf, int n_mols,
                             const int32_t *conf_idx, const double *rot, const double *pos, int64_t n_poses, const int32_t *heavy_idx, int n_heavy);
int tsc_embed_masked_dev(tsc_ctx *ctx, const double *frags, const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf, int n_mols,
                         const int32_t *conf_idx, const double *rot, const double *pos, int64_t n_poses, const uint8_t *mask,
                         const int32_t *heavy_idx, int n_heavy, double *structures, double *heavy, int64_t *n_sel_host);

/* ---- K3: Kabsch RMSD (no centring) ------------------------------------------------------------
 * Replaces rmsd_and_max_numba (tscode/rmsd_pruning.py:6-41) on listed pairs of one heavy-atom array:
 * (rmsd[k], maxdev[k]) = rmsd_and_max_numba(heavy[pairs[k,0]], heavy[pairs[k,1]]).
 *   heavy f64[n_structs, h, 3]; pairs i32[n_pairs, 2]. */
int tsc_rmsd_pairs(tsc_ctx *ctx, const double *heavy, int64_t n_structs, int h, const int32_t *pairs, int64_t n_pairs,
                   double *rmsd, double *maxdev);
int tsc_rmsd_pairs_dev(tsc_ctx *ctx, const double *heavy, int64_t n_structs, int h, const int32_t *pairs, int64_t n_pairs,
                       double *rmsd, double *maxdev);

/* The descriptor screen of the prune's pair kernel as the matrix cores compute it (csrc/mm.hpp), for tests: the screen values
 * S[fam][r][c] (two feature families, rows r < 64, columns c < n) of the first 64 of n descriptors against all n, in the units the
 * kernel compares them in -- out_scale^2 times |D_fam[r] - D_fam[c]|^2 up to the error bound of mm.hpp -- and the limit the kernel
 * holds them against for a squared-distance limit `limit` = h thr^2 (as a float's bit pattern; INT32_MAX - 1: nothing is dropped).
 *   D f32[n, 16] host (component 2 k + fam, as the library stores descriptors); S f32[2, 64, n] host. */
int tsc_screen_mm_values(tsc_ctx *ctx, const float *D, int64_t n, double limit, float *S, int32_t *limit_bits, float *out_scale);

/* Torsion-fingerprint pruning (tscode/numba_functions.py:142-264).
 * tsc_torsion_fingerprints: _get_tf_mat -- out f32[n_structs, n_quads] of dihedral angles in degrees (tscode/algebra.py:24-55)
 * over quads i32[n_quads, 4]; coords f64[n_structs, n_atoms, 3].
 * tsc_tfd_first_similar: the pair search of one pass (:171-199) over fingerprints tf f32[n_structs, n_quads]: chunk `step`
 * is [d*step, d*(step+1)), the last one [d*(k-1), num_active); first i32[n_structs] = absolute index of the first j > i of
 * i's chunk with tfd_similarity(tf[i], tf[j], thresh) (:242-253), or -1.  The graph step that turns the matches into
 * rejects (:201-226) is the caller's (tscode_amd/numba_functions.py keeps the reference's networkx objects). */
int tsc_torsion_fingerprints(tsc_ctx *ctx, const double *coords, int64_t n_structs, int n_atoms, const int32_t *quads, int n_quads,
                             float *out);
int tsc_tfd_first_similar(tsc_ctx *ctx, const float *tf, int64_t n_structs, int n_quads, int64_t d, int64_t k, int64_t num_active,
                          double thresh, int32_t *first);

/* Moments of inertia and embed scores.
 * tsc_inertia_moments: tscode/algebra.py:165-186 get_inertia_moments for every structure -- out f64[n_structs, 3], the
 * eigenvalues of the inertia tensor about the centre of mass ordered by absolute value; masses f64[n_atoms].
 * tsc_moi_first_similar: the pair search of tscode/algebra.py:188-205 -- first i32[n_structs] = first j > i whose three moments
 * all differ by less than max_deviation relative to structure i's, or -1 (the graph step of prune_by_moment_of_inertia,
 * tscode/optimization_methods.py:341-358, is the caller's).
 * tsc_embed_scores: tscode/numba_functions.py:273-288 _score_embed_poses (scores f32[n_structs], float32 accumulation) and the
 * signed error of fitness_check (tscode/optimization_methods.py:544-557; fitness_error f64[n_structs]); indices i32[n_structs,
 * n_c, 2], distances f64[n_structs, n_c] (NaN = no target).  Host pointers. */
int tsc_inertia_moments(tsc_ctx *ctx, const double *structures, int64_t n_structs, int n_atoms, const double *masses, double *out);
int tsc_moi_first_similar(tsc_ctx *ctx, const double *moments, int64_t n_structs, double max_deviation, int32_t *first);
int tsc_embed_scores(tsc_ctx *ctx, const double *structures, int64_t n_structs, int n_atoms, const int32_t *indices,
                     const double *distances, int n_c, float *scores, double *fitness_error);

/* Pose parameters of the string embed (tscode/embeds.py:98-116), for n_sites (conformer pair, reactive-
 * centre pair) combinations x n_angles angles, pose = site * n_angles + angle index:
 *   R0 = rotation_matrix_from_vectors(mol_vec, -ref_vec) (:108, tscode/utils.py:183-208);
 *   R = rot_mat_from_pointer(ref_vec, angle) @ R0 when angle != 0 (:110-112);  t = p1 - R @ p2 (:114);
 * molecule 0 keeps identity / origin.  p1, p2, ref_vec, mol_vec f64[n_sites, 3]; conf_pair i32[n_sites, 2];
 * angles f64[n_angles] degrees; rot f64[N, 2, 9], pos f64[N, 2, 3], conf_idx i32[N, 2] -- the inputs of
 * tsc_transform_batch / tsc_embed_clash_mask_dev / tsc_pipeline_dev.  Host-pointer and device-pointer (_dev) forms. */
int tsc_string_embed_params(tsc_ctx *ctx, const double *p1, const double *p2, const double *ref_vec, const double *mol_vec,
                            const int32_t *conf_pair, int64_t n_sites, const double *angles, int n_angles, double *rot, double *pos,
                            int32_t *conf_idx);
int tsc_string_embed_params_dev(tsc_ctx *ctx, const double *p1, const double *p2, const double *ref_vec, const double *mol_vec,
                                const int32_t *conf_pair, int64_t n_sites, const double *angles, int n_angles, double *rot,
                                double *pos, int32_t *conf_idx);

/* Pose parameters of the cyclical embed (tscode/embeds.py:676-713), one row per (pose, molecule):
 * alignment = align_vec_pair([end - start, direction], [pivot, meanpoint - mean(reactive atoms)]) (tscode/algebra.py:258-282),
 * step = rot_mat_from_pointer(alignment @ (r0 - r1) or alignment @ pivot, angle), rotation = step @ alignment,
 * position = centre - step @ centre + mean(start, end) - alignment @ meanpoint with centre = alignment @ mean(reactive atoms).
 * start, end, direction, pivot, meanpoint, r0, r1 f64[n, 3] (r1 ignored where n_reactive is 1); n_reactive i32[n] (1 or 2);
 * angle f64[n] degrees; rot f64[n, 9], pos f64[n, 3].  Host pointers. */
int tsc_cyclical_embed_params(tsc_ctx *ctx, const double *start, const double *end, const double *direction, const double *pivot,
                              const double *meanpoint, const double *r0, const double *r1, const int32_t *n_reactive,
                              const double *angle, int64_t n, double *rot, double *pos);

/* Conformational-search rotations.  tsc_csearch_rotate builds every candidate of
 * tscode/torsion_module.py:463-500 from one start structure: for each torsion t with angles[m][t] != 0 the atoms of
 * masks[t] turn about the bond torsions[t][1]-torsions[t][2] (tscode/utils.py:389-414 rotate_dihedral, in place, in
 * torsion order), a rotation that fails tscode/numba_functions.py:26-47 torsion_comp_check is walked back in 5-degree
 * steps (angle // 5 of them, Python floor division) until it passes.  coords f64[n_atoms, 3]; torsions i32[n_tors, 4];
 * masks u8[n_tors, n_atoms] (tscode/torsion_module.py:301-325 _get_rotation_mask, computed by the caller);
 * angles i32[n_cand, n_tors] degrees; out f64[n_cand, n_atoms, 3]; rotated_bonds i32[n_cand] (the reference keeps a
 * candidate iff this is non-zero, :505).  tsc_torsion_comp_check: ok i32[n_structs] = 1 / 0 for structures sharing one
 * torsion and mask.  Host-pointer and device-pointer (_dev) forms. */
int tsc_csearch_rotate(tsc_ctx *ctx, const double *coords, int n_atoms, const int32_t *torsions, const uint8_t *masks, int n_tors,
                       const int32_t *angles, int64_t n_cand, double thresh, int64_t max_clashes, double *out, int32_t *rotated_bonds);
int tsc_csearch_rotate_dev(tsc_ctx *ctx, const double *coords, int n_atoms, const int32_t *torsions, const uint8_t *masks, int n_tors,
                           const int32_t *angles, int64_t n_cand, double thresh, int64_t max_clashes, double *out,
                           int32_t *rotated_bonds);
/* tsc_rotate_dihedral: rotate_dihedral (tscode/utils.py:389-414) for n_structs structures f64[n_structs, n_atoms, 3] that share the
 * torsion (i1, i2, i3, i4) and the mask u8[n_atoms] of the atoms that move: structure s turns them by angles[s] degrees -- any real
 * number, tscode/torsion_module.py:984-1005 searches fractional corrections -- about its own i2 - i3 bond (centre i3).  No clash
 * check, no walk-back.  out f64[n_structs, n_atoms, 3] must not alias coords.  Host arrays. */
int tsc_rotate_dihedral(tsc_ctx *ctx, const double *coords, int64_t n_structs, int n_atoms, const int32_t *torsion, const uint8_t *mask,
                        const double *angles, double *out);
int tsc_torsion_comp_check(tsc_ctx *ctx, const double *coords, int64_t n_structs, int n_atoms, const int32_t *torsion,
                           const uint8_t *mask, double thresh, int64_t max_clashes, int32_t *ok);

/* Greedy per-group filter of the embed loops (tscode/embeds.py:715, :843): inside each group a pose is accepted iff
 * it is not similar (tscode/rmsd_pruning.py:208-224, all atoms, rmsd < thr and maxdev < 2 thr) to any pose accepted
 * before it in that group.  poses f64[n_poses, n_atoms, 3]; group g is poses[group_off[g] : group_off[g+1]]
 * (group_off i32[n_groups + 1], group sizes <= 8192); accepted u8[n_poses]. */
int tsc_greedy_group_filter(tsc_ctx *ctx, const double *poses, const int32_t *group_off, int n_groups, int n_atoms,
                            double rmsd_thr, uint8_t *accepted);
int tsc_greedy_group_filter_dev(tsc_ctx *ctx, const double *poses, const int32_t *group_off_dev, int n_groups, int64_t n_poses,
                                int n_atoms, double rmsd_thr, uint8_t *accepted);

/* The embed loops as one call each.  Both take HOST pointers, keep every intermediate on the device
 * (pose parameters, candidate poses, fingerprints) and return, per candidate in the reference's loop order, the
 * compenetration_check verdict (clash_ok u8[N]) and whether the reference would have appended the pose (kept u8[N]);
 * the kept poses themselves come back compacted in candidate order (poses f64[n_kept, n_atoms_total, 3]; poses may be
 * NULL; poses_capacity = rows the buffer holds, an error if fewer than n_kept).
 *
 * tsc_tfd_greedy_filter: is_new_structure (tscode/embeds.py:47-69) over a whole ordered list of torsion fingerprints
 *   tf f32[n, n_quads]: accepted[s] = 1 iff no fingerprint accepted before s is tfd_similar (tscode/numba_functions.py:242-253,
 *   sum of wrapped differences < thresh) to s's.  The reference's list never evicts (`lru_cache = lru_cache[1:]` rebinds a
 *   local, :66-67) and neither does this.
 *
 * tsc_string_embed: the loop of tscode/embeds.py:91-120.  Two fragments (frags / frag_off / n_atoms / n_conf as in
 *   tsc_transform_batch); n_sites rows (conformer pair, reactive-centre pair) in the reference's order and n_angles angles,
 *   candidate = site * n_angles + angle index (tsc_string_embed_params); compenetration_check(ids = the two fragments,
 *   thresh = clash_thresh, max_clashes) (:118); is_new_structure over the passing poses with the torsion fingerprints of
 *   quads i32[n_quads, 4] (atom indices in the embedded structure) and tfd_thresh (the reference uses 10) (:119).
 *
 * tsc_cyclical_embed: the inner loops of tscode/embeds.py:657-717 and :785-847 for any number of (conformers, pivots,
 *   polygon orientation) groups at once.  One row per (pose, molecule), row = pose * n_mols + m, with the inputs of
 *   tsc_cyclical_embed_params plus conf_idx i32[rows]; group_off i32[n_groups + 1] cuts the poses into the reference's
 *   `angular_poses` groups (consecutive, sizes <= 8192); compenetration_check (:714) and then, inside each group and in
 *   order, `not _rmsd_similarity(pose, kept poses of the group, rmsd_thr)` (:715; the reference passes 1). */
int tsc_tfd_greedy_filter(tsc_ctx *ctx, const float *tf, int64_t n_structs, int n_quads, double thresh, uint8_t *accepted,
                          int64_t *n_kept);
int tsc_string_embed(tsc_ctx *ctx, const double *frags, const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf,
                     const double *p1, const double *p2, const double *ref_vec, const double *mol_vec, const int32_t *conf_pair,
                     int64_t n_sites, const double *angles, int n_angles, double clash_thresh, int64_t max_clashes,
                     const int32_t *quads, int n_quads, double tfd_thresh, uint8_t *clash_ok, uint8_t *kept, double *poses,
                     int64_t poses_capacity, int64_t *n_pass, int64_t *n_kept);
int tsc_cyclical_embed(tsc_ctx *ctx, const double *frags, const int64_t *frag_off, const int32_t *n_atoms, const int32_t *n_conf,
                       int n_mols, const double *start, const double *end, const double *direction, const double *pivot,
                       const double *meanpoint, const double *r0, const double *r1, const int32_t *n_reactive, const double *angle,
                       const int32_t *conf_idx, int64_t n_poses, const int32_t *group_off, int n_groups, double clash_thresh,
                       int64_t max_clashes, double rmsd_thr, uint8_t *clash_ok, uint8_t *kept, double *poses, int64_t poses_capacity,
                       int64_t *n_pass, int64_t *n_kept);

/* HOST-side helper (no device, no context): the graph step of tscode/numba_functions.py:201-226 (prune_conformers_tfd) and
 * tscode/optimization_methods.py:341-358 (prune_by_moment_of_inertia) for any number of chunks: the matches (rel_i[q], rel_j[q]),
 * q in [chunk_ptr[c], chunk_ptr[c+1]), of chunk c are node indices relative to the chunk (0 <= index < chunk_len[c]), listed in the
 * order the reference adds them to its set (rows ascending); of every connected component of a chunk's match graph only
 * `tuple(subgraph.nodes)[0]` is kept: keep[chunk_off[c] + node] is cleared for the others (keep u8[n_total], set by the caller).
 * Which node that expression names is decided by CPython's set / dict iteration orders inside networkx 3.x; this call re-plays
 * them (tscode_amd/csrc/host_order.hpp).  The Python caller checks the emulation against the real objects before relying on it. */
int tsc_host_graph_step(const int64_t *rel_i, const int64_t *rel_j, const int64_t *chunk_ptr, const int64_t *chunk_off,
                        const int64_t *chunk_len, int64_t n_chunks, int64_t n_total, uint8_t *keep);

/* Per-pass statistics of a prune run (one entry per executed k of the schedule). */
typedef struct {
    int64_t k;               /* number of chunks (tscode/rmsd_pruning.py:186-188) */
    int64_t n_active_before; /* count_nonzero(mask) entering the pass */
    int64_t n_active_after;
    int64_t pairs_evaluated; /* pair evaluations the reference's sequential scan performs in this pass (:70) */
    int64_t pairs_computed;  /* pairs for which the GPU formed H = p^T q and ran the sign test */
    int64_t candidates;      /* pairs that reached the explicit-rotation path */
    int64_t pairs_screened;  /* pairs looked at by the descriptor sieve (0 when the register-tiled kernel ran) */
    int64_t new_keys;        /* cache keys appended (:76, :204) */
    double gpu_ms;           /* HIP-event time of the whole pass on this device (0 unless "pass_timing" is 2) */
    double tile_ms;          /* HIP-event time of the pass's pair kernel alone (0 unless "pass_timing" >= 1; chunk-local passes: 2) */
    int32_t algo;            /* kernel that ran the pass: 1 = register-tiled all-pairs, 2 = descriptor sieve, 3 = chunk-local kernel */
    int32_t nonfinite_input; /* 1: the run met a structure with a NaN or infinite coordinate (the same in every entry of a run; descriptor-sieve
                                runs only).  Such a structure is similar to nothing -- every comparison of :75 with a NaN is false -- and is
                                kept; the reference itself raises LinAlgError there (np.linalg.svd, :19), and so does the Python drop-in */
} tsc_pass_stats;

#define TSC_MAX_PASSES 18

/* prune_conformers_rmsd (tscode/rmsd_pruning.py:164-206) on the heavy-atom array:
 *   heavy f64[n, h, 3] = structures[:, atomnos != 1]; rmsd_thr as in the reference (max deviation
 *   threshold is 2*rmsd_thr, :95);
 *   mode 0 = reference-exact, including the pair-cache behaviour of :65-67 / :75-77;
 *   mode 1 = cache-free (the cache test is skipped);
 *   mask u8[n] out (1 = kept); stats (optional) up to TSC_MAX_PASSES entries, n_passes out (optional).
 * For n > 200 000 the reference itself fails (a float k reaches range()); the schedule is used with int(k). */
int tsc_prune_rmsd(tsc_ctx *ctx, const double *heavy, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask,
                   tsc_pass_stats *stats, int *n_passes);
/* The same from the arrays the reference's prune_conformers_rmsd receives (rmsd_pruning.py:164-206): structures f64[n, n_atoms, 3]
 * with ALL atoms in host memory and heavy_idx i32[n_heavy] (= flatnonzero(atomnos != 1), increasing): the gather of :178-179 runs on
 * the device. */
int tsc_prune_structures(tsc_ctx *ctx, const double *structures, int64_t n, int n_atoms, const int32_t *heavy_idx, int n_heavy,
                         double rmsd_thr, int mode, uint8_t *mask, tsc_pass_stats *stats, int *n_passes);
int tsc_prune_rmsd_dev(tsc_ctx *ctx, const double *heavy, int64_t n, int h, double rmsd_thr, int mode, uint8_t *mask,
                       tsc_pass_stats *stats, int *n_passes); /* synchronises (the schedule gate reads counts) */

/* Stepping form of the same run, for one-process-per-GPU sharding of a pass (rows of a pass are
 * independent: tscode/rmsd_pruning.py:92,101-113).  Every rank holds the full `heavy` array and calls
 *   tsc_prune_create; loop { k = tsc_prune_next_pass; if k == 0 break;
 *                            tsc_prune_pass_local(rank, world);      // this rank's row tiles -> best[]
 *                            <all-reduce MIN over tsc_prune_best_ptr, n_active int32 entries>   (RCCL)
 *                            tsc_prune_pass_finish; }                 // identical mask/cache update on every rank
 *   tsc_prune_mask_dev gives the device mask; tsc_prune_destroy frees the state. */
/* A context serves at most 64 live runs at a time (each owns a word of the context's pinned memory; the 65th tsc_prune_create fails with
 * TSC_ERR_STATE).  tsc_prune_create / tsc_prune_destroy take the context's scratch cache: calls of them on ONE context must not overlap
 * (a context is for one thread at a time, see above); runs that exist may then be stepped from different threads.  Runs still alive
 * when their context is destroyed are destroyed with it -- do not hand them to tsc_prune_destroy afterwards. */
int tsc_prune_create(tsc_ctx *ctx, const double *heavy_dev, int64_t n, int h, double rmsd_thr, int mode, tsc_prune **out);
int tsc_prune_next_pass(tsc_prune *p, int64_t *k_out);            /* 0 when the schedule is exhausted; does not wait:
                                                                      the gate of rmsd_pruning.py:192 is evaluated on the
                                                                      device, a pass whose gate is closed does nothing */
int tsc_prune_pass_estimate(tsc_prune *p, int64_t *pairs);        /* upper bound of the pairs of the open pass: lets every
                                                                      rank decide alike whether sharding it pays */
/* Runs every pass that needs no exchange (world == 1: all; world > 1: those with an estimate below min_pairs, computed whole by
 * every rank) and returns with the first pass that does left open (*k = its k) or *k = 0 at the end of the schedule
 * (rmsd_pruning.py:186-204).  One host call instead of three per small pass. */
int tsc_prune_run_replicated(tsc_prune *run, int world, int64_t min_pairs, int64_t *k);
int tsc_prune_pass_local(tsc_prune *p, int rank, int world_size); /* asynchronous -- except on a pass that MAY be culled (option "cull":
                                                                      at least "cull_min_pairs" pairs, fewer than 64 chunks), where the
                                                                      host waits once for the device's culled-or-walked verdict (the
                                                                      run's own word of pinned memory: runs of one context driven from
                                                                      different host threads do not share it); tsc_prune_pass_range
                                                                      likewise.  With world_size == 1 the verdicts are applied
                                                                      in here as well -- by the pair kernel itself, tile by tile
                                                                      (option "fused_apply"; best[] stays readable), or, for a pass
                                                                      whose chunks are short, by the chunk-local kernel (option
                                                                      "local_pass"; best[] is then not produced) -- and
                                                                      tsc_prune_pass_finish only does the host's bookkeeping */
int tsc_prune_pass_rows(tsc_prune *p, int rank, int world_size);  /* after tsc_prune_pass_local: the pair search of ANOTHER rank's row
                                                                     tiles of the same pass, into the same best[] (atomicMin).  Lets
                                                                     one GPU stand in for several ranks (tools/predict_scaling.py times
                                                                     every rank's share this way; repeating a share changes nothing) */
/* RANK-PARTITIONED passes ("early passes": whole chunks to GPUs).  The chunks of a pass are independent
 * (tscode/rmsd_pruning.py:139-157: every chunk reads the same input mask and the same cache), so while a pass has at least
 * min_chunks_per_rank chunks per rank each rank runs the WHOLE pass flow -- rows, stop columns, pair search, verdicts -- on the
 * chunks that start inside its block [n rank / world, n (rank + 1) / world) of the structure axis and on nothing else: no
 * per-row work is replicated, and what crosses the ranks is one bit per structure plus five counters.
 *   tsc_prune_exchange_words(n, mode) -> words of the exchange buffer (int64: n / 64 + 40 removed-row bits, 8 statistics, then --
 *                                mode 0 -- the storage of the run's cache views, which move into this buffer)
 *   tsc_prune_set_partition      right after tsc_prune_create; exch_dev = caller-owned device buffer of that many int64 (e.g. a
 *                                torch tensor that torch.distributed can all-reduce); zeroed here.  The per-pass exchange is over
 *                                its first n / 64 + 48 words
 *   per pass:  tsc_prune_next_pass;  tsc_prune_pass_partitioned -> 1:
 *                  tsc_prune_pass_range;                       // this rank's chunks, asynchronous
 *                  <all-reduce SUM over the exchange buffer>   // the ranks' bits are disjoint: the sum is their union
 *                  tsc_prune_pass_merge;                       // mask, bit copy, scan counts, record, the gate of :192, the next
 *                                                              // pass's rows -- identical on every rank; asynchronous
 *              -> 0: the first such pass after partitioned ones needs the cache keys of every rank (a key (a, b) is only ever
 *                  hit in the chunk that starts at a, which belongs to the same rank in every partitioned pass -- so the keys
 *                  stayed where they were made):  tsc_prune_views_ptr -> words > 0:  <all-reduce SUM over that block> (views_dev,
 *                  or exch_dev + offset_words: the views of the passes still to run), tsc_prune_views_merged;  then
 *                  tsc_prune_pass_local / tsc_prune_pass_finish as above.
 * The per-pass statistics of a partitioned pass (tsc_prune_stats) are the sums over all ranks. */
int tsc_prune_exchange_words(int64_t n, int mode, int64_t *words);
int tsc_prune_set_partition(tsc_prune *p, int rank, int world_size, int min_chunks_per_rank, void *exch_dev_i64, int64_t exch_words);
int tsc_prune_pass_partitioned(tsc_prune *p, int *flag);
int tsc_prune_pass_range(tsc_prune *p);
int tsc_prune_pass_merge(tsc_prune *p);
int tsc_prune_views_ptr(tsc_prune *p, void **views_dev_i64, int64_t *offset_words, int64_t *words);
int tsc_prune_views_merged(tsc_prune *p);
int tsc_prune_best_ptr(tsc_prune *p, void **best_dev, int64_t *n_entries); /* i32[n_entries], valid until finish */
/* Make the run keep best[] in a caller-owned device buffer of n int32 (e.g. a torch tensor that
 * torch.distributed can all-reduce); call right after tsc_prune_create. */
int tsc_prune_use_best_buffer(tsc_prune *p, void *best_dev_i32_n);
int tsc_prune_pass_finish(tsc_prune *p);                          /* asynchronous */

/* ---- the pass loop of a sharded run behind ONE call (the multi-rank variant of the prune) ----
 * tsc_prune_run_sharded walks the whole schedule of `run` as the step-by-step calls above would -- passes below `min_pairs` pairs whole
 * on every rank; passes with at least `min_chunks_per_rank` chunks per rank partitioned by chunks (0: never; needs the exchange buffer
 * exch_dev of tsc_prune_exchange_words words, which then also holds the cache views); the cache views summed once before the first pass
 * of the other kind; the remaining passes dealt by row tiles -- and calls back only for the collectives:
 *     exchange(user, kind, buf_dev, count)   reduce `count` elements at the DEVICE address buf_dev over all ranks, in place:
 *                                            TSC_XCHG_SUM_I64 = all-reduce SUM of int64 (removed-row bits and statistics of a partitioned
 *                                            pass; the cache views), TSC_XCHG_MIN_I32 = all-reduce MIN of int32 (best[] of a pass dealt
 *                                            by row tiles).  Everything the library enqueued before the call is on the context's stream:
 *                                            enqueue the collective on that stream (RCCL: ncclAllReduce(buf, buf, count, ncclInt64 /
 *                                            ncclInt32, ncclSum / ncclMin, comm, stream)) or synchronise around it.  Return 0; anything
 *                                            else aborts the run with TSC_ERR_STATE.
 * One process per GPU owns the communicator; the library opens none.  Every rank makes the same sequence of calls (it depends on n, k
 * and the world size only).  log (optional, log_cap entries): the exchanges made, in order -- k < 0 marks the one exchange of the cache
 * views in front of pass |k|.  With world == 1 the function is tsc_prune_run_replicated to the end (exchange may be NULL).
 * Afterwards: tsc_prune_copy_mask_dev / tsc_prune_stats / tsc_prune_destroy as usual.
 * tscode/rmsd_pruning.py:139-157 (chunks independent), :92,101-113 (rows independent). */
enum { TSC_XCHG_SUM_I64 = 1, TSC_XCHG_MIN_I32 = 2 };
typedef int (*tsc_exchange_fn)(void *user, int kind, void *buf_dev, int64_t count);
typedef struct tsc_exchange_record {
    int64_t k;      /* the pass (negative: the cache views in front of pass -k) */
    int32_t kind;   /* TSC_XCHG_* */
    int64_t count;  /* elements reduced */
} tsc_exchange_record;
int tsc_prune_run_sharded(tsc_prune *run, int rank, int world_size, int min_chunks_per_rank, int64_t min_pairs, void *exch_dev, int64_t exch_words,
                          tsc_exchange_fn exchange, void *user, tsc_exchange_record *log, int log_cap, int *n_log);
/* ---- the exchange inside the library (optional): a one-shot all-reduce over memory the ranks map into each other ----
 * For the small per-pass messages above a collective library's fixed cost (and, from a scripting host, a frame of the host language per
 * collective) is most of what an exchange costs.  A tsc_xchg gives every rank a receive area of FINE-GRAINED device memory that the other
 * ranks of the node map (hipIpcGetMemHandle / hipIpcOpenMemHandle): an exchange is then one kernel that writes this rank's contribution
 * into every peer and raises a flag there, and one that waits for the peers' flags and folds what they delivered into the caller's buffer
 * -- enqueued on the context's stream, no host in between.  The library still opens no communicator: the HOST carries the 64-byte
 * handles from rank to rank, once, over whatever it has (in this repository torch.distributed.all_gather_object).
 *   tsc_xchg_slot_bytes   bytes a slot must hold for runs over up to n structures (the largest message of tsc_prune_run_sharded)
 *   tsc_xchg_create       allocates the area (header + 2 x world slots of slot_bytes) and returns its IPC handle (TSC_XCHG_HANDLE_BYTES bytes)
 *   tsc_xchg_connect      handles = world x TSC_XCHG_HANDLE_BYTES bytes, in rank order (the own entry is ignored): maps the peers.  Call it on
 *                         every rank after all have created; ranks may share a device (other PROCESSES; tests do)
 *   tsc_xchg_allreduce    has the signature of tsc_exchange_fn with user = the tsc_xchg: pass it to tsc_prune_run_sharded as the exchange
 *                         function, or call it directly (buf 8-byte aligned; count elements of int64 / int32 as `kind` says)
 *   tsc_xchg_status       exchanges made so far and how many of them gave up waiting for a peer (tsc_xchg_set_timeout, default 5 s: a
 *                         rank that died must not hang the others' GPUs); after a timeout the reduced buffers are NOT valid -- check after
 *                         the run's synchronisation.  Every rank must make the same sequence of exchanges (tsc_prune_run_sharded does).
 * What these messages are: tscode/rmsd_pruning.py:149-157 (disjoint out_mask[first:last] per chunk), :92,101-113 (rows independent). */
typedef struct tsc_xchg tsc_xchg;
#define TSC_XCHG_HANDLE_BYTES 64
int tsc_xchg_slot_bytes(int64_t n, int mode, int64_t *bytes);
int tsc_xchg_create(tsc_ctx *ctx, int rank, int world_size, int64_t slot_bytes, tsc_xchg **out, void *handle_out);
int tsc_xchg_connect(tsc_xchg *x, const void *handles);
int tsc_xchg_set_timeout(tsc_xchg *x, double seconds);
int tsc_xchg_allreduce(void *xchg, int kind, void *buf_dev, int64_t count);
int tsc_xchg_status(tsc_xchg *x, int64_t *n_exchanges, int *n_timeouts);
int tsc_xchg_destroy(tsc_xchg *x);
int tsc_prune_mask_dev(tsc_prune *p, const uint8_t **mask_dev);
int tsc_prune_copy_mask_dev(tsc_prune *p, uint8_t *dst_dev); /* dst[0..n) <- mask, asynchronous on the stream */
int tsc_prune_stats(tsc_prune *p, tsc_pass_stats *stats, int *n_passes); /* synchronises */
int tsc_prune_destroy(tsc_prune *p);

/* ---- whole pipeline on one device ----------------------------------------------------------------
 * generate -> clash-filter -> similarity-prune, the sequence RunEmbedding.run drives through
 * generate_candidates / compenetration_refining / similarity_refining (tscode/embedder.py:1136-1154,
 * 1230-1266, 1356-1368), with everything resident in HBM:
 *   K1+K2 fused verdicts, ordered compaction of the passing poses (all atoms + heavy atoms), K3 prune.
 * Inputs as tsc_transform_batch_dev plus heavy_idx (indices of atoms with atomnos != 1).
 * Outputs (device, caller-allocated for the worst case n_poses):
 *   clash_mask u8[n_poses]; structures f64[n_pass, n_atoms, 3] (poses that pass the clash check, in order);
 *   keep_mask u8[n_pass] (prune verdict on those); keep_mask_host (optional, host, n_poses bytes) receives a copy of
 *   keep_mask[0 .. n_pass) before the call returns; n_pass_host / n_keep_host scalars on the host.
 * timings_ms (optional, host) float[4] = {embed+clash, compaction, prune, total} from HIP events.
 * HOST COST: the call returns after ONE synchronisation at its end, but in the middle it needs the number of poses that passed the clash
 * check to size the prune (the schedule depends on it).  The scan kernel writes that count into the context's pinned memory and the calling
 * thread SPINS on it (a pause instruction per look: x86; about 40 us per call at 100k poses -- the clash kernel + the scan -- with a fall-back
 * to a copy + synchronise after 5 s): a host core per context is busy for that long in every call.  A caller that keeps several contexts in
 * flight from one thread each pays it per context. */
int tsc_pipeline_dev(tsc_ctx *ctx, const double *frags, const int64_t *frag_off_host, const int32_t *n_atoms_host,
                     const int32_t *n_conf_host, int n_mols, const int32_t *conf_idx, const double *rot, const double *pos,
                     int64_t n_poses, const int32_t *heavy_idx_host, int n_heavy, double clash_thresh, int64_t max_clashes,
                     double rmsd_thr, int mode, uint8_t *clash_mask, double *structures, uint8_t *keep_mask,
                     uint8_t *keep_mask_host, int64_t *n_pass_host, int64_t *n_keep_host, tsc_pass_stats *stats, int *n_passes,
                     float *timings_ms);

#ifdef __cplusplus
}
#endif
#endif /* TSCODE_HIP_H */
