// adjacent.hip -- the adjacent rows: greedy filters, csearch rotations, TFD, moments of inertia, the embed drivers
// gfx950 only.  There is deliberately no CPU implementation behind these entry points.
#include "host.hpp"
#include "scan.hpp"
#include "group_filter.hpp"
#include "csearch.hpp"
#include "tfd.hpp"
#include "moi.hpp"
#include "host_order.hpp"

// --------------------------------------------------------------------------------------------------
// greedy per-group filter

extern "C" __attribute__((visibility("default"))) int tsc_greedy_group_filter_dev(tsc_ctx *c, const double *poses, const int32_t *group_off_dev, int n_groups,
                                                                             int64_t n_poses, int n_atoms, double rmsd_thr, uint8_t *accepted) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && poses && group_off_dev && accepted, "tsc_greedy_group_filter_dev: null argument");
    TSC_REQUIRE(n_groups >= 0 && n_poses >= 0 && n_atoms > 0 && rmsd_thr > 0, "bad sizes");
    if (n_groups == 0 || n_poses == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *G;
    TSC_TRY(s.get(size_t(n_poses), &G));
    hipLaunchKernelGGL(k_greedy_group_filter, dim3(n_groups), dim3(256), 0, c->stream, poses, group_off_dev, n_groups, n_atoms, rmsd_thr,
                       accepted, G);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_greedy_group_filter(tsc_ctx *c, const double *poses, const int32_t *group_off, int n_groups, int n_atoms,
                                                                         double rmsd_thr, uint8_t *accepted) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && poses && group_off && accepted, "tsc_greedy_group_filter: null argument");
    TSC_REQUIRE(n_groups >= 0 && n_atoms > 0 && rmsd_thr > 0, "bad sizes");
    if (n_groups == 0) return 0;
    TSC_REQUIRE(group_off[0] == 0, "group_off must start at 0");
    for (int g = 0; g < n_groups; ++g)
        TSC_REQUIRE(group_off[g + 1] >= group_off[g] && group_off[g + 1] - group_off[g] <= GF_MAX_GROUP,
                    "group %d: sizes must be in [0, %d]", g, GF_MAX_GROUP);
    const int64_t n_poses = group_off[n_groups];
    if (n_poses == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_poses;
    int32_t *d_off;
    uint8_t *d_acc;
    TSC_TRY(upload(c, s, poses, size_t(n_poses) * n_atoms * 3, &d_poses));
    TSC_TRY(upload(c, s, group_off, size_t(n_groups) + 1, &d_off));
    TSC_TRY(s.get(size_t(n_poses), &d_acc));
    TSC_TRY(tsc_greedy_group_filter_dev(c, d_poses, d_off, n_groups, n_poses, n_atoms, rmsd_thr, d_acc));
    TSC_HIP(hipMemcpyAsync(accepted, d_acc, size_t(n_poses), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// conformational-search rotations

static int csearch_args(int n_atoms, int n_tors, int64_t n_cand, double thresh, int64_t max_clashes, CsearchArgs *a) {
    TSC_REQUIRE(n_atoms > 0 && n_tors >= 0 && n_cand >= 0, "bad sizes (%d atoms, %d torsions, %lld candidates)", n_atoms, n_tors, (long long)n_cand);
    TSC_REQUIRE(n_atoms <= 65535 && torsion_lists_bytes(n_tors, n_atoms) + csearch_wave_bytes(n_atoms) <= 150 * 1024,
                "%d atoms x %d torsions exceed the LDS staging of the csearch kernels", n_atoms, n_tors);
    a->n = n_atoms, a->n_tors = n_tors, a->n_cand = n_cand;
    a->sq_bound = clash_sq_bound(thresh), a->max_clashes = max_clashes;
    return 0;
}

// wavefronts per workgroup (4, 2 or 1) that fit the LDS, and the dynamic LDS size of the launch
static int csearch_waves(int n_atoms, int n_tors, size_t *lds) {
    int w = 4;
    while (w > 1 && torsion_lists_bytes(n_tors, n_atoms) + w * csearch_wave_bytes(n_atoms) > 150 * 1024) w >>= 1;
    *lds = torsion_lists_bytes(n_tors, n_atoms) + w * csearch_wave_bytes(n_atoms);
    return w;
}

static int check_torsions(const int32_t *torsions, int n_tors, int n_atoms) {
    for (int t = 0; t < n_tors; ++t)
        for (int q = 0; q < 4; ++q)
            TSC_REQUIRE(torsions[4 * t + q] >= 0 && torsions[4 * t + q] < n_atoms, "torsion %d: atom index %d out of range", t, torsions[4 * t + q]);
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_csearch_rotate_dev(tsc_ctx *c, const double *coords, int n_atoms, const int32_t *torsions,
                                                                             const uint8_t *masks, int n_tors, const int32_t *angles, int64_t n_cand,
                                                                             double thresh, int64_t max_clashes, double *out, int32_t *rotated_bonds) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && out && rotated_bonds && torsions && masks && angles, "tsc_csearch_rotate_dev: null argument");
    CsearchArgs a;
    TSC_TRY(csearch_args(n_atoms, n_tors, n_cand, thresh, max_clashes, &a));
    if (n_cand == 0) return 0;
    DeviceGuard guard(c->device);
    size_t lds;
    const int waves = csearch_waves(n_atoms, n_tors, &lds);
    if (lds > 64 * 1024)
        TSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_csearch_rotate), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k_csearch_rotate, dim3(grid_for(n_cand, waves, 256 * 8)), dim3(64 * waves), lds, c->stream, a, coords, torsions, masks, angles, out,
                       rotated_bonds);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_csearch_rotate(tsc_ctx *c, const double *coords, int n_atoms, const int32_t *torsions,
                                                                         const uint8_t *masks, int n_tors, const int32_t *angles, int64_t n_cand,
                                                                         double thresh, int64_t max_clashes, double *out, int32_t *rotated_bonds) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && out && rotated_bonds && (n_tors == 0 || (torsions && masks && angles)), "tsc_csearch_rotate: null argument");
    TSC_REQUIRE(n_atoms > 0 && n_tors >= 0 && n_cand >= 0, "bad sizes");
    TSC_TRY(check_torsions(torsions, n_tors, n_atoms));
    if (n_cand == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_coords, *d_out;
    int32_t *d_tors, *d_angles, *d_rb;
    uint8_t *d_masks;
    TSC_TRY(upload(c, s, coords, size_t(n_atoms) * 3, &d_coords));
    TSC_TRY(upload(c, s, torsions, size_t(n_tors) * 4, &d_tors));
    TSC_TRY(upload(c, s, masks, size_t(n_tors) * n_atoms, &d_masks));
    TSC_TRY(upload(c, s, angles, size_t(n_cand) * n_tors, &d_angles));
    TSC_TRY(s.get(size_t(n_cand) * n_atoms * 3, &d_out));
    TSC_TRY(s.get(size_t(n_cand), &d_rb));
    TSC_TRY(tsc_csearch_rotate_dev(c, d_coords, n_atoms, d_tors, d_masks, n_tors, d_angles, n_cand, thresh, max_clashes, d_out, d_rb));
    TSC_HIP(hipMemcpyAsync(out, d_out, size_t(n_cand) * n_atoms * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(rotated_bonds, d_rb, size_t(n_cand) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_rotate_dihedral(tsc_ctx *c, const double *coords, int64_t n_structs, int n_atoms, const int32_t *torsion,
                                                                          const uint8_t *mask, const double *angles, double *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && torsion && mask && angles && out, "tsc_rotate_dihedral: null argument");
    TSC_REQUIRE(n_structs >= 0 && n_atoms > 0, "bad sizes");
    TSC_REQUIRE(out != coords, "tsc_rotate_dihedral: out must not alias coords");
    for (int q = 1; q <= 2; ++q) TSC_REQUIRE(torsion[q] >= 0 && torsion[q] < n_atoms, "torsion index %d out of range", torsion[q]);
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_c, *d_o, *d_a;
    uint8_t *d_m;
    TSC_TRY(upload(c, s, coords, size_t(n_structs) * n_atoms * 3, &d_c));
    TSC_TRY(upload(c, s, angles, size_t(n_structs), &d_a));
    TSC_TRY(upload(c, s, mask, size_t(n_atoms), &d_m));
    TSC_TRY(s.get(size_t(n_structs) * n_atoms * 3, &d_o));
    hipLaunchKernelGGL(k_rotate_dihedral, dim3(grid_for(n_structs * n_atoms, 256)), dim3(256), 0, c->stream, (const double *)d_c, n_structs, n_atoms, int(torsion[1]),
                       int(torsion[2]), (const uint8_t *)d_m, (const double *)d_a, d_o);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(out, d_o, size_t(n_structs) * n_atoms * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_torsion_comp_check(tsc_ctx *c, const double *coords, int64_t n_structs, int n_atoms,
                                                                             const int32_t *torsion, const uint8_t *mask, double thresh,
                                                                             int64_t max_clashes, int32_t *ok) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && torsion && mask && ok, "tsc_torsion_comp_check: null argument");
    CsearchArgs a;
    TSC_TRY(csearch_args(n_atoms, 1, n_structs, thresh, max_clashes, &a));
    TSC_TRY(check_torsions(torsion, 1, n_atoms));
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_coords;
    int32_t *d_tors, *d_ok;
    uint8_t *d_mask;
    TSC_TRY(upload(c, s, coords, size_t(n_structs) * n_atoms * 3, &d_coords));
    TSC_TRY(upload(c, s, torsion, size_t(4), &d_tors));
    TSC_TRY(upload(c, s, mask, size_t(n_atoms), &d_mask));
    TSC_TRY(s.get(size_t(n_structs), &d_ok));
    size_t lds;
    const int waves = csearch_waves(n_atoms, 1, &lds);
    if (lds > 64 * 1024)
        TSC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_torsion_comp_check), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k_torsion_comp_check, dim3(grid_for(n_structs, waves, 256 * 8)), dim3(64 * waves), lds, c->stream, a, (const double *)d_coords,
                       (const int32_t *)d_tors, (const uint8_t *)d_mask, d_ok);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(ok, d_ok, size_t(n_structs) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// torsion-fingerprint pruning

extern "C" __attribute__((visibility("default"))) int tsc_torsion_fingerprints(tsc_ctx *c, const double *coords, int64_t n_structs, int n_atoms,
                                                                               const int32_t *quads, int n_quads, float *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && coords && quads && out, "tsc_torsion_fingerprints: null argument");
    TSC_REQUIRE(n_structs >= 0 && n_atoms > 0 && n_quads >= 0, "bad sizes");
    for (int q = 0; q < 4 * n_quads; ++q) TSC_REQUIRE(quads[q] >= 0 && quads[q] < n_atoms, "quadruplet atom index %d out of range", quads[q]);
    if (n_structs == 0 || n_quads == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_coords;
    int32_t *d_quads;
    float *d_out;
    TSC_TRY(upload(c, s, coords, size_t(n_structs) * n_atoms * 3, &d_coords));
    TSC_TRY(upload(c, s, quads, size_t(n_quads) * 4, &d_quads));
    TSC_TRY(s.get(size_t(n_structs) * n_quads, &d_out));
    hipLaunchKernelGGL(k_torsion_fingerprints, dim3(grid_for(n_structs * n_quads, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)d_coords,
                       n_structs, n_atoms, (const int32_t *)d_quads, n_quads, d_out);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(out, d_out, size_t(n_structs) * n_quads * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_tfd_first_similar(tsc_ctx *c, const float *tf, int64_t n_structs, int n_quads, int64_t d,
                                                                            int64_t k, int64_t num_active, double thresh, int32_t *first) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && tf && first, "tsc_tfd_first_similar: null argument");
    if (n_structs == 0) return 0;
    TSC_REQUIRE(n_structs >= 0 && n_quads >= 0 && d > 0 && k > 0 && num_active >= 0 && num_active <= n_structs && d * k <= n_structs,
                "bad pass geometry (n = %lld, d = %lld, k = %lld, active = %lld)", (long long)n_structs, (long long)d, (long long)k, (long long)num_active);
    TSC_REQUIRE(n_structs < INT32_MAX, "too many structures");
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    float *d_tf;
    int32_t *d_first;
    TSC_TRY(upload(c, s, tf, size_t(n_structs) * n_quads, &d_tf));
    TSC_TRY(s.get(size_t(n_structs), &d_first));
    hipLaunchKernelGGL(k_tfd_first_similar, dim3(grid_for(n_structs, 4, 256 * 16)), dim3(256), 0, c->stream, (const float *)d_tf, n_structs, n_quads, d, k,
                       num_active, thresh, d_first);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(first, d_first, size_t(n_structs) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// the graph step of the similarity prunings on the host (host_order.hpp): no device involved
extern "C" __attribute__((visibility("default"))) int tsc_host_graph_step(const int64_t *rel_i, const int64_t *rel_j, const int64_t *chunk_ptr,
                                                                          const int64_t *chunk_off, const int64_t *chunk_len, int64_t n_chunks,
                                                                          int64_t n_total, uint8_t *keep) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(rel_i && rel_j && chunk_ptr && chunk_off && chunk_len && keep && n_chunks >= 0 && n_total >= 0, "tsc_host_graph_step: bad argument");
    int64_t longest = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        TSC_REQUIRE(chunk_ptr[c + 1] >= chunk_ptr[c] && chunk_off[c] >= 0 && chunk_len[c] >= 0 && chunk_off[c] + chunk_len[c] <= n_total,
                    "chunk %lld: bad bounds", (long long)c);
        for (int64_t q = chunk_ptr[c]; q < chunk_ptr[c + 1]; ++q)
            TSC_REQUIRE(rel_i[q] >= 0 && rel_i[q] < chunk_len[c] && rel_j[q] >= 0 && rel_j[q] < chunk_len[c] && rel_i[q] != rel_j[q],
                        "match %lld lies outside its chunk", (long long)q);
        longest = std::max(longest, chunk_len[c]);
    }
    std::vector<int32_t> index_of(size_t(longest), -1);
    for (int64_t c = 0; c < n_chunks; ++c)
        tsc_host::graph_step_chunk(rel_i + chunk_ptr[c], rel_j + chunk_ptr[c], chunk_ptr[c + 1] - chunk_ptr[c], chunk_off[c], keep, index_of);
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// moments of inertia, embed scores

extern "C" __attribute__((visibility("default"))) int tsc_inertia_moments(tsc_ctx *c, const double *structures, int64_t n_structs, int n_atoms,
                                                                          const double *masses, double *out) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && structures && masses && out && n_structs >= 0 && n_atoms > 0, "tsc_inertia_moments: bad argument");
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_s, *d_m, *d_o;
    TSC_TRY(upload(c, s, structures, size_t(n_structs) * n_atoms * 3, &d_s));
    TSC_TRY(upload(c, s, masses, size_t(n_atoms), &d_m));
    TSC_TRY(s.get(size_t(n_structs) * 3, &d_o));
    hipLaunchKernelGGL(k_inertia_moments, dim3(grid_for(n_structs, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)d_s, n_structs, n_atoms,
                       (const double *)d_m, d_o);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(out, d_o, size_t(n_structs) * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_moi_first_similar(tsc_ctx *c, const double *moments, int64_t n_structs, double max_deviation,
                                                                            int32_t *first) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && moments && first && n_structs >= 0 && n_structs < INT32_MAX, "tsc_moi_first_similar: bad argument");
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_m;
    int32_t *d_f;
    TSC_TRY(upload(c, s, moments, size_t(n_structs) * 3, &d_m));
    TSC_TRY(s.get(size_t(n_structs), &d_f));
    hipLaunchKernelGGL(k_moi_first_similar, dim3(grid_for(n_structs, 4, 256 * 16)), dim3(256), 0, c->stream, (const double *)d_m, n_structs, max_deviation, d_f);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(first, d_f, size_t(n_structs) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_embed_scores(tsc_ctx *c, const double *structures, int64_t n_structs, int n_atoms,
                                                                       const int32_t *indices, const double *distances, int n_c, float *scores,
                                                                       double *fitness_error) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && structures && indices && distances && scores && fitness_error && n_structs >= 0 && n_atoms > 0 && n_c >= 0, "tsc_embed_scores: bad argument");
    for (int64_t q = 0; q < n_structs * n_c * 2; ++q) TSC_REQUIRE(indices[q] >= 0 && indices[q] < n_atoms, "constrained index %d out of range", indices[q]);
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_s, *d_d, *d_e;
    int32_t *d_i;
    float *d_sc;
    TSC_TRY(upload(c, s, structures, size_t(n_structs) * n_atoms * 3, &d_s));
    TSC_TRY(upload(c, s, indices, size_t(n_structs) * n_c * 2, &d_i));
    TSC_TRY(upload(c, s, distances, size_t(n_structs) * n_c, &d_d));
    TSC_TRY(s.get(size_t(n_structs), &d_sc));
    TSC_TRY(s.get(size_t(n_structs), &d_e));
    hipLaunchKernelGGL(k_embed_scores, dim3(grid_for(n_structs, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)d_s, n_structs, n_atoms,
                       (const int32_t *)d_i, (const double *)d_d, n_c, d_sc, d_e);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(scores, d_sc, size_t(n_structs) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(fitness_error, d_e, size_t(n_structs) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// string-embed pose parameters

extern "C" __attribute__((visibility("default"))) int tsc_string_embed_params_dev(tsc_ctx *c, const double *p1, const double *p2, const double *ref_vec,
                                                                                  const double *mol_vec, const int32_t *conf_pair, int64_t n_sites,
                                                                                  const double *angles, int n_angles, double *rot, double *pos,
                                                                                  int32_t *conf_idx) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && p1 && p2 && ref_vec && mol_vec && conf_pair && angles && rot && pos && conf_idx, "tsc_string_embed_params_dev: null argument");
    TSC_REQUIRE(n_sites >= 0 && n_angles >= 0, "bad sizes");
    if (n_sites == 0 || n_angles == 0) return 0;
    DeviceGuard guard(c->device);
    hipLaunchKernelGGL(k_string_embed_params, dim3(grid_for(n_sites * n_angles, 256, 256 * 8)), dim3(256), 0, c->stream, p1, p2, ref_vec, mol_vec,
                       conf_pair, n_sites, angles, n_angles, rot, pos, conf_idx);
    TSC_HIP(hipGetLastError());
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_string_embed_params(tsc_ctx *c, const double *p1, const double *p2, const double *ref_vec,
                                                                              const double *mol_vec, const int32_t *conf_pair, int64_t n_sites,
                                                                              const double *angles, int n_angles, double *rot, double *pos,
                                                                              int32_t *conf_idx) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && p1 && p2 && ref_vec && mol_vec && conf_pair && angles && rot && pos && conf_idx, "tsc_string_embed_params: null argument");
    TSC_REQUIRE(n_sites >= 0 && n_angles >= 0, "bad sizes");
    if (n_sites == 0 || n_angles == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_p1, *d_p2, *d_rv, *d_mv, *d_ang, *d_rot, *d_pos;
    int32_t *d_cp, *d_ci;
    const size_t N = size_t(n_sites) * n_angles;
    TSC_TRY(upload(c, s, p1, size_t(n_sites) * 3, &d_p1));
    TSC_TRY(upload(c, s, p2, size_t(n_sites) * 3, &d_p2));
    TSC_TRY(upload(c, s, ref_vec, size_t(n_sites) * 3, &d_rv));
    TSC_TRY(upload(c, s, mol_vec, size_t(n_sites) * 3, &d_mv));
    TSC_TRY(upload(c, s, conf_pair, size_t(n_sites) * 2, &d_cp));
    TSC_TRY(upload(c, s, angles, size_t(n_angles), &d_ang));
    TSC_TRY(s.get(N * 18, &d_rot));
    TSC_TRY(s.get(N * 6, &d_pos));
    TSC_TRY(s.get(N * 2, &d_ci));
    TSC_TRY(tsc_string_embed_params_dev(c, d_p1, d_p2, d_rv, d_mv, d_cp, n_sites, d_ang, n_angles, d_rot, d_pos, d_ci));
    TSC_HIP(hipMemcpyAsync(rot, d_rot, N * 18 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(pos, d_pos, N * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(conf_idx, d_ci, N * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_cyclical_embed_params(tsc_ctx *c, const double *start, const double *end,
                                                                                const double *direction, const double *pivot, const double *meanpoint,
                                                                                const double *r0, const double *r1, const int32_t *n_reactive,
                                                                                const double *angle, int64_t n, double *rot, double *pos) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && start && end && direction && pivot && meanpoint && r0 && r1 && n_reactive && angle && rot && pos, "tsc_cyclical_embed_params: null argument");
    TSC_REQUIRE(n >= 0, "bad size");
    for (int64_t q = 0; q < n; ++q) TSC_REQUIRE(n_reactive[q] == 1 || n_reactive[q] == 2, "row %lld: n_reactive must be 1 or 2", (long long)q);
    if (n == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    const double *host[7] = {start, end, direction, pivot, meanpoint, r0, r1};
    double *dev[7], *d_angle, *d_rot, *d_pos;
    int32_t *d_nr;
    for (int i = 0; i < 7; ++i) TSC_TRY(upload(c, s, host[i], size_t(n) * 3, &dev[i]));
    TSC_TRY(upload(c, s, n_reactive, size_t(n), &d_nr));
    TSC_TRY(upload(c, s, angle, size_t(n), &d_angle));
    TSC_TRY(s.get(size_t(n) * 9, &d_rot));
    TSC_TRY(s.get(size_t(n) * 3, &d_pos));
    hipLaunchKernelGGL(k_cyclical_embed_params, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)dev[0], (const double *)dev[1],
                       (const double *)dev[2], (const double *)dev[3], (const double *)dev[4], (const double *)dev[5], (const double *)dev[6],
                       (const int32_t *)d_nr, (const double *)d_angle, n, d_rot, d_pos);
    TSC_HIP(hipGetLastError());
    TSC_HIP(hipMemcpyAsync(rot, d_rot, size_t(n) * 9 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipMemcpyAsync(pos, d_pos, size_t(n) * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TSC_HIP(hipStreamSynchronize(c->stream));
    return 0;
    TSC_API_GUARD_END
}

// --------------------------------------------------------------------------------------------------
// the embed loops as drivers: string_embed (tscode/embeds.py:91-120) and cyclical_embed (:636-717, :771-847)

// is_new_structure over fingerprints on the device (tfd.hpp): super-blocks of TG_SUPER candidates, three launches each.
// d_acc u8[n], d_list i32[n], d_nk i32[1] (the number kept, on the device)
static int launch_tfd_greedy(tsc_ctx *c, Scratch &s, const float *d_tf, int64_t n, int T, double thresh, uint8_t *d_acc, int32_t *d_list, int32_t *d_nk) {
    TSC_REQUIRE(T <= 12288, "fingerprints of %d torsions: the greedy filter takes up to 12288", T);
    // per super-block: dead u8[TG_SUPER] and, right behind it, nz u64[TG_WORDS] -- one memset clears both
    uint8_t *d_flags;
    unsigned long long *d_sim;
    float *d_kfp;  // the kept fingerprints once more, compact and in the order they were kept (k_tfd_greedy_prior)
    TSC_TRY(s.get(size_t(TG_SUPER) + TG_WORDS * sizeof(unsigned long long), &d_flags));
    TSC_TRY(s.get(size_t(TG_SUPER) * TG_WORDS, &d_sim));
    TSC_TRY(s.get(std::max<size_t>(size_t(n) * T, 1), &d_kfp));
    int32_t *d_nk_before;
    TSC_TRY(s.get(1, &d_nk_before));
    unsigned long long *d_nz = reinterpret_cast<unsigned long long *>(d_flags + TG_SUPER);
    const size_t lds_pairs = 64 * T <= 12288 ? size_t(64) * std::max(T, 1) * sizeof(float) : 0;
    TSC_HIP(hipMemsetAsync(d_nk, 0, sizeof(int32_t), c->stream));
    for (int64_t base = 0; base < n; base += TG_SUPER) {
        const int nc = int(std::min<int64_t>(TG_SUPER, n - base));
        TSC_HIP(hipMemsetAsync(d_flags, 0, size_t(TG_SUPER) + TG_WORDS * sizeof(unsigned long long), c->stream));
        if (base > 0)
            hipLaunchKernelGGL(k_tfd_greedy_prior, dim3(ceil_div(nc, 256), 64), dim3(256), 0, c->stream, d_tf, base, nc, T, thresh, (const float *)d_kfp,
                               (const int32_t *)d_nk, d_flags);
        hipLaunchKernelGGL(k_tfd_greedy_pairs, dim3(TG_WORDS, ceil_div(nc, 256)), dim3(256), lds_pairs, c->stream, d_tf, base, nc, T, thresh, d_sim, d_nz);
        hipLaunchKernelGGL(k_tfd_greedy_replay, dim3(1), dim3(64), 0, c->stream, (const unsigned long long *)d_sim, (const unsigned long long *)d_nz, base, nc,
                           (const uint8_t *)d_flags, d_acc, d_list, d_nk, d_nk_before);
        if (base + TG_SUPER < n && T > 0)  // (the last super-block's fingerprints are compared with nothing any more)
            hipLaunchKernelGGL(k_tfd_greedy_keep, dim3(1), dim3(256), 0, c->stream, d_tf, T, (const int32_t *)d_list, (const int32_t *)d_nk_before,
                               (const int32_t *)d_nk, d_kfp);
    }
    TSC_HIP(hipGetLastError());
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_tfd_greedy_filter(tsc_ctx *c, const float *tf, int64_t n_structs, int n_quads, double thresh,
                                                                            uint8_t *accepted, int64_t *n_kept) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && tf && accepted, "tsc_tfd_greedy_filter: null argument");
    TSC_REQUIRE(n_structs >= 0 && n_structs < INT32_MAX && n_quads >= 0, "bad sizes");
    if (n_kept) *n_kept = 0;
    if (n_structs == 0) return 0;
    DeviceGuard guard(c->device);
    Scratch s(c);
    float *d_tf;
    uint8_t *d_acc;
    int32_t *d_list, *d_nk;
    TSC_TRY(upload(c, s, tf, std::max<size_t>(size_t(n_structs) * n_quads, 1), &d_tf));
    TSC_TRY(s.get(size_t(n_structs), &d_acc));
    TSC_TRY(s.get(size_t(n_structs), &d_list));
    TSC_TRY(s.get(1, &d_nk));
    TSC_TRY(launch_tfd_greedy(c, s, d_tf, n_structs, n_quads, thresh, d_acc, d_list, d_nk));
    TSC_HIP(hipMemcpyAsync(accepted, d_acc, size_t(n_structs), hipMemcpyDeviceToHost, c->stream));
    int32_t nk = 0;
    TSC_TRY(read_i32(c, d_nk, &nk));
    if (n_kept) *n_kept = nk;
    return 0;
    TSC_API_GUARD_END
}

// What both drivers share once the pose parameters are on the device: clash verdicts of all candidates, the passing poses
// embedded in candidate order, a filter over them (`filter(d_structs, n_pass, d_pos_scan, d_acc)`), the kept poses compacted
// and copied out, both verdicts per candidate.
template <typename Filter>
static int embed_filter_run(tsc_ctx *c, Scratch &s, const double *d_frags, const FragTable &ft, const int64_t *frag_off, const int32_t *n_atoms,
                            const int32_t *n_conf, const int32_t *d_ci, const double *d_rot, const double *d_pos, int64_t N, double clash_thresh,
                            int64_t max_clashes, uint8_t *clash_ok, uint8_t *kept, double *poses, int64_t poses_capacity, int64_t *n_pass_out,
                            int64_t *n_kept_out, Filter filter) {
    hipStream_t st = c->stream;
    const int n = ft.n_total;
    // copies into the CALLER's host arrays are enqueued long before this function returns: whatever path leaves it -- an error
    // included -- the stream is idle first, so that no copy lands in memory the caller has meanwhile freed
    struct SyncOnExit {
        hipStream_t st;
        ~SyncOnExit() { (void)hipStreamSynchronize(st); }
    } sync_on_exit{st};
    uint8_t *d_mask, *d_kept_full, *d_acc;
    int32_t *bsum, *act, *pos_scan, *total, *act2, *total2;
    TSC_TRY(s.get(size_t(N), &d_mask));
    TSC_TRY(s.get(size_t(N), &d_kept_full));
    TSC_TRY(s.get(scan_bsum_count(N), &bsum));
    TSC_TRY(s.get(size_t(N), &act));
    TSC_TRY(s.get(size_t(N) + 1, &pos_scan));
    TSC_TRY(s.get(1, &total));
    TSC_TRY(s.get(size_t(N), &act2));
    TSC_TRY(s.get(1, &total2));
    TSC_TRY(tsc_embed_clash_mask_dev(c, d_frags, frag_off, n_atoms, n_conf, ft.n_mols, d_ci, d_rot, d_pos, N, clash_thresh, max_clashes, d_mask, nullptr));
    TSC_TRY(scan_mask(st, d_mask, N, bsum, pos_scan, act, nullptr, total));
    TSC_HIP(hipMemcpyAsync(clash_ok, d_mask, size_t(N), hipMemcpyDeviceToHost, st));
    TSC_HIP(hipMemsetAsync(d_kept_full, 0, size_t(N), st));
    int32_t n_pass = 0, n_kept = 0;
    TSC_TRY(read_i32(c, total, &n_pass));
    *n_pass_out = n_pass;
    if (n_pass > 0) {
        double *d_structs, *d_out;
        TSC_TRY(s.get(size_t(n_pass) * n * 3, &d_structs));
        TSC_TRY(s.get(size_t(n_pass), &d_acc));
        hipLaunchKernelGGL(k_transform, dim3(grid_for(n_pass, TR_POSES, 256 * 64)), dim3(256), transform_lds_bytes(ft.n_mols), st, d_frags, ft, d_ci, d_rot, d_pos,
                           (const int32_t *)act, int64_t(n_pass), d_structs, (const int32_t *)nullptr, 0, (double *)nullptr, (const int32_t *)nullptr);
        TSC_HIP(hipGetLastError());
        TSC_TRY(filter(d_structs, n_pass, pos_scan, total, d_acc));
        hipLaunchKernelGGL(k_scatter_flags, dim3(grid_for(n_pass, 256, 1024)), dim3(256), 0, st, (const uint8_t *)d_acc, (const int32_t *)act, (const int32_t *)total,
                           d_kept_full);
        TSC_TRY(scan_mask(st, d_acc, n_pass, bsum, nullptr, act2, nullptr, total2));
        TSC_TRY(read_i32(c, total2, &n_kept));
        if (poses && n_kept > 0) {
            TSC_REQUIRE(n_kept <= poses_capacity, "poses holds %lld rows, %d poses were kept", (long long)poses_capacity, n_kept);
            TSC_TRY(s.get(size_t(n_kept) * n * 3, &d_out));
            TSC_TRY(launch_gather_rows(st, d_structs, act2, n_kept, n * 3, nullptr, n * 3, d_out));
            TSC_HIP(hipMemcpyAsync(poses, d_out, size_t(n_kept) * n * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        }
    }
    TSC_HIP(hipMemcpyAsync(kept, d_kept_full, size_t(N), hipMemcpyDeviceToHost, st));
    TSC_HIP(hipStreamSynchronize(st));
    *n_kept_out = n_kept;
    return 0;
}

extern "C" __attribute__((visibility("default"))) int tsc_string_embed(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                                                       const int32_t *n_conf, const double *p1, const double *p2, const double *ref_vec,
                                                                       const double *mol_vec, const int32_t *conf_pair, int64_t n_sites, const double *angles,
                                                                       int n_angles, double clash_thresh, int64_t max_clashes, const int32_t *quads, int n_quads,
                                                                       double tfd_thresh, uint8_t *clash_ok, uint8_t *kept, double *poses,
                                                                       int64_t poses_capacity, int64_t *n_pass, int64_t *n_kept) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && frag_off && n_atoms && n_conf && p1 && p2 && ref_vec && mol_vec && conf_pair && angles && clash_ok && kept && n_pass && n_kept,
                "tsc_string_embed: null argument");
    TSC_REQUIRE(n_sites >= 0 && n_angles >= 0 && n_quads >= 0 && (n_quads == 0 || quads), "bad sizes");
    const int64_t N = n_sites * n_angles;
    TSC_REQUIRE(N < INT32_MAX, "too many candidates");
    *n_pass = *n_kept = 0;
    if (N == 0) return 0;
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, 2, &ft));
    for (int64_t q = 0; q < n_sites; ++q)
        TSC_REQUIRE(conf_pair[2 * q] >= 0 && conf_pair[2 * q] < n_conf[0] && conf_pair[2 * q + 1] >= 0 && conf_pair[2 * q + 1] < n_conf[1],
                    "site %lld: conformer index out of range", (long long)q);
    for (int q = 0; q < 4 * n_quads; ++q) TSC_REQUIRE(quads[q] >= 0 && quads[q] < ft.n_total, "quadruplet atom index %d out of range", quads[q]);
    DeviceGuard guard(c->device);
    Scratch s(c);
    double *d_frags, *d_p1, *d_p2, *d_rv, *d_mv, *d_ang, *d_rot, *d_pos;
    int32_t *d_cp, *d_ci, *d_quads = nullptr;
    TSC_TRY(upload(c, s, frags, size_t(frags_total_doubles(frag_off, n_atoms, n_conf, 2)), &d_frags));
    TSC_TRY(upload(c, s, p1, size_t(n_sites) * 3, &d_p1));
    TSC_TRY(upload(c, s, p2, size_t(n_sites) * 3, &d_p2));
    TSC_TRY(upload(c, s, ref_vec, size_t(n_sites) * 3, &d_rv));
    TSC_TRY(upload(c, s, mol_vec, size_t(n_sites) * 3, &d_mv));
    TSC_TRY(upload(c, s, conf_pair, size_t(n_sites) * 2, &d_cp));
    TSC_TRY(upload(c, s, angles, size_t(n_angles), &d_ang));
    if (n_quads) TSC_TRY(upload(c, s, quads, size_t(n_quads) * 4, &d_quads));
    TSC_TRY(s.get(size_t(N) * 18, &d_rot));
    TSC_TRY(s.get(size_t(N) * 6, &d_pos));
    TSC_TRY(s.get(size_t(N) * 2, &d_ci));
    TSC_TRY(tsc_string_embed_params_dev(c, d_p1, d_p2, d_rv, d_mv, d_cp, n_sites, d_ang, n_angles, d_rot, d_pos, d_ci));
    const int n = ft.n_total;
    // is_new_structure (:47-69, :119): torsion fingerprints of the passing poses, then the greedy filter over the whole list
    auto filter = [&](const double *d_structs, int32_t np, const int32_t *, const int32_t *, uint8_t *d_acc) -> int {
        float *d_tf;
        int32_t *d_list, *d_nk;
        TSC_TRY(s.get(std::max<size_t>(size_t(np) * n_quads, 1), &d_tf));
        TSC_TRY(s.get(size_t(np), &d_list));
        TSC_TRY(s.get(1, &d_nk));
        if (n_quads)
            hipLaunchKernelGGL(k_torsion_fingerprints, dim3(grid_for(int64_t(np) * n_quads, 256, 256 * 8)), dim3(256), 0, c->stream, d_structs, int64_t(np), n,
                               (const int32_t *)d_quads, n_quads, d_tf);
        TSC_HIP(hipGetLastError());
        return launch_tfd_greedy(c, s, d_tf, int64_t(np), n_quads, tfd_thresh, d_acc, d_list, d_nk);
    };
    return embed_filter_run(c, s, d_frags, ft, frag_off, n_atoms, n_conf, d_ci, d_rot, d_pos, N, clash_thresh, max_clashes, clash_ok, kept, poses, poses_capacity,
                            n_pass, n_kept, filter);
    TSC_API_GUARD_END
}

extern "C" __attribute__((visibility("default"))) int tsc_cyclical_embed(tsc_ctx *c, const double *frags, const int64_t *frag_off, const int32_t *n_atoms,
                                                                         const int32_t *n_conf, int n_mols, const double *start, const double *end,
                                                                         const double *direction, const double *pivot, const double *meanpoint, const double *r0,
                                                                         const double *r1, const int32_t *n_reactive, const double *angle, const int32_t *conf_idx,
                                                                         int64_t n_poses, const int32_t *group_off, int n_groups, double clash_thresh,
                                                                         int64_t max_clashes, double rmsd_thr, uint8_t *clash_ok, uint8_t *kept, double *poses,
                                                                         int64_t poses_capacity, int64_t *n_pass, int64_t *n_kept) {
    TSC_API_GUARD_BEGIN
    TSC_REQUIRE(c && frags && frag_off && n_atoms && n_conf && start && end && direction && pivot && meanpoint && r0 && r1 && n_reactive && angle && conf_idx &&
                    group_off && clash_ok && kept && n_pass && n_kept,
                "tsc_cyclical_embed: null argument");
    TSC_REQUIRE(n_poses >= 0 && n_poses < INT32_MAX && n_groups >= 0 && rmsd_thr > 0, "bad sizes");
    *n_pass = *n_kept = 0;
    if (n_poses == 0) return 0;
    FragTable ft;
    TSC_TRY(make_frag_table(frag_off, n_atoms, n_conf, n_mols, &ft));
    const int64_t rows = n_poses * n_mols;
    for (int64_t q = 0; q < rows; ++q) {
        TSC_REQUIRE(n_reactive[q] == 1 || n_reactive[q] == 2, "row %lld: n_reactive must be 1 or 2", (long long)q);
        TSC_REQUIRE(conf_idx[q] >= 0 && conf_idx[q] < n_conf[q % n_mols], "row %lld: conformer index out of range", (long long)q);
    }
    TSC_REQUIRE(n_groups > 0 && group_off[0] == 0 && group_off[n_groups] == n_poses, "group_off must run from 0 to n_poses");
    for (int g = 0; g < n_groups; ++g)
        TSC_REQUIRE(group_off[g + 1] >= group_off[g] && group_off[g + 1] - group_off[g] <= GF_MAX_GROUP, "group %d: sizes must be in [0, %d]", g, GF_MAX_GROUP);
    DeviceGuard guard(c->device);
    Scratch s(c);
    const double *host[7] = {start, end, direction, pivot, meanpoint, r0, r1};
    double *d_frags, *dev[7], *d_angle, *d_rot, *d_pos;
    int32_t *d_nr, *d_ci, *d_goff, *d_goff_pass;
    TSC_TRY(upload(c, s, frags, size_t(frags_total_doubles(frag_off, n_atoms, n_conf, n_mols)), &d_frags));
    for (int i = 0; i < 7; ++i) TSC_TRY(upload(c, s, host[i], size_t(rows) * 3, &dev[i]));
    TSC_TRY(upload(c, s, n_reactive, size_t(rows), &d_nr));
    TSC_TRY(upload(c, s, angle, size_t(rows), &d_angle));
    TSC_TRY(upload(c, s, conf_idx, size_t(rows), &d_ci));
    TSC_TRY(upload(c, s, group_off, size_t(n_groups) + 1, &d_goff));
    TSC_TRY(s.get(size_t(n_groups) + 1, &d_goff_pass));
    TSC_TRY(s.get(size_t(rows) * 9, &d_rot));
    TSC_TRY(s.get(size_t(rows) * 3, &d_pos));
    hipLaunchKernelGGL(k_cyclical_embed_params, dim3(grid_for(rows, 256, 256 * 8)), dim3(256), 0, c->stream, (const double *)dev[0], (const double *)dev[1],
                       (const double *)dev[2], (const double *)dev[3], (const double *)dev[4], (const double *)dev[5], (const double *)dev[6],
                       (const int32_t *)d_nr, (const double *)d_angle, rows, d_rot, d_pos);
    TSC_HIP(hipGetLastError());
    const int n = ft.n_total;
    // not _rmsd_similarity(pose, angular_poses, rmsd_thr=1) (:715, :843): greedy inside each group of passing poses
    auto filter = [&](const double *d_structs, int32_t np, const int32_t *pos_scan, const int32_t *total, uint8_t *d_acc) -> int {
        hipLaunchKernelGGL(k_group_offsets_after_filter, dim3(grid_for(n_groups + 1, 256, 1024)), dim3(256), 0, c->stream, (const int32_t *)d_goff, n_groups, pos_scan,
                           n_poses, total, d_goff_pass);
        TSC_HIP(hipGetLastError());
        return tsc_greedy_group_filter_dev(c, d_structs, d_goff_pass, n_groups, np, n, rmsd_thr, d_acc);
    };
    return embed_filter_run(c, s, d_frags, ft, frag_off, n_atoms, n_conf, d_ci, d_rot, d_pos, n_poses, clash_thresh, max_clashes, clash_ok, kept, poses,
                            poses_capacity, n_pass, n_kept, filter);
    TSC_API_GUARD_END
}

