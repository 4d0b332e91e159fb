// pairs_tile.hip -- the register-tiled all-pairs kernel k_rmsd_tile<HP, 16> (rmsd.hpp), one instantiation per padded heavy-atom count.
// gfx950 only.  A translation unit of its own: the eight instantiations compile beside the other pair kernels, not behind them.
#include "prune_host.hpp"

template <int HP>
static void launch_tile(hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *Xr, const double *Xc, const double *G, const int32_t *cend,
                        int32_t *best, PassCounters *counters, const PruneState *state, const TileArgs &a) {
    hipExtLaunchKernelGGL((k_rmsd_tile<HP, TILE_ROWS>), grid, dim3(256), 0, st, e0, e1, 0, Xr, Xc, G, cend, best, counters, state, a);
}

int launch_rmsd_tile(int hp, hipStream_t st, dim3 grid, hipEvent_t e0, hipEvent_t e1, const double *Xr, const double *Xc, const double *G, const int32_t *cend,
                     int32_t *best, PassCounters *counters, const PruneState *state, const TileArgs &a) {
    switch (hp) {
        case 4: launch_tile<4>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        case 8: launch_tile<8>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        case 12: launch_tile<12>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        case 16: launch_tile<16>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        case 20: launch_tile<20>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        case 24: launch_tile<24>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        case 28: launch_tile<28>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        case 32: launch_tile<32>(st, grid, e0, e1, Xr, Xc, G, cend, best, counters, state, a); break;
        default: return fail(TSC_ERR_INVALID, "unsupported padded atom count %d", hp);
    }
    TSC_HIP(hipGetLastError());
    return 0;
}
