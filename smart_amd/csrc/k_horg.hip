// k_horg.hip — Horspool on GRAMS, for texts of at most four distinct byte values: hor_scan_gram (the loop: gram_skip.hpp)
// (one translation unit per kernel family: dev_common.hpp — a unit of its own so that the code object of hor_scan, the
// headline kernel, stays what it is)
#include "gram_skip.hpp"

namespace sg {

template <int THREADS, int L, bool LONG, int Q, int GRAM>  // LONG: m > 32 — the window does not lie in the lane's column
__global__ __launch_bounds__(THREADS) void hor_scan_gram(ScanArgs a1, uint64_t tile_first, uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    gram_skip_scan<THREADS, L, LONG, Q, GRAM, false>(a, tile_first, ntiles, smem);
}

// ---------------------------------------------------------------------------
// launcher: gram = 1 (eight one-bit symbols, m >= 8) / 2 (four two-bit symbols, m >= 4), chosen by launch_scan from the text's codes
// ---------------------------------------------------------------------------
hipError_t launch_hor_gram(const ScanArgs& a, int gram, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m;
    // workgroups per CU as bndm_scan's gram form: short windows iterate (six), long ones stream (four)
    const int wgs = m >= 32 ? 4 : m >= 16 ? 5 : 6;
    const size_t lds = 1072 + ColTile<kBndmT>::bytes();
    const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBndmT * kBndmL);
    if (gram == 1) {
        if (m > 32) return launch_tiled(hor_scan_gram<kBndmT, kBndmL, true, 8, 1>, a, tr, kBndmT, lds, wgs, num_cus, stream);
        return launch_tiled(hor_scan_gram<kBndmT, kBndmL, false, 8, 1>, a, tr, kBndmT, lds, wgs, num_cus, stream);
    }
    if (m > 32) return launch_tiled(hor_scan_gram<kBndmT, kBndmL, true, 4, 2>, a, tr, kBndmT, lds, wgs, num_cus, stream);
    return launch_tiled(hor_scan_gram<kBndmT, kBndmL, false, 4, 2>, a, tr, kBndmT, lds, wgs, num_cus, stream);
}

}  // namespace sg
