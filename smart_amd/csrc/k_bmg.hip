// k_bmg.hip — Boyer-Moore on GRAMS, for texts of at most four distinct byte values: bm_scan_gram (the loop: gram_skip.hpp)
// (one translation unit per kernel family: dev_common.hpp)
#include "gram_skip.hpp"

namespace sg {

template <int THREADS, int L, bool LONG, int Q, int GRAM>  // LONG: m > 32 — the window does not lie in the lane's column
__global__ __launch_bounds__(THREADS) void bm_scan_gram(ScanArgs a1, uint64_t tile_first, uint32_t ntiles, const BatchItem* __restrict__ batch)
{
    const ScanArgs a = pick_args(a1, batch);  // a pattern set in one grid: blockIdx.y = pattern (launch_batch)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    gram_skip_scan<THREADS, L, LONG, Q, GRAM, true>(a, tile_first, ntiles, smem);
}

// ---------------------------------------------------------------------------
// launcher: gram = 1 (eight one-bit symbols, m >= 16) / 2 (four two-bit symbols, m >= 8), chosen by launch_scan from the text's codes
// ---------------------------------------------------------------------------
hipError_t launch_bm_gram(const ScanArgs& a, int gram, int num_cus, hipStream_t stream)
{
    const uint32_t m = a.m;
    const int wgs = m >= 32 ? 4 : m >= 16 ? 5 : 6;  // as launch_hor_gram
    const size_t lds = 1168 + ColTile<kBndmT>::bytes();
    const TileRange tr = tiles_for(a.s_begin + m - 1, a.s_end + m - 1, (uint64_t)kBndmT * kBndmL);
    if (gram == 1) {
        if (m > 32) return launch_tiled(bm_scan_gram<kBndmT, kBndmL, true, 8, 1>, a, tr, kBndmT, lds, wgs, num_cus, stream);
        return launch_tiled(bm_scan_gram<kBndmT, kBndmL, false, 8, 1>, a, tr, kBndmT, lds, wgs, num_cus, stream);
    }
    if (m > 32) return launch_tiled(bm_scan_gram<kBndmT, kBndmL, true, 4, 2>, a, tr, kBndmT, lds, wgs, num_cus, stream);
    return launch_tiled(bm_scan_gram<kBndmT, kBndmL, false, 4, 2>, a, tr, kBndmT, lds, wgs, num_cus, stream);
}

}  // namespace sg
