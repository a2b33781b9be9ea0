// tables.cpp — host-side preprocessing for the scan kernels.  See tables.hpp.
#include "tables.hpp"

#include <algorithm>
#include <cstring>

namespace sg {

std::vector<int32_t> bad_char(const uint8_t* P, uint32_t m)
{
    std::vector<int32_t> shift(kSigma, static_cast<int32_t>(m));
    for (uint32_t i = 0; i + 1 < m; ++i) shift[P[i]] = static_cast<int32_t>(m - 1 - i);
    return shift;
}

// suff[i] = |longest common suffix of P[0..i] and P|.  Computed as the
// Z-array of the reversed pattern: suff[i] = Z[m-1-i].
static std::vector<int32_t> suffix_lengths(const uint8_t* P, uint32_t m)
{
    const int32_t M = static_cast<int32_t>(m);
    std::vector<uint8_t> r(P, P + m);
    std::reverse(r.begin(), r.end());
    std::vector<int32_t> z(m, 0);
    z[0] = M;
    int32_t lo = 0, hi = 0;  // [lo,hi) = right-most Z-box
    for (int32_t i = 1; i < M; ++i) {
        int32_t k = 0;
        if (i < hi) k = std::min(z[i - lo], hi - i);
        while (i + k < M && r[k] == r[i + k]) ++k;
        z[i] = k;
        if (i + k > hi) { lo = i; hi = i + k; }
    }
    std::vector<int32_t> suff(m);
    for (int32_t i = 0; i < M; ++i) suff[i] = z[M - 1 - i];
    return suff;
}

std::vector<int32_t> good_suffix(const uint8_t* P, uint32_t m)
{
    const int32_t M = static_cast<int32_t>(m);
    const std::vector<int32_t> suff = suffix_lengths(P, m);
    std::vector<int32_t> gs(m, M);
    // a border of P (prefix == suffix of length i+1) bounds the shift for every
    // mismatch position left of it
    int32_t j = 0;
    for (int32_t i = M - 1; i >= 0; --i) {
        if (suff[i] != i + 1) continue;
        for (; j < M - 1 - i; ++j)
            if (gs[j] == M) gs[j] = M - 1 - i;
    }
    // an inner re-occurrence of the matched suffix
    for (int32_t i = 0; i <= M - 2; ++i) gs[M - 1 - suff[i]] = M - 1 - i;
    return gs;
}

std::vector<int32_t> kmp_next(const uint8_t* P, uint32_t m)
{
    // border[i] = length of the longest proper border of P[0..i)
    std::vector<int32_t> border(m + 1, 0);
    border[0] = -1;
    int32_t b = -1;
    for (uint32_t i = 0; i < m; ++i) {
        while (b >= 0 && P[i] != P[b]) b = border[b];
        border[i + 1] = ++b;
    }
    // strong links: never fall back to a state that fails on the same byte
    std::vector<int32_t> next(m + 1);
    next[0] = -1;
    for (uint32_t i = 1; i <= m; ++i) {
        const int32_t bi = border[i];
        next[i] = (i < m && P[i] == P[bi]) ? next[bi] : bi;
    }
    return next;
}

std::vector<uint8_t> kmp_dfa(const uint8_t* P, uint32_t m)
{
    // border[s] = longest proper border of P[0..s); delta(s,c) = s+1 on a match,
    // otherwise delta(border[s], c) — the failure links followed once and for all
    std::vector<int32_t> border(m + 1, 0);
    border[0] = -1;
    int32_t b = -1;
    for (uint32_t i = 0; i < m; ++i) {
        while (b >= 0 && P[i] != P[b]) b = border[b];
        border[i + 1] = ++b;
    }
    std::vector<uint8_t> dfa(static_cast<size_t>(m + 1) * 256, 0);
    dfa[P[0]] = 1;
    for (uint32_t s = 1; s <= m; ++s) {
        const uint8_t* from = &dfa[static_cast<size_t>(border[s]) * 256];
        uint8_t* row = &dfa[static_cast<size_t>(s) * 256];
        for (int c = 0; c < 256; ++c) row[c] = from[c];
        if (s < m) row[P[s]] = static_cast<uint8_t>(s + 1);
    }
    return dfa;
}

uint32_t kmp_runs_table_bytes(uint32_t w) { return (w < 63 ? w + 1 : 256u) * 256u; }

// The two-bit codes of a text of at most four distinct byte values (so_runs<., FOUR>, kmp_runs<., FOUR>: four text
// bytes per table step, the codes (c >> shift) & 3 of a dword's bytes make an 8-bit index — the kernels derive their
// four-byte tables from the byte tables when they start).
bool four_symbol_codes(const uint32_t set[8], uint32_t* shift, uint32_t* symtab)
{
    uint8_t syms[4];
    uint32_t distinct = 0;
    for (uint32_t c = 0; c < 256; ++c)
        if (set[c >> 5] >> (c & 31) & 1u) {
            if (distinct == 4) return false;
            syms[distinct++] = static_cast<uint8_t>(c);
        }
    for (uint32_t sh = 0; sh < 7; ++sh) {
        int sym_of[4] = {-1, -1, -1, -1};
        bool ok = true;
        for (uint32_t i = 0; i < distinct && ok; ++i) {
            const uint32_t c = (syms[i] >> sh) & 3u;
            ok = sym_of[c] < 0;
            sym_of[c] = syms[i];
        }
        if (!ok) continue;
        *shift = sh;
        *symtab = 0;
        for (uint32_t c = 0; c < 4; ++c) *symtab |= static_cast<uint32_t>(sym_of[c] >= 0 ? sym_of[c] : static_cast<int>(c << sh)) << (8 * c);
        return true;
    }
    return false;
}

void kmp_runs_tables(const uint8_t* P, uint32_t w, std::vector<uint8_t>& out, bool compact)
{
    // ids: fewer than 63 states: id(s) = 4s, Z = 4w + 1, the table ends there.  Otherwise id(s) = rotl8(s, 2) — the
    // low states a lane is usually in then differ in the bits that select the LDS bank —, id(w) = 254, Z = 255;
    // rotl8 maps only s = 191 to 254 and only s = 255 to 255, so state 191 (if there is one besides w) takes the
    // slot w gave up.  Z = id(w) + 1 is the largest id (the kernel's min(next, id(w)) turns Z back into w).
    const bool small = w < 63;
    const uint32_t idw = small ? 4 * w : 254u, Z = (compact && small) ? idw + 4 : idw + 1;
    uint8_t id[256];
    for (uint32_t st = 0; st <= w; ++st) id[st] = static_cast<uint8_t>(small ? 4 * st : ((st << 2) | (st >> 6)) & 255u);
    if (!small && w > 191) id[191] = id[w];
    id[w] = static_cast<uint8_t>(idw);
    // bd[s] = longest proper border of P[0..s) (kmp.c:27-41 without the kmpNext[i] = kmpNext[j] shortcut)
    uint8_t bd[256];
    bd[0] = bd[1] = 0;
    for (uint32_t s = 2, k = 0; s <= w; ++s) {
        while (k && P[s - 1] != P[k]) k = bd[k];
        if (P[s - 1] == P[k]) ++k;
        bd[s] = static_cast<uint8_t>(k);
    }
    // Stored rows: with fewer than 63 states row s of the blob is the table's row 4s (the kernel spreads them out
    // in LDS and fills row Z itself: a quarter of the bytes to build, stage, upload and fetch); otherwise all 256.
    const size_t base = out.size();
    out.resize(base + kmp_runs_table_bytes(w) + 272, 0);
    uint8_t* const tab = out.data() + base;
    auto into = [&](uint32_t st) { return static_cast<uint8_t>(st == w ? Z : id[st]); };  // a transition INTO st
    auto row = [&](uint32_t st) { return tab + (small ? st : id[st]) * 256u; };
    // state 0 (row 0, swizzle 0): everything to 0 except P[0]
    tab[P[0]] = into(1);
    for (uint32_t s = 1; s <= w; ++s) {
        // delta(s, c) = s+1 on a match, otherwise delta(border(s), c): the border's finished row, re-swizzled —
        // entry c sits at c ^ id, so dst[j] = src[j ^ id(s) ^ id(border)]
        const uint32_t r = id[s], rb = id[bd[s]], d = r ^ rb;
        const uint8_t* src = row(bd[s]);
        uint8_t* dst = row(s);
        if ((d & 3) == 0) {
            for (uint32_t j = 0; j < 256; j += 4) std::memcpy(dst + j, src + (j ^ d), 4);
        } else {  // the ids of 63+ states are not multiples of 4: the bytes of every dword change places too
            for (uint32_t j = 0; j < 256; j += 4) {
                uint32_t x;
                std::memcpy(&x, src + ((j ^ d) & ~3u), 4);
                if (d & 1) x = ((x & 0x00FF00FFu) << 8) | ((x >> 8) & 0x00FF00FFu);
                if (d & 2) x = (x << 16) | (x >> 16);
                std::memcpy(dst + j, &x, 4);
            }
        }
        if (s < w) dst[P[s] ^ r] = into(s + 1);
    }
    if (!small) std::memset(tab + Z * 256, static_cast<int>(Z), 256);
    // the four-bytes-at-a-time forms (kmp_chunk_skip4): K + 4 < w (no occurrence ends in the dword) and K + 4 <= 62
    // (the ids are 4s up there in both numberings)
    uint8_t* const q = tab + kmp_runs_table_bytes(w);
    if (w >= 5) {
        const uint32_t cap = std::min<uint32_t>(w - 5, 58);
        uint32_t K = 0;
        while (K < cap && bd[K + 1] == 0) ++K;
        for (uint32_t s = 0; s <= K; ++s) std::memcpy(q + 4 * s, P + s, 4);  // the dword as the text holds it
        const uint32_t thr = 4 * K;
        std::memcpy(q + 256, &thr, 4);
    }
}

std::vector<uint8_t> kmp_dfa_compressed(const uint8_t* P, uint32_t m, uint32_t* k1)
{
    const std::vector<uint8_t> full = kmp_dfa(P, m);
    std::vector<uint8_t> out(256, 0);  // colmap
    uint32_t ncol = 1;                 // column 0: bytes that do not occur in P
    for (uint32_t i = 0; i < m; ++i)
        if (out[P[i]] == 0) out[P[i]] = static_cast<uint8_t>(ncol++);  // at most 255 distinct for m <= 255
    *k1 = ncol;
    out.resize(256 + static_cast<size_t>(m + 1) * ncol, 0);
    for (uint32_t s = 0; s <= m; ++s)
        for (int c = 0; c < 256; ++c)
            if (out[c] != 0) out[256 + static_cast<size_t>(s) * ncol + out[c]] = full[static_cast<size_t>(s) * 256 + c];
    return out;
}

std::vector<uint32_t> shift_or_masks(const uint8_t* P, uint32_t m)
{
    const uint32_t w = std::min<uint32_t>(m, 32);
    std::vector<uint32_t> S(kSigma, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < w; ++i) S[P[i]] &= ~(1u << i);
    return S;
}

std::vector<uint32_t> bndm_masks(const uint8_t* P, uint32_t m)
{
    const uint32_t w = std::min<uint32_t>(m, 32);
    std::vector<uint32_t> B(kSigma, 0u);
    for (uint32_t i = 0; i < w; ++i) B[P[i]] |= 1u << (w - 1 - i);
    return B;
}

std::vector<uint32_t> shift_and_masks(const uint8_t* P, uint32_t m)
{
    const uint32_t w = std::min<uint32_t>(m, 32);
    std::vector<uint32_t> S(kSigma, 0u);
    for (uint32_t i = 0; i < w; ++i) S[P[i]] |= 1u << i;
    return S;
}

std::vector<int32_t> quick_search_shifts(const uint8_t* P, uint32_t m)
{
    std::vector<int32_t> shift(kSigma, static_cast<int32_t>(m + 1));
    for (uint32_t i = 0; i < m; ++i) shift[P[i]] = static_cast<int32_t>(m - i);
    return shift;
}

std::vector<int32_t> qgram_hash_shifts(const uint8_t* P, uint32_t m, uint32_t q, int32_t* after)
{
    auto hash = [&](uint32_t end) {  // q-gram ending at P[end]
        uint32_t h = 0;
        for (uint32_t k = 0; k < q; ++k) h = (h << 1) + P[end + 1 - q + k];
        return h & 0xFFu;
    };
    std::vector<int32_t> shift(kSigma, static_cast<int32_t>(m - q + 1));
    for (uint32_t i = q - 1; i + 1 < m; ++i) shift[hash(i)] = static_cast<int32_t>(m - 1 - i);
    const uint32_t hl = hash(m - 1);
    *after = shift[hl] == 0 ? 1 : shift[hl];
    shift[hl] = 0;
    return shift;
}

}  // namespace sg
