// Host-side model compiler: CSC W + C  ->  tiled rank-bitmap layout in HBM (see xrl_model.h).
//
// Mirrors the load-time work of the reference without sharing its data structures:
//   LayerData<chunked>::init                pecos/core/xmc/inference.hpp:1849-1883
//   check_if_contiguously_ordered           :658-668
//   rearrangement_t::initialize_from_codes  :1746-1761   (perm / perm_inv)
//   make_chunked_from_csc                   :557-650     (transpose-by-parent, rows ascending,
//                                                         columns ascending inside a row)
//   check_bias_explicit                     :500-502
#include "xrl_model.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <thread>

namespace xrl {

PostProc parse_post_processor(const char* name_c) {
    // PostProcessor<T>::get, inference.hpp:192-240.  Unknown names yield the default-constructed
    // processor there (identity transform, combiner keeps x) -> PP_NOOP here.
    PostProc pp;
    if (!name_c) return pp;
    const std::string name(name_c);
    auto ends = [&](const char* s) { const size_t n = std::strlen(s); return name.size() >= n && name.compare(name.size() - n, n, s) == 0; };
    if (name == "noop") return pp;
    if (name == "sigmoid") { pp.kind = PP_SIGMOID; return pp; }
    if (name == "log-sigmoid") { pp.kind = PP_LOG_SIGMOID; return pp; }
    if (name.rfind("log-l", 0) == 0 && ends("-hinge")) {
        pp.kind = PP_LOG_LP_HINGE;
        pp.p = std::atoi(name.substr(5, name.size() - 5 - 6).c_str());
        return pp;
    }
    if (name.rfind("l", 0) == 0 && ends("-hinge")) {
        pp.kind = PP_LP_HINGE;
        pp.p = std::atoi(name.substr(1, name.size() - 1 - 6).c_str());
        return pp;
    }
    return pp;
}

uint64_t Layer::cand_bound(uint32_t beam) const {
    uint64_t s = 0;
    for (uint32_t i = 0; i < beam && i < chunk_sizes_desc.size(); ++i) s += chunk_sizes_desc[i];
    return s;
}

Model::Model() {}
Model::~Model() {
    replicas.clear();                       // each replica releases its objects with ITS device current
    (void)hipSetDevice(device);
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    if (host_lanes.done[1] && host_lanes.done[1] != ws_done) (void)hipEventDestroy(host_lanes.done[1]);   // (done[0] aliases ws_done)
    if (host_lanes.join) (void)hipEventDestroy(host_lanes.join);
    if (aux_stream) (void)hipStreamDestroy(aux_stream);
    if (ws_done) (void)hipEventDestroy(ws_done);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (d2h_stream) (void)hipStreamDestroy(d2h_stream);
    for (hipEvent_t e : d2h_events) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
    if (fb_host) (void)hipHostFree(fb_host);
}
uint64_t Model::device_bytes() const {
    uint64_t b = 0;
    for (auto& l : layers) b += l->device_bytes;
    return b;
}

namespace {
template <class F> void parallel_for(size_t n, F&& fn) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 64) nt = 64;
    if (n < 2 * nt) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr err;
    std::mutex emu;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&] {
            try {
                for (;;) {
                    const size_t i0 = next.fetch_add(16);
                    if (i0 >= n) break;
                    for (size_t i = i0; i < std::min(n, i0 + 16); ++i) fn(i);
                }
            } catch (...) { std::lock_guard<std::mutex> g(emu); err = std::current_exception(); }
        });
    for (auto& t : th) t.join();
    if (err) std::rethrow_exception(err);
}
struct Nz { uint32_t row, col; float val; };
}  // namespace

// K1 packs {entry offset, count} of a row unit into 32 bits: tile-relative entry offsets stay below 2^25.
// XRL_MAX_TILE_ENTRIES lowers the limit (tests of the tile splitter).
static uint64_t max_tile_entries() {
    static const uint64_t v = [] {
        const char* e = std::getenv("XRL_MAX_TILE_ENTRIES");
        const uint64_t lim = (1ull << 25) - 64;
        if (!e) return lim;
        const uint64_t x = std::strtoull(e, nullptr, 10);
        return x >= 2 && x < lim ? x : lim;
    }();
    return v;
}

// Even split of a chunk of n columns into column tiles: the smallest count (starting from ceil(n / kMaxTileCols), grown
// by a quarter at a time) for which every tile [n*t/k, n*(t+1)/k) holds fewer than `limit` entries; cum[0..n] are the
// prefix sums of the columns' entry counts.  Returns 0 when even one column per tile does not fit.
uint32_t split_chunk(const uint64_t* cum, uint32_t n, uint64_t limit) {
    uint32_t nt = (n + kMaxTileCols - 1) / kMaxTileCols;
    if (n == 0) return nt;
    auto fits = [&](uint32_t k) {
        for (uint32_t t = 0; t < k; ++t)
            if (cum[(uint64_t)n * (t + 1) / k] - cum[(uint64_t)n * t / k] >= limit) return false;
        return true;
    };
    while (!fits(nt)) {
        if (nt >= n) return 0;
        nt = std::min<uint32_t>(n, nt + std::max<uint32_t>(1, nt / 4));
    }
    return nt;
}

// Placement of a tile's rows in its entry block.  rptr[0..nrows] are the rows' packed (CSR) starts; with `align` a row
// that would touch more 128-byte lines (16 entries) than its length requires starts at the next 16-entry boundary.
// Writes the packed extents (pack_row_extent) when ext != nullptr; returns the block's entry count, a multiple of 16.
uint64_t layout_tile_rows(const uint32_t* rptr, uint32_t nrows, bool align, uint32_t* ext) {
    uint64_t cur = 0;
    for (uint32_t r = 0; r < nrows; ++r) {
        const uint32_t len = rptr[r + 1] - rptr[r];
        if (len == 0 || len > kMaxTileCols) fail("layer: internal error, tile row length");
        if (align && ((cur & 15) + len + 15) / 16 > ((uint64_t)len + 15) / 16) cur = (cur + 15) & ~15ull;
        if (ext) ext[r] = pack_row_extent((uint32_t)(cur & 0x1FFFFFFu), len);
        cur += len;
    }
    return (cur + 15) & ~15ull;
}

static bool presence_enabled() {   // XRL_PRESENCE=0: no presence words (A/B, tests)
    const char* e = std::getenv("XRL_PRESENCE");
    return !(e && e[0] == '0');
}

std::unique_ptr<Layer> compile_layer(const HostCsc& W_full, const HostCsc& C, float bias, uint32_t only_topk,
                                     const std::string& post_processor, const std::vector<uint32_t>* perm_inv_override,
                                     uint32_t orig_rows, bool structure_only) {
    // structure_only: compile the same tree around an EMPTY weight pattern (no tile rows, no dense matrix, tiny bucket tables)
    HostCsc W_empty;
    if (structure_only) { W_empty.rows = W_full.rows; W_empty.cols = W_full.cols; W_empty.col_ptr.assign((size_t)W_full.cols + 1, 0); }
    const HostCsc& W = structure_only ? W_empty : W_full;
    auto L = std::make_unique<Layer>();
    L->w_rows = W.rows; L->w_cols = W.cols; L->c_rows = C.rows; L->c_cols = C.cols;
    L->bias = bias; L->only_topk = only_topk; L->pp_name = post_processor;
    L->pp = parse_post_processor(post_processor.c_str());
    if (C.rows != W.cols) fail("layer: C.rows (" + std::to_string(C.rows) + ") != W.cols (" + std::to_string(W.cols) + ")");
    const bool has_bias = bias > 0.0f;
    const uint64_t c_nnz = C.nnz();
    if (c_nnz > 0xFFFFFFFFull) fail("layer: too many children");
    for (uint64_t i = 0; i < c_nnz; ++i) if (C.row_idx[i] >= W.cols) fail("layer: C row index out of range");

    // children must be contiguous by parent; otherwise rearrange (inference.hpp:658-668,1855-1872)
    bool contiguous = (c_nnz == C.rows);
    if (contiguous) for (uint64_t i = 0; i < c_nnz; ++i) if (C.row_idx[i] != i) { contiguous = false; break; }
    L->reordered = !contiguous;
    L->n_children = (uint32_t)c_nnz;
    const uint32_t P = C.cols;
    L->h_c_ptr = C.col_ptr; L->h_c_idx.assign(C.row_idx.begin(), C.row_idx.begin() + c_nnz);
    L->h_parent.assign(C.rows, 0xFFFFFFFFu);
    for (uint32_t p = 0; p < P; ++p) for (uint64_t c = C.col_ptr[p]; c < C.col_ptr[p + 1]; ++c) L->h_parent[C.row_idx[c]] = p;

    // tiles
    std::vector<uint32_t> ptile(P + 1, 0), chunk_col(P + 1, 0);
    std::vector<TileDesc> tiles;
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t cb = (uint32_t)C.col_ptr[p], ce = (uint32_t)C.col_ptr[p + 1];
        chunk_col[p] = cb;
        const uint32_t n = ce - cb;
        L->chunk_sizes_desc.push_back(n);
        L->max_chunk_cols = std::max(L->max_chunk_cols, n);
        // column tiles: at most kMaxTileCols children and fewer than max_tile_entries() weights each (K1 packs a
        // tile-relative entry offset into 25 bits); an even split, refined until every tile fits
        uint32_t nt = (n + kMaxTileCols - 1) / kMaxTileCols;
        if (n > 0) {
            std::vector<uint64_t> cum(n + 1, 0);
            for (uint32_t c = 0; c < n; ++c) {
                const uint32_t oc = contiguous ? cb + c : C.row_idx[cb + c];
                if (oc >= W.cols) fail("layer: C row index out of range of W's columns");
                cum[c + 1] = cum[c] + (W.col_ptr[oc + 1] - W.col_ptr[oc]);
            }
            nt = split_chunk(cum.data(), n, max_tile_entries());
            if (nt == 0) fail("layer: one weight column holds " + std::to_string(max_tile_entries()) + " or more entries");
        }
        for (uint32_t t = 0; t < nt; ++t) {
            TileDesc td{};
            const uint32_t b = cb + (uint32_t)((uint64_t)n * t / nt), e = cb + (uint32_t)((uint64_t)n * (t + 1) / nt);
            td.col_begin = b; td.ncols = e - b; td.bias_slot = kNoBias;
            L->max_tile_cols = std::max(L->max_tile_cols, td.ncols);
            tiles.push_back(td);
        }
        ptile[p + 1] = (uint32_t)tiles.size();
        L->max_tiles_per_parent = std::max(L->max_tiles_per_parent, nt);
    }
    chunk_col[P] = (uint32_t)c_nnz;
    std::sort(L->chunk_sizes_desc.begin(), L->chunk_sizes_desc.end(), std::greater<uint32_t>());
    const uint32_t T = (uint32_t)tiles.size();
    L->n_tiles = T;
    L->nwords = (W.rows + 31) / 32;

    auto orig_col = [&](uint32_t c) -> uint32_t { return contiguous ? c : C.row_idx[c]; };

    // entry bases are known from column nnz alone
    std::vector<uint64_t> tile_nnz(T, 0);
    for (uint32_t t = 0; t < T; ++t) {
        uint64_t n = 0;
        for (uint32_t c = tiles[t].col_begin; c < tiles[t].col_begin + tiles[t].ncols; ++c) {
            const uint32_t oc = orig_col(c);
            n += W.col_ptr[oc + 1] - W.col_ptr[oc];
        }
        if (n >= max_tile_entries()) fail("layer: internal error, tile over the entry limit");
        tile_nnz[t] = n;
    }
    uint64_t nnz = 0;
    for (uint32_t t = 0; t < T; ++t) { tiles[t].ent_base = nnz; nnz += tile_nnz[t]; }
    L->nnz = nnz;

    // ---- row lookup structure.  The rank-bitmap costs rows/4 bytes per tile (one load per probe); when that would
    //      take more than a quarter of the device's free HBM (many tiles x many features, e.g. 32768 leaf tiles over
    //      337k features = 2.8 TB) the layer uses a bucket table + binary search over the tile's row ids instead
    //      (O(rows of the tile) memory).  XRL_LOOKUP=bitmap|bucket forces one (tests).
    uint64_t bm_words = (uint64_t)T * L->nwords;
    const uint32_t nwords64 = (W.rows + 63) / 64;
    bool use_bucket = false, use_bm64 = false;
    {
        size_t free_b = 0, total_b = 0;
        const bool have = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
        const char* lk = std::getenv("XRL_LOOKUP");
        if (structure_only) use_bucket = true;
        else if (lk && !std::strcmp(lk, "bucket")) use_bucket = true;
        else if (lk && !std::strcmp(lk, "bitmap")) use_bucket = false;
        else if (lk && !std::strcmp(lk, "bitmap64")) use_bm64 = true;
        else {
            use_bucket = have ? bm_words * 8 > (uint64_t)(free_b / 4) : bm_words * 8 > (48ull << 30);
            // sparse tiles: at most ~4 rows per 64-feature word on average -> most hits are the first row of their word,
            // and the 64-feature word (same bytes per feature) hands back that row's extent with the probe
            if (!use_bucket && T > 0 && nnz > 0) {
                uint64_t rows_ub = 0;   // sum over tiles of distinct rows <= sum of column nnz; exact count comes later, this is a cheap bound
                for (uint32_t t = 0; t < T; ++t) rows_ub += std::min<uint64_t>(tile_nnz[t], W.rows);
                use_bm64 = rows_ub <= 4ull * T * nwords64;
            }
        }
        if (use_bucket) bm_words = 0;
        if (have && bm_words * 8 + nnz * 8 > (uint64_t)(free_b * 0.9))
            fail("layer: the device layout needs " + std::to_string((bm_words * 8 + nnz * 8) >> 20) + " MiB (" + std::to_string(T) +
                 " tiles x " + std::to_string(W.rows) + " features) but only " + std::to_string(free_b >> 20) + " MiB of HBM are free");
        if (W.rows >= (1u << 25)) use_bm64 = false;   // the hit queue packs a row slot into 25 bits in that mode
        if (use_bm64) bm_words = 0;   // the 64-feature words replace the 32-feature ones (same size)
    }
    const bool want_bm32 = !use_bucket && !use_bm64;
    std::vector<Entry> entries(nnz);
    std::vector<BmWord> bitmap(bm_words, BmWord{0, 0});
    std::vector<std::vector<uint32_t>> t_rows(T), t_rptr(T);

    parallel_for(T, [&](size_t t) {
        TileDesc& td = tiles[t];
        std::vector<Nz> nz;
        nz.reserve(tile_nnz[t]);
        for (uint32_t c = td.col_begin; c < td.col_begin + td.ncols; ++c) {
            const uint32_t oc = orig_col(c);
            for (uint64_t e = W.col_ptr[oc]; e < W.col_ptr[oc + 1]; ++e) {
                const uint32_t r = W.row_idx[e];
                if (r >= W.rows) fail("layer: W row index out of range");
                nz.push_back(Nz{r, c - td.col_begin, W.val[e]});
            }
        }
        // rows ascending; inside a row the column order (ascending) is kept: stable
        std::stable_sort(nz.begin(), nz.end(), [](const Nz& a, const Nz& b) { return a.row < b.row; });
        auto& rows = t_rows[t]; auto& rptr = t_rptr[t];
        BmWord* bm = want_bm32 ? bitmap.data() + t * (uint64_t)L->nwords : nullptr;
        Entry* ent = entries.data() + td.ent_base;
        for (size_t i = 0; i < nz.size(); ++i) {
            if (i == 0 || nz[i].row != nz[i - 1].row) {
                rows.push_back(nz[i].row);
                rptr.push_back((uint32_t)i);
                if (bm) bm[nz[i].row >> 5].bits |= 1u << (nz[i].row & 31);
            }
            ent[i] = Entry{nz[i].col, nz[i].val};
        }
        rptr.push_back((uint32_t)nz.size());
        td.nrows = (uint32_t)rows.size();
        uint32_t run = 0;
        if (bm) for (uint32_t w = 0; w < L->nwords; ++w) { bm[w].rank = run; run += (uint32_t)__builtin_popcount(bm[w].bits); }
        // check_bias_explicit, inference.hpp:500-502: last row of the chunk is W's last row
        td.bias_slot = (has_bias && td.nrows > 0 && rows.back() == W.rows - 1) ? td.nrows - 1 : kNoBias;
    });

    uint64_t total_rows = 0;
    for (uint32_t t = 0; t < T; ++t) { tiles[t].rowptr_base = total_rows; total_rows += tiles[t].nrows; }
    L->total_rows = total_rows;
    std::vector<uint32_t> row_idx(total_rows);
    parallel_for(T, [&](size_t t) {
        if (!t_rows[t].empty()) std::memcpy(row_idx.data() + tiles[t].rowptr_base, t_rows[t].data(), t_rows[t].size() * 4);
    });

    // ---- bucket lookup (see above): per tile, first row slot of every feature-id range of 2^bk_shift ids
    std::vector<uint32_t> bucket;
    if (use_bucket) {
        uint32_t max_rows = 1;
        for (uint32_t t = 0; t < T; ++t) max_rows = std::max(max_rows, tiles[t].nrows);
        uint32_t want = 16;                                        // ~4 rows per bucket on the fullest tile
        while (want < 4096 && want * 4 < max_rows) want <<= 1;
        uint32_t shift = 0;
        while ((((uint64_t)W.rows - 1) >> shift) + 1 > want) ++shift;
        const uint32_t NBK = W.rows ? (uint32_t)((((uint64_t)W.rows - 1) >> shift) + 1) : 1;
        bucket.assign((size_t)T * (NBK + 1) + 1, 0u);             // + one readable element past the end
        std::vector<uint32_t> tile_maxlen(T, 0);
        parallel_for(T, [&](size_t t) {
            const std::vector<uint32_t>& rows = t_rows[t];
            uint32_t* bk = bucket.data() + t * (size_t)(NBK + 1);
            const uint32_t R = (uint32_t)rows.size();
            uint32_t r0 = 0, mx = 0;
            for (uint32_t k = 0; k < NBK; ++k) {
                while (r0 < R && (rows[r0] >> shift) < k) ++r0;
                bk[k] = r0;
                if (k > 0) mx = std::max(mx, bk[k] - bk[k - 1]);
            }
            bk[NBK] = R;
            tile_maxlen[t] = std::max(mx, R - bk[NBK - 1]);
        });
        uint32_t maxlen = 1;
        for (uint32_t t = 0; t < T; ++t) maxlen = std::max(maxlen, tile_maxlen[t]);
        uint32_t levels = 0;
        while ((1u << levels) < maxlen) ++levels;                  // steps of 2^(levels-1) .. 1 cover the longest bucket
        L->bk_shift = shift; L->bk_n = NBK; L->bk_levels = levels;
    }

    // algorithmic bytes of the REFERENCE chunk layout per parent:
    // 8*E_p (entries) + 4*R_p (row_idx) + 4*(R_p+1) (row_ptr as u32)
    std::vector<float> chunk_alg(P, 0.f);
    for (uint32_t p = 0; p < P; ++p) {
        uint64_t E = 0, R = 0;
        const uint32_t t0 = ptile[p], t1 = ptile[p + 1];
        if (t1 - t0 == 1) { E = tile_nnz[t0]; R = tiles[t0].nrows; }
        else if (t1 > t0) {
            std::vector<uint32_t> u;
            for (uint32_t t = t0; t < t1; ++t) { E += tile_nnz[t]; u.insert(u.end(), t_rows[t].begin(), t_rows[t].end()); }
            std::sort(u.begin(), u.end());
            R = std::unique(u.begin(), u.end()) - u.begin();
        }
        chunk_alg[p] = (float)(8.0 * E + 4.0 * R + (R ? 4.0 * (R + 1) : 0.0));
    }

    // bias contribution of every child column: the reference adds fl32(bias * w) to the column's
    // accumulator (inference.hpp:806-811 / :824-830); columns without an explicit bias entry add
    // nothing, and acc + (+0.0f) == acc for every reachable acc, so a dense vector is equivalent.
    {   // largest weight magnitude (the bound-pruning guard, xrl_predict.cpp): any inf / NaN makes it +inf
        float mx = 0.0f; bool fin = std::isfinite(bias);
        const uint64_t wn = W_full.col_ptr[W_full.cols];
        for (uint64_t e = 0; e < wn; ++e) { const float a = std::fabs(W_full.val[e]); if (!(a <= 3.0e38f)) { fin = false; break; } mx = std::max(mx, a); }
        L->w_absmax = fin ? mx * std::max(1.0f, std::fabs(bias)) : INFINITY;
        if (!(L->w_absmax <= 3.0e38f)) L->w_absmax = INFINITY;
    }
    std::vector<float> bias_prod(c_nnz, 0.0f);
    if (has_bias) {
        for (uint32_t c = 0; c < (uint32_t)c_nnz; ++c) {
            const uint32_t oc = orig_col(c);
            const uint64_t cb = W.col_ptr[oc], ce = W.col_ptr[oc + 1];
            for (uint64_t e = cb; e < ce; ++e)
                if (W.row_idx[e] == W.rows - 1) { volatile float pr = bias * W.val[e]; bias_prod[c] = 0.0f + pr; }
        }
    }
    // ---- device layout of rows: {start, length} per row, and the entries re-laid so that NO ROW TOUCHES MORE
    //      128-BYTE LINES THAN ITS LENGTH REQUIRES (a row that would straddle an extra line starts at the next
    //      16-entry boundary).  K1 is bound by the number of cache lines its 8-byte gathers request from the L2;
    //      unaligned, a 27-entry row costs 2.7 lines on average instead of 2.  XRL_ROW_ALIGN=0 keeps rows packed.
    std::vector<uint32_t> row_ext(total_rows + 2, 0u);
    {
        const char* ra = std::getenv("XRL_ROW_ALIGN");
        bool align = !(ra && ra[0] == '0');
        auto lay_out = [&](size_t t, uint32_t* ext) -> uint64_t { return layout_tile_rows(t_rptr[t].data(), tiles[t].nrows, align, ext); };
        std::vector<uint64_t> dev_base((size_t)T + 1, 0);
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<uint64_t> padded(T, 0);
            parallel_for(T, [&](size_t t) { padded[t] = lay_out(t, nullptr); });
            bool fits = true;
            for (uint32_t t = 0; t < T; ++t) { dev_base[t + 1] = dev_base[t] + padded[t]; fits = fits && padded[t] < (1ull << 25); }
            if (fits || !align) break;
            align = false;                                         // a tile would leave the 25-bit offset range: keep this layer packed
        }
        std::vector<Entry> dev_entries(dev_base[T] + 64, Entry{0u, 0.0f});   // + readable elements past the end (unconditional loads)
        parallel_for(T, [&](size_t t) {
            uint32_t* ext = row_ext.data() + tiles[t].rowptr_base;
            lay_out(t, ext);
            const Entry* src = entries.data() + tiles[t].ent_base;
            Entry* dst = dev_entries.data() + dev_base[t];
            const uint32_t* trp = t_rptr[t].data();
            for (uint32_t r = 0; r < tiles[t].nrows; ++r) std::memcpy(dst + (ext[r] & 0x1FFFFFFu), src + trp[r], (size_t)((ext[r] >> 25) + 1u) * sizeof(Entry));
        });
        for (uint32_t t = 0; t < T; ++t) tiles[t].ent_base = dev_base[t];
        entries.swap(dev_entries);
    }
    std::vector<BmWord64> bitmap64;
    if (use_bm64) {
        bitmap64.assign((size_t)T * nwords64 + 1, BmWord64{0u, 0u, 0u, 0u});
        parallel_for(T, [&](size_t t) {
            BmWord64* bw = bitmap64.data() + t * (size_t)nwords64;
            const std::vector<uint32_t>& rows = t_rows[t];
            const uint32_t* ext = row_ext.data() + tiles[t].rowptr_base;
            for (uint32_t r = 0; r < (uint32_t)rows.size(); ++r) {
                BmWord64& w = bw[rows[r] >> 6];
                if ((w.lo | w.hi) == 0u) {                                         // first row of the word (rows ascend)
                    w.rank = r;
                    w.ext0 = (ext[r] >> 25) == 0x7Fu ? (0xFE000000u | r) : ext[r];   // a 128-entry row reads like the "slot" marker: send it through the table
                }
                const uint32_t b = rows[r] & 63u;
                if (b < 32) w.lo |= 1u << b; else w.hi |= 1u << (b - 32);
            }
            uint32_t run = 0;                                                   // empty words still need a rank (hits never read it)
            for (uint32_t k = 0; k < nwords64; ++k) { if ((bw[k].lo | bw[k].hi) == 0u) bw[k].rank = run; run = bw[k].rank + (uint32_t)__builtin_popcount(bw[k].lo) + (uint32_t)__builtin_popcount(bw[k].hi); }
        });
    }
    // ---- DENSE row format (K1Q, xrl_k1q.hip) for layers whose padded dense matrix fits the HBM budget: chunks of
    //      <= 64 children are one dense tile (padded to a power of two), wider chunks are cut evenly into tiles of
    //      <= 32.  Built ON the device from the CSC columns (memset to kMissing + scatter); the tile format above is
    //      kept too (it serves beams / top-k sizes K1Q cannot hold in registers).  XRL_DENSE=0 disables it,
    //      XRL_DENSE_MAX_MB caps one layer's matrix (default 64 GiB, and never more than a quarter of the free HBM).
    std::vector<uint32_t> d_ptile, d_tcol;
    uint32_t d_gp_log2 = 0, d_max_tiles = 0, pres_words = 0; uint64_t d_ld = 0;
    bool d_full = false;
    {
        const char* de = std::getenv("XRL_DENSE");
        bool want = !(de && de[0] == '0') && c_nnz > 0 && W.rows > 0 && !structure_only;
        // a column with duplicate or unsorted row ids cannot be scattered into one cell per (feature, column): such a layer stays in the
        // tile format.  (A weight whose bits equal the "no entry" marker -- an explicit -0.0 -- is stored as +0.0 by densify_kernel.)
        for (uint32_t c = 0; want && c < W.cols; ++c)
            for (uint64_t e = W.col_ptr[c] + 1; e < W.col_ptr[c + 1]; ++e)
                if (W.row_idx[e] <= W.row_idx[e - 1]) { want = false; break; }
        if (want) {
            const uint32_t wide = L->max_chunk_cols;
            uint32_t gp = 1;
            if (wide <= 64) { while (gp < wide) gp <<= 1; } else gp = 32;
            d_ptile.assign(P + 1, 0);
            for (uint32_t p = 0; p < P; ++p) {
                const uint32_t cb = chunk_col[p], n = chunk_col[p + 1] - cb;
                const uint32_t nt = n == 0 ? 0u : (wide <= 64 ? 1u : (n + 31u) / 32u);
                for (uint32_t t = 0; t < nt; ++t) d_tcol.push_back(cb + (uint32_t)((uint64_t)n * t / nt));
                d_ptile[p + 1] = (uint32_t)d_tcol.size();
                d_max_tiles = std::max(d_max_tiles, nt);
            }
            const uint64_t n_dt = d_tcol.size();
            d_tcol.push_back((uint32_t)c_nnz);
            d_tcol.push_back((uint32_t)c_nnz);                     // one readable element past the end
            while ((1u << d_gp_log2) < gp) ++d_gp_log2;
            d_ld = (n_dt * gp + 31) & ~31ull;
            const uint64_t bytes = ((uint64_t)W.rows + 1) * d_ld * 4;        // + one all-kMissing row (features outside the layer)
            uint64_t cap_b = 64ull << 30;
            if (const char* mb = std::getenv("XRL_DENSE_MAX_MB")) cap_b = std::strtoull(mb, nullptr, 10) << 20;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_b = std::min<uint64_t>(cap_b, free_b / 4);
            if (n_dt == 0 || d_ld >= (1ull << 30) || bytes > cap_b) want = false;
        }
        if (want) {
            std::vector<uint32_t> src_col(c_nnz), dst_off(c_nnz);
            for (uint64_t dt = 0; dt + 2 < d_tcol.size(); ++dt)
                for (uint32_t c = d_tcol[dt]; c < d_tcol[dt + 1]; ++c) { src_col[c] = orig_col(c); dst_off[c] = (uint32_t)(dt << d_gp_log2) + (c - d_tcol[dt]); }
            DevBuf t_ptr, t_idx, t_val, t_src, t_dst;
            t_ptr.upload(W.col_ptr); t_idx.upload(W.row_idx); t_val.upload(W.val); t_src.upload(src_col); t_dst.upload(dst_off);
            L->d_wd.reserve(((size_t)W.rows + 1) * d_ld * 4);
            launch_densify(t_ptr.as<uint64_t>(), t_idx.as<uint32_t>(), t_val.as<float>(), t_src.as<uint32_t>(), t_dst.as<uint32_t>(),
                           (uint32_t)c_nnz, W.rows, d_ld, L->d_wd.as<uint32_t>(), nullptr);
            // presence words (LayerDev::pres): layers of many narrow dense tiles only (XRL_PRESENCE=0: none)
            const uint64_t n_dtiles = d_tcol.size() - 2;
            if (n_dtiles >= 16 && d_gp_log2 >= 1 && d_gp_log2 <= 5 && presence_enabled()) {
                pres_words = 1; while ((uint64_t)pres_words * 32 < n_dtiles) pres_words <<= 1;
                if (((uint64_t)W.rows + 1) * pres_words * 4 >= (1ull << 31)) pres_words = 0;   // K1Q addresses the whole array through ONE buffer resource
            }
            if (pres_words) {
                L->d_pres.reserve(((size_t)W.rows + 1) * pres_words * 4);
                launch_presence(L->d_wd.as<uint32_t>(), d_ld, W.rows + 1, d_gp_log2, (uint32_t)n_dtiles, pres_words, L->d_pres.as<uint32_t>(), nullptr);
            }
            XRL_HIP(hipStreamSynchronize(nullptr));
            L->d_dptile.upload(d_ptile); L->d_dtcol.upload(d_tcol);
            L->dense_bytes = L->d_wd.cap;
            // does every kept child hold a weight for every feature row (dense-input models do)?
            const uint32_t n_feat = has_bias ? W.rows - 1 : W.rows;
            d_full = true;
            for (uint32_t c = 0; d_full && c < (uint32_t)c_nnz; ++c) {
                const uint32_t oc = src_col[c];
                uint64_t n = W.col_ptr[oc + 1] - W.col_ptr[oc];
                if (has_bias && n > 0 && W.row_idx[W.col_ptr[oc + 1] - 1] == W.rows - 1) --n;
                d_full = n == n_feat;
            }
            std::vector<uint32_t> tile_parent(T, 0);
            for (uint32_t p = 0; p < P; ++p) for (uint32_t t = ptile[p]; t < ptile[p + 1]; ++t) tile_parent[t] = p;
            L->d_tile_parent.upload(tile_parent);
        } else {
            d_ptile.clear(); d_tcol.clear();
        }
    }
    row_idx.push_back(0u);
    L->d_bias_prod.upload(bias_prod);
    L->d_tiles.upload(tiles); L->d_ptile.upload(ptile); L->d_chunk_col.upload(chunk_col);
    if (want_bm32) L->d_bitmap.upload(bitmap);
    if (use_bucket) L->d_bucket.upload(bucket);
    if (use_bm64) L->d_bitmap64.upload(bitmap64);
    L->d_row_ptr.upload(row_ext); L->d_row_idx.upload(row_idx);
    L->d_entries.upload(entries); L->d_chunk_alg.upload(chunk_alg);
    if (!contiguous) {
        std::vector<uint32_t> perm_inv(C.row_idx.begin(), C.row_idx.begin() + c_nnz);
        L->d_perm_inv.upload(perm_inv);
    } else if (perm_inv_override) {
        if (perm_inv_override->size() != c_nnz) fail("layer: perm_inv size does not match C");
        L->d_perm_inv.upload(*perm_inv_override);
        L->reordered = true;
        L->c_rows = orig_rows;          // predictions carry ORIGINAL ids; result CSR has perm.size() columns
        // host maps in original ids (predict_on_selected_outputs is CSC-only in the reference; mmap models
        // have no CSC copy, so K4 is unavailable for them, but the maps stay consistent)
        L->h_parent.assign(orig_rows, 0xFFFFFFFFu);
        for (uint32_t p = 0; p < P; ++p) for (uint64_t c = C.col_ptr[p]; c < C.col_ptr[p + 1]; ++c) L->h_parent[(*perm_inv_override)[C.row_idx[c]]] = p;
        for (auto& v : L->h_c_idx) v = (*perm_inv_override)[v];
    }
    L->device_bytes = L->d_tiles.cap + L->d_ptile.cap + L->d_chunk_col.cap + L->d_bitmap.cap + L->d_row_ptr.cap +
                      L->d_row_idx.cap + L->d_entries.cap + L->d_perm_inv.cap + L->d_chunk_alg.cap + L->d_bias_prod.cap +
                      L->d_bucket.cap + L->d_bitmap64.cap + L->d_wd.cap + L->d_dptile.cap + L->d_dtcol.cap + L->d_pres.cap;

    LayerDev& d = L->dev;
    d.tiles = L->d_tiles.as<TileDesc>(); d.ptile = L->d_ptile.as<uint32_t>(); d.chunk_col = L->d_chunk_col.as<uint32_t>();
    d.bitmap = want_bm32 ? L->d_bitmap.as<BmWord>() : nullptr;
    d.bitmap64 = use_bm64 ? L->d_bitmap64.as<BmWord64>() : nullptr; d.nwords64 = nwords64;
    d.bucket = use_bucket ? L->d_bucket.as<uint32_t>() : nullptr; d.bk_shift = L->bk_shift; d.bk_n = L->bk_n; d.bk_levels = L->bk_levels;
    d.row_ext = L->d_row_ptr.as<uint32_t>(); d.row_idx = L->d_row_idx.as<uint32_t>();
    d.entries = L->d_entries.as<Entry>(); d.perm_inv = (contiguous && !perm_inv_override) ? nullptr : L->d_perm_inv.as<uint32_t>();
    d.chunk_alg_bytes = L->d_chunk_alg.as<float>();
    d.bias_prod = L->d_bias_prod.as<float>();
    d.n_parents = P; d.n_children = L->n_children; d.n_tiles = T; d.nwords = L->nwords; d.w_rows = W.rows;
    d.max_tiles_per_parent = L->max_tiles_per_parent; d.max_tile_cols = L->max_tile_cols;
    d.bias = bias; d.has_bias = has_bias ? 1 : 0;
    d.wd = L->dense_bytes ? L->d_wd.as<uint32_t>() : nullptr; d.d_ld = d_ld; d.d_gp_log2 = d_gp_log2; d.d_max_tiles = d_max_tiles;
    d.d_ptile = L->dense_bytes ? L->d_dptile.as<uint32_t>() : nullptr; d.d_tcol = L->dense_bytes ? L->d_dtcol.as<uint32_t>() : nullptr;
    {
        const uint64_t wp = (uint64_t)d_max_tiles << d_gp_log2;
        const double per_segment = (P && W.rows) ? (double)L->nnz / ((double)W.rows * (double)P) : 0.0;   // weights per (feature, parent)
        d.d_sparse_ok = (wp <= 32 || per_segment >= 1.0) ? 1 : 0;
    }
    d.pres = pres_words ? L->d_pres.as<uint32_t>() : nullptr; d.pres_words = pres_words;
    d.d_regular = 0;
    if (L->dense_bytes && d_max_tiles == 1 && d_ptile.size() == (size_t)P + 1 && d_tcol.size() >= (size_t)P + 1) {
        bool reg = true;
        for (uint32_t p2 = 0; reg && p2 <= P; ++p2) reg = d_ptile[p2] == p2 && (p2 == P || d_tcol[p2] == (p2 << d_gp_log2));
        reg = reg && (uint64_t)c_nnz == ((uint64_t)P << d_gp_log2);
        d.d_regular = reg ? 1 : 0;
    }
    d.d_full = d_full ? 1 : 0; d.tile_parent = L->dense_bytes ? L->d_tile_parent.as<uint32_t>() : nullptr;
    // ---- TILE ROWS held densely (K1T, xrl_k1t.hip): built on the device from the tile format just uploaded.  Needs one cell per (row, column)
    //      (no duplicate row ids inside a weight column), a rank-bitmap lookup (the slots it returns index the rows) and room:
    //      (rows + tiles) x stride x 4 bytes, at most a quarter of the free HBM / XRL_TILE_ROWS_MAX_MB (default 64 GiB; Amazon-670K's leaf -- 5 094 rows per
    //      82-column tile, most of them single-entry -- takes 16 GB: 22 GB of model instead of 6, +0.2 s of load).  XRL_TILE_ROWS=0 disables it.
    d.wt = nullptr; d.wt_base = nullptr; d.wt_stride = 0; d.wt_bytes = 0;
    {
        const char* te = std::getenv("XRL_TILE_ROWS");
        bool want = !(te && te[0] == '0') && !structure_only && !use_bucket && T > 0 && nnz > 0 && L->max_tile_cols <= kMaxTileCols;
        for (uint32_t c = 0; want && c < W.cols; ++c)
            for (uint64_t e = W.col_ptr[c] + 1; e < W.col_ptr[c + 1]; ++e)
                if (W.row_idx[e] <= W.row_idx[e - 1]) { want = false; break; }
        int g = 0, nr = 0;
        k1t_shape(L->max_tile_cols, g, nr);
        const uint64_t stride = (uint64_t)g * nr;
        const uint64_t floats = (total_rows + T) * stride;
        if (want) {
            uint64_t cap_b = 64ull << 30;
            if (const char* mb = std::getenv("XRL_TILE_ROWS_MAX_MB")) cap_b = std::strtoull(mb, nullptr, 10) << 20;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_b = std::min<uint64_t>(cap_b, free_b / 4);
            uint32_t max_rows = 0;
            for (uint32_t t = 0; t < T; ++t) max_rows = std::max(max_rows, tiles[t].nrows);
            if (nr > 4 || floats * 4 > cap_b || ((uint64_t)max_rows + 1) * stride * 4 >= (1ull << 32)) want = false;   // (a row's byte offset inside its tile is 32 bits)
        }
        if (want) {
            std::vector<uint64_t> base(T);
            for (uint32_t t = 0; t < T; ++t) base[t] = (tiles[t].rowptr_base + t) * stride;
            L->d_wt_base.upload(base);
            L->d_wt.reserve(floats * 4 + 64);
            d.wt_base = L->d_wt_base.as<uint64_t>(); d.wt_stride = (uint32_t)stride;
            launch_tile_rows(d, floats + 16, L->d_wt.as<uint32_t>(), nullptr);
            XRL_HIP(hipStreamSynchronize(nullptr));
            d.wt = L->d_wt.as<float>(); d.wt_bytes = floats * 4 + 64;
            L->device_bytes += L->d_wt.cap + L->d_wt_base.cap;
        }
    }
    return L;
}

void ensure_device_csc(Layer& L) {
    if (L.csc_ready) return;
    HostCsc tmp;
    const HostCsc* W = L.w_host.get();
    if (!W) {
        if (L.w_path.empty()) fail("predict_on_selected_outputs: the layer's CSC weights are not available");
        load_csc_npz(L.w_path, tmp);
        W = &tmp;
    }
    L.d_csc_ptr.upload(W->col_ptr); L.d_csc_idx.upload(W->row_idx); L.d_csc_val.upload(W->val);
    L.device_bytes += L.d_csc_ptr.cap + L.d_csc_idx.cap + L.d_csc_val.cap;
    L.csc_ready = true;
}

void finalize_model(Model& m) {
    if (m.layers.empty()) fail("model has no layers");
    const Layer& last = *m.layers.back();
    // MLModel::{label,code,feature}_count, inference.hpp:2241-2255 (chunked W.cols = #children kept)
    m.nr_labels = last.reordered ? last.n_children : last.w_cols;
    m.nr_codes = last.c_cols;
    m.nr_features = last.bias > 0.0f ? last.w_rows - 1 : last.w_rows;
    for (size_t l = 1; l < m.layers.size(); ++l) {
        const uint32_t prev_out = m.layers[l - 1]->reordered ? m.layers[l - 1]->c_rows : m.layers[l - 1]->w_cols;
        if (m.layers[l]->c_cols != prev_out)
            fail("layer " + std::to_string(l) + ": C.cols (" + std::to_string(m.layers[l]->c_cols) +
                 ") does not match the previous layer's label count (" + std::to_string(prev_out) + ")");
    }
    if (!m.stream) XRL_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    // levels 0 and 1 in one 64-column dense matrix (LayerDev::wd01): K1Q's fused walk of the two levels then issues ONE load per feature
    if (m.layers.size() >= 2) {
        Layer& L0 = *m.layers[0]; const Layer& L1 = *m.layers[1];
        const LayerDev& d0 = L0.dev; const LayerDev& d1 = L1.dev;
        const char* e = std::getenv("XRL_K1Q_MERGE01");
        if (!(e && e[0] == '0') && d0.wd && d1.wd && d0.n_parents == 1 && d0.d_max_tiles == 1 && d0.w_rows == d1.w_rows && d1.n_parents == d0.n_children) {
            const uint64_t K0 = d0.n_children, c1 = (K0 * d1.d_max_tiles) << d1.d_gp_log2;
            if (K0 >= 1 && c1 + K0 <= 64 && c1 <= d1.d_ld && K0 <= d0.d_ld) {
                m.d_wd01.reserve(((size_t)d0.w_rows + 1) * 64 * 4);
                launch_merge01(d0.wd, d0.d_ld, (uint32_t)K0, d1.wd, d1.d_ld, (uint32_t)c1, d0.w_rows + 1, m.d_wd01.as<uint32_t>(), nullptr);
                XRL_HIP(hipStreamSynchronize(nullptr));
                L0.dev.wd01 = m.d_wd01.as<uint32_t>(); L0.dev.wd01_c1 = (uint32_t)c1;
                L0.device_bytes += m.d_wd01.cap;
            }
        }
    }
}

std::unique_ptr<Model> load_model_from_disk(const std::string& path, int weight_matrix_type) {
    const JsonValue meta = parse_json_file(path + "/param.json");   // HierarchicalMLModelMetadata, inference.hpp:52-99
    const JsonValue* depth_v = meta.get("depth");
    if (!depth_v || depth_v->type != JsonValue::NUMBER) fail(path + "/param.json: missing \"depth\"");
    if (const JsonValue* mm = meta.get("is_mmap"))
        if (mm->type == JsonValue::BOOL && mm->b) fail("This folder contains mmap model. Cannot load in npz format.");
    const int depth = (int)depth_v->num;
    auto m = std::make_unique<Model>();
    XRL_HIP(hipGetDevice(&m->device));
    m->weight_matrix_type = weight_matrix_type;
    m->csc_route = weight_matrix_type == 0;
    for (int d = 0; d < depth; ++d) {
        const std::string lp = path + "/" + std::to_string(d) + ".model";
        const JsonValue p = parse_json_file(lp + "/param.json");     // MLModelMetadata, inference.hpp:101-157
        const JsonValue* bias = p.get("bias");
        const JsonValue* kw = p.get("pred_kwargs");
        if (!bias || bias->type != JsonValue::NUMBER || !kw) fail(lp + "/param.json: missing bias / pred_kwargs");
        const JsonValue* topk = kw->get("only_topk");
        const JsonValue* pp = kw->get("post_processor");
        if (!topk || !pp || pp->type != JsonValue::STRING) fail(lp + "/param.json: missing pred_kwargs.only_topk / post_processor");
        HostCsc W, C;
        load_csc_npz(lp + "/W.npz", W);
        if (d == 0 && !file_exists(lp + "/C.npz")) {   // inference.hpp:1580-1583: root without codes
            C.rows = W.cols; C.cols = 1;
            C.col_ptr = {0, W.cols};
            C.row_idx.resize(W.cols); C.val.assign(W.cols, 1.0f);
            for (uint32_t i = 0; i < W.cols; ++i) C.row_idx[i] = i;
        } else {
            load_csc_npz(lp + "/C.npz", C);
        }
        // weight_matrix_type CSC (pecos/core/base.py:49): the reference then runs w_ops<csc_t> (inference.hpp:1081-1149) --
        // bias first, dot product summed separately -- which differs from the chunked arithmetic in the last bits
        const bool csc = weight_matrix_type == 0;
        m->layers.push_back(compile_layer(W, C, (float)bias->num, (uint32_t)topk->num, pp->str, nullptr, 0, csc));
        m->layers.back()->w_path = lp + "/W.npz";
        if (csc) m->layers.back()->w_host = std::make_shared<HostCsc>(std::move(W));
    }
    finalize_model(*m);
    return m;
}

}  // namespace xrl
