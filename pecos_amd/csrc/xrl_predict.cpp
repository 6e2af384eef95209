// Host orchestrator: per-layer K0 -> K1 -> K2 launches on one HIP stream, ping-pong beam
// buffers, no host synchronisation inside the layer loop (grid sizes depend only on shapes).
//
// Reference driver being restated: HierarchicalMLModel::predict (inference.hpp:2446-2488) calling
// MLModel::predict_internal (:2029-2080) per layer.
#include "xrl_predict.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace xrl {

namespace {
size_t profile_slot(Model& m, const char* name, uint32_t layer) {
    for (size_t i = 0; i < m.profile.size(); ++i)
        if (m.profile[i].layer == layer && m.profile[i].name == name) return i;
    m.profile.push_back(ProfileSlot{name, layer});
    return m.profile.size() - 1;
}
}  // namespace

// dense-X SGEMM layers, bound pruning: beam slots whose children fill about one candidate register (64) are scored first -- never all of
// them (the second stage must keep at least one slot)
static uint32_t k1g_first_slots(const Layer& L, uint32_t beam_in, int forced = 0) {
    if (forced > 0) return (uint32_t)std::max<int64_t>(1, std::min<int64_t>(beam_in > 1 ? (int64_t)beam_in - 1 : 1, forced));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(beam_in > 1 ? beam_in - 1 : 1, 64 / std::max<uint64_t>(1, L.cand_bound(1))));
}

uint32_t effective_topk(const Model& m, uint32_t only_topk) {
    return only_topk ? only_topk : m.layers.back()->only_topk;   // inference.hpp:2055
}

void predict_device(Model& m, const QueriesDev& X, const PredictOpts& o, uint32_t* d_out_idx, float* d_out_val,
                    uint32_t* d_out_cnt, uint32_t out_stride, hipStream_t stream, bool sync, uint32_t row_begin, uint32_t row_count) {
    const size_t T = m.layers.size();
    // rows [row_begin, row_end) of X (results land at the same rows of the output buffers); default: all rows
    if (row_begin > X.rows) fail("predict: row range outside X");
    const uint64_t row_end = std::min<uint64_t>(X.rows, (uint64_t)row_begin + row_count);
    const uint64_t n_rows = row_end - row_begin;
    if (!m.ws) m.ws = std::make_unique<Workspace>();
    Workspace& ws = *m.ws;
    if (!stream) stream = m.stream;
    // the scratch buffers are shared by every predict of the handle: an asynchronous predict still running on another stream must finish first
    if (m.ws_done && m.ws_stream != stream) XRL_HIP(hipStreamWaitEvent(stream, m.ws_done, 0));
    static const bool step_marker = [] { const char* e = std::getenv("XRL_STEP_MARKER"); return e && e[0] == '1'; }();
    if (step_marker && !o.stats_out) launch_step_marker(stream);

    // MLModel::predict_internal's shape checks live in Python for the reference
    // (xmc/base.py:1603-1607); here a mismatch is a loud error instead of UB.
    if (!X.dense && X.cols != m.nr_features && X.cols != m.layers[0]->w_rows)
        fail("X.shape[1] (" + std::to_string(X.cols) + ") != nr_features (" + std::to_string(m.nr_features) + ")");
    if (X.dense && X.cols < m.nr_features)
        fail("dense X has fewer columns (" + std::to_string(X.cols) + ") than nr_features (" + std::to_string(m.nr_features) + ")");

    // ---- resolve per-layer k / post-processor (inference.hpp:2471, 2055-2058)
    std::vector<uint32_t> k(T), beam_in(T), cstride(T);
    std::vector<PostProc> pp(T);
    const bool has_init = o.initial != nullptr;
    const bool csc = o.csc_route || m.csc_route;
    for (size_t l = 0; l < T; ++l) {
        const Layer& L = *m.layers[l];
        const uint32_t ov = (l == T - 1) ? o.only_topk : o.beam_size;
        k[l] = ov ? ov : L.only_topk;
        if (k[l] == 0) fail("layer " + std::to_string(l) + ": only_topk resolved to 0");
        pp[l] = o.post_processor ? parse_post_processor(o.post_processor) : L.pp;
        uint64_t bin = (l == 0) ? (has_init ? std::max<uint32_t>(1, o.initial_max) : 1)
                                : std::min<uint64_t>(k[l - 1], cstride[l - 1]);
        if (!(l == 0 && has_init)) bin = std::min<uint64_t>(bin, L.c_cols ? L.c_cols : 1);   // explicit codes may list a parent more than once
        beam_in[l] = (uint32_t)std::max<uint64_t>(1, bin);
        const uint64_t cb = std::max<uint64_t>(1, (l == 0 && has_init && o.initial_cand_bound) ? o.initial_cand_bound : L.cand_bound(beam_in[l]));
        if (cb > 0x7FFFFFFFull) fail("candidate row too long; lower beam_size");
        cstride[l] = (uint32_t)cb;
    }
    const uint32_t k_last = k[T - 1];
    if (k_last > out_stride) fail("out_stride smaller than the effective only_topk");
    uint32_t beam_stride = 1;
    for (size_t l = 0; l + 1 < T; ++l) beam_stride = std::max(beam_stride, k[l]);
    uint32_t bin_max = 1, cs_max = 1;
    for (size_t l = 0; l < T; ++l) { bin_max = std::max(bin_max, beam_in[l]); cs_max = std::max(cs_max, cstride[l]); }

    // ---- batch rows so that the candidate buffer stays bounded (default 6 GiB of 288)
    const uint64_t cand_budget = 6ull << 30;
    uint64_t nb = std::max<uint64_t>(1, cand_budget / ((uint64_t)cs_max * 4));
    nb = std::min<uint64_t>(nb, 1u << 22);
    if (m.max_batch_rows > 0) nb = std::min<uint64_t>(nb, (uint64_t)m.max_batch_rows);
    nb = std::min<uint64_t>(nb, std::max<uint64_t>(1, std::max<uint64_t>(n_rows, o.reserve_rows)));   // reserve_rows: size the scratch for the caller's largest batch up front
    // two lanes: the row batches alternate between the caller's stream and an auxiliary one.  K1 launches are chained
    // across the lanes (one K1 at a time owns the memory system); a lane's K0 / sort / K2 run under the other lane's K1.
    const int lanes = (m.overlap_min_rows > 0 && !o.stats_out && !m.profiling && n_rows >= (uint64_t)m.overlap_min_rows) ? 2 : 1;
    if (lanes == 2) nb = std::min<uint64_t>(nb, (n_rows + 1) / 2);

    for (int ln = 0; ln < lanes; ++ln) {
        LaneWs& lw = ws.lane[ln];
        for (int i = 0; i < 2; ++i) {
            lw.beam_idx[i].reserve(nb * beam_stride * 4);
            lw.beam_val[i].reserve(nb * beam_stride * 4);
            lw.beam_cnt[i].reserve(nb * 4);
        }
    }
    uint64_t slots_max = 1;
    for (size_t l = 0; l < T; ++l) slots_max = std::max<uint64_t>(slots_max, nb * beam_in[l] * m.layers[l]->max_tiles_per_parent);
    for (int ln = 0; ln < lanes; ++ln) ws.lane[ln].items.reserve(slots_max * k0_item_bytes());
    // per layer: 0 = K1 on the items in natural order, 1 = K1 on tile-sorted items, 3 = K1G (dense X, tiled SGEMM over tile-sorted items)
    auto layer_mode = [&](size_t l, uint64_t rows) -> int {
        const Layer& L = *m.layers[l];
        if (L.n_tiles > sort_max_tiles()) return 0;
        const uint64_t slots = rows * beam_in[l] * L.max_tiles_per_parent;
        // 3 = K1G: dense queries against a dense-format layer as a tiled SGEMM over tile-sorted items
        if (X.dense && m.dense_layers && m.k1g_min_items > 0 && !csc && k1g_cols(L.dev) != 0 && k[l] <= k2_max_k() &&
            slots / std::max<uint32_t>(1, L.n_tiles) >= (uint64_t)m.k1g_min_items) return 3;
        if (m.sort_min_tiles > 0 && L.n_tiles >= (uint32_t)m.sort_min_tiles) return 1;
        return 0;
    };
    // the second phase of a bound-pruned tile-format layer runs on TILE-SORTED items (option sort_rest): what is left after the first phase
    // is, on a model that does not let the bound stop much, most of the layer's work, and in query order every (query, tile) item finds its
    // tile's lookup words and entries cold (Amazon-670K-hard: 7 % L2 hits, the fabric's request ceiling); tile-sorted, the items of a tile
    // run back to back on one XCD and share them
    auto sorts_rest = [&](size_t l) { return m.sort_rest != 0 && !o.stats_out && m.layers[l]->n_tiles <= sort_max_tiles() && m.layers[l]->n_tiles >= 64u; };
    {
        size_t hist_max = 0; uint32_t tiles_max = 0; bool any = false;
        for (size_t l = 0; l < T; ++l) {
            const Layer& L = *m.layers[l];
            if (layer_mode(l, nb) != 0 || (n_rows % nb && layer_mode(l, n_rows % nb) != 0) || (m.prune && sorts_rest(l))) {
                any = true;
                hist_max = std::max(hist_max, sort_hist_bytes(nb * beam_in[l] * L.max_tiles_per_parent, L.n_tiles));
                tiles_max = std::max(tiles_max, L.n_tiles);
            }
        }
        if (any) for (int ln = 0; ln < lanes; ++ln) { LaneWs& lw = ws.lane[ln]; lw.items_sorted.reserve(slots_max * k0_item_bytes()); lw.sort_hist.reserve(hist_max); lw.sort_start.reserve(((size_t)tiles_max + 1) * 4); lw.blk_start.reserve(((size_t)tiles_max + 1) * 4); }
    }
    for (int ln = 0; ln < lanes; ++ln) { LaneWs& lw = ws.lane[ln]; lw.cand_off.reserve(nb * bin_max * 4); lw.ncand.reserve(nb * 4); lw.cand.reserve(nb * (uint64_t)cs_max * 4); }
    if (o.stats_out) {
        ws.stats.reserve(T * kStatsPerLayer * sizeof(double));
        XRL_HIP(hipMemsetAsync(ws.stats.p, 0, T * kStatsPerLayer * sizeof(double), stream));
    }

    // ---- streams and cross-stream ordering
    hipStream_t lane_stream[2] = {stream, stream};
    size_t ev_next = 0;
    auto next_event = [&]() -> hipEvent_t {
        if (ev_next == m.events.size()) { hipEvent_t e; XRL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m.events.push_back(e); }
        return m.events[ev_next++];
    };
    if (lanes == 2) {
        if (!m.aux_stream) XRL_HIP(hipStreamCreateWithFlags(&m.aux_stream, hipStreamNonBlocking));
        lane_stream[1] = m.aux_stream;
        hipEvent_t e = next_event();                                   // the auxiliary lane starts after everything queued so far
        XRL_HIP(hipEventRecord(e, stream));
        XRL_HIP(hipStreamWaitEvent(m.aux_stream, e, 0));
    }
    hipEvent_t k1_done = nullptr;                                      // the most recent K1 launch (either lane)
    // the bound-pruning guard (prune_guard_ok, xrl_device.h) prices a query's largest |x| with the model's largest |weight| x max(1, |bias|)
    float prune_wmax = 0.0f;
    for (size_t l = 0; l < T; ++l) prune_wmax = std::max(prune_wmax, m.layers[l]->w_absmax);
    if (!(prune_wmax <= 3.0e38f)) prune_wmax = INFINITY;
    // ---- pruning feedback (Model::fb_*): which layers run UNSTAGED this time because their first stage settled (almost) no query the last times
    bool unst[Model::kFbLayers] = {false}, probe_now[Model::kFbLayers] = {false};
    if (m.prune && m.adaptive) {
        constexpr uint32_t kFbReprobe = 32;                             // an unstaged layer is staged again every so many predicts: the data may have changed
        constexpr uint32_t kPending = 0xFFFFFFFFu;
        if (!m.fb_host) {
            XRL_HIP(hipHostMalloc(reinterpret_cast<void**>(&m.fb_host), 3 * Model::kFbLayers * 4, hipHostMallocDefault));
            for (int i = 0; i < 3 * Model::kFbLayers; ++i) m.fb_host[i] = i < 2 * Model::kFbLayers ? 0u : kPending;
            m.fb_dev.reserve(2 * Model::kFbLayers * 4);
            XRL_HIP(hipMemset(m.fb_dev.p, 0, 2 * Model::kFbLayers * 4));
        }
        volatile uint32_t* fh = m.fb_host;
        for (size_t l = 0; l < T && l < (size_t)Model::kFbLayers; ++l) {
            if (!o.stats_out) {
                if (m.fb_unstaged[l]) {
                    // An unstaged layer is PROBED every kFbReprobe predicts: that one predict stages it, the following ones keep running
                    // unstaged until the probe's counters arrive (a caller that queues predicts without synchronising is many predicts
                    // ahead of the device: waiting for the outcome in the staged state would run all of them staged).
                    bool decided = false, stage_again = false;
                    if (m.fb_probing[l]) {
                        const uint32_t seen = fh[2 * l], sec = fh[2 * l + 1], dseen = seen - m.fb_seen[l], dsec = sec - m.fb_second[l];
                        const uint32_t cnt = fh[2 * Model::kFbLayers + l];
                        if (dseen >= 256u) { decided = true; stage_again = !((double)dsec > 0.7 * (double)dseen); m.fb_seen[l] = seen; m.fb_second[l] = sec; }
                        else if (cnt != kPending && m.fb_tile_slots[l] > 0) { decided = true; stage_again = !((double)cnt > 0.6 * (double)m.fb_tile_slots[l]); }
                        if (decided) { m.fb_probing[l] = 0; m.fb_unstaged_calls[l] = 0; if (stage_again) m.fb_unstaged[l] = 0; }
                    }
                    if (m.fb_unstaged[l] && ++m.fb_unstaged_calls[l] >= kFbReprobe) {    // (an undecided probe -- a batch too small to sample 256 queries -- is repeated)
                        m.fb_unstaged_calls[l] = 0; m.fb_probing[l] = 1; probe_now[l] = true;
                        if (!decided) { m.fb_seen[l] = fh[2 * l]; m.fb_second[l] = fh[2 * l + 1]; }
                        fh[2 * Model::kFbLayers + l] = kPending; m.fb_tile_slots[l] = 0;
                    }
                } else {
                    // query-stationary layers: of the sampled queries that ran staged, how many needed the second pass
                    const uint32_t seen = fh[2 * l], sec = fh[2 * l + 1], dseen = seen - m.fb_seen[l], dsec = sec - m.fb_second[l];
                    if (dseen >= 256u) {
                        if ((double)dsec > 0.7 * (double)dseen) { m.fb_unstaged[l] = 1; m.fb_unstaged_calls[l] = 0; m.fb_probing[l] = 0; }
                        m.fb_seen[l] = seen; m.fb_second[l] = sec;
                    }
                    // tile-format layers: the item count of the last stage against the slots it was sized for
                    const uint32_t cnt = fh[2 * Model::kFbLayers + l];
                    if (cnt != kPending && m.fb_tile_slots[l] > 0 && (double)cnt > 0.6 * (double)m.fb_tile_slots[l]) { m.fb_unstaged[l] = 1; m.fb_unstaged_calls[l] = 0; m.fb_probing[l] = 0; }
                }
            }
            unst[l] = m.fb_unstaged[l] != 0 && !probe_now[l];
        }
    }

    uint64_t batch = 0;
    for (uint64_t row0 = row_begin; row0 < row_end; row0 += nb, ++batch) {
        const uint32_t nrows = (uint32_t)std::min<uint64_t>(nb, row_end - row0);
        const int ln = (int)(batch % (uint64_t)lanes);
        LaneWs& lw = ws.lane[ln];
        hipStream_t S = lane_stream[ln];
        static const bool debug_sync = [] { const char* e = std::getenv("XRL_DEBUG_SYNC"); return e && e[0] == '1'; }();
        auto timed = [&](const char* name, uint32_t layer, auto&& fn) {
            if (debug_sync) {   // XRL_DEBUG_SYNC=1: every launch group is announced and synchronised on its own, a device fault names the kernel family and the layer
                std::fprintf(stderr, "[xrl debug] %s layer %u rows %llu+%u\n", name, layer, (unsigned long long)row0, nrows); std::fflush(stderr);
                fn();
                const hipError_t e = hipStreamSynchronize(S);
                if (e != hipSuccess) fail(std::string("device fault in ") + name + " (layer " + std::to_string(layer) + ", rows " + std::to_string(row0) + "+" + std::to_string(nrows) + "): " + hipGetErrorString(e));
                return;
            }
            if (!m.profiling) { fn(); return; }
            PendingEvent ev; ev.slot = profile_slot(m, name, layer);
            XRL_HIP(hipEventCreate(&ev.a)); XRL_HIP(hipEventCreate(&ev.b));
            XRL_HIP(hipEventRecord(ev.a, S));
            fn();
            XRL_HIP(hipEventRecord(ev.b, S));
            m.pending.push_back(ev);
        };
        // per row batch: lw.x_ok[q] = the pruning guard of query q (also "all x finite" for K1G's fast loop); written by launch_xguard or,
        // for free, by a K1Q launch that runs before the first layer that needs it
        bool x_ok_done = false;
        auto need_x_ok = [&]() {
            if (x_ok_done) return;
            lw.x_ok.reserve((size_t)nb * 4);
            timed("xguard", 0u, [&] { launch_xguard(X, (uint32_t)row0, nrows, prune_wmax, lw.x_ok.as<uint32_t>(), S); });
            x_ok_done = true;
        };
        for (size_t l = 0; l < T; ++l) {
            const Layer& L = *m.layers[l];
            LayerPlan P{};
            P.layer = (int)l;
            P.row0 = (uint32_t)row0; P.nrows = nrows; P.beam_in = beam_in[l]; P.k = k[l];
            P.cand_stride = cstride[l]; P.pp = pp[l];
            P.tune.wpb = m.k1_wpb; P.tune.lds_pad = m.k1_lds_pad; P.tune.ablate = m.k1_ablate; P.tune.k1g_variant = m.k1g_variant; P.tune.pres_mode = m.presence; P.tune.tile_rows = m.tile_rows; P.tune.k2_big_min_k = m.k2_big_min_k;
            P.first_layer = (l == 0 && (!has_init || o.no_prev_pred)) ? 1 : 0;   // no_prev_pred
            P.implicit_root = (l == 0 && !has_init) ? 1 : 0;
            P.bias_first = (m.weight_matrix_type == 1 && !X.dense) ? 1 : 0;
            P.prune = (m.prune && !(l < (size_t)Model::kFbLayers && unst[l])) ? 1 : 0;
            if (m.prune && m.adaptive && !o.stats_out && l < (size_t)Model::kFbLayers) { P.fb_host = m.fb_host; P.fb_dev = m.fb_dev.as<uint32_t>(); }
            BeamDev prev{};
            if (l == 0 && has_init) {
                prev = *o.initial;
                prev.idx += row0 * prev.stride; prev.val += row0 * prev.stride; prev.cnt += row0;
            } else if (l > 0) {
                const int b = (int)((l - 1) & 1);
                prev = BeamDev{lw.beam_idx[b].as<uint32_t>(), lw.beam_val[b].as<float>(), lw.beam_cnt[b].as<uint32_t>(), beam_stride};
            }
            uint32_t *oi, *oc; float* ov; uint32_t os;
            if (l == T - 1) { oi = d_out_idx + row0 * out_stride; ov = d_out_val + row0 * out_stride; oc = d_out_cnt + row0; os = out_stride; }
            else { const int b = (int)(l & 1); oi = lw.beam_idx[b].as<uint32_t>(); ov = lw.beam_val[b].as<float>(); oc = lw.beam_cnt[b].as<uint32_t>(); os = beam_stride; }

            // layers held in the dense row format: the whole layer is one query-stationary kernel (beam in, beam out)
            if (csc) {
                // CSC route (weight_matrix_type CSC, single-layer API): one (query, child) dot product per candidate, bias first
                Layer& Lm = *m.layers[l];
                ensure_device_csc(Lm);
                timed("k0_prolongate", (uint32_t)l, [&] { launch_k0_prolongate(L.dev, P, X, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.items.p, S); });
                timed("k1c_csc", (uint32_t)l, [&] { launch_k1c_csc(L.dev, Lm.d_csc_ptr.as<uint64_t>(), Lm.d_csc_idx.as<uint32_t>(), Lm.d_csc_val.as<float>(), P, X, prev,
                                                                   lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), S); });
                timed("k2_topk", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S); });
                continue;
            }
            if (!o.stats_out && layer_mode(l, nrows) == 3) {
                need_x_ok();   // once per row batch: which dense query rows are finite (K1G's fast loop needs it on layers with missing cells) + the pruning guard
                // ---- exact bound pruning (see the tile-format path below): the GEMM over the children of the J best beam parents first,
                //      then a second, tile-sorted GEMM over the remaining slots of the queries whose top-k is not final yet.  J covers about
                //      one candidate register (64 candidates), like K1Q's first stage.
                if (m.prune && !P.implicit_root && !P.first_layer && P.pp.kind != PP_NOOP && beam_in[l] > 1 && k2_wave_path(P)) {
                    const uint32_t J = k1g_first_slots(L, beam_in[l], m.k1g_first);
                    const uint64_t slots_a = (uint64_t)nrows * J * L.max_tiles_per_parent, slots_b = (uint64_t)nrows * (beam_in[l] - J) * L.max_tiles_per_parent;
                    lw.prune_done.reserve((size_t)nb * 4); lw.prune_cnt.reserve(256);
                    LayerPlan PA = P; PA.beam_in = J;
                    LayerPlan PB = P; PB.beam_in = beam_in[l] - J;
                    timed("k0_prolongate", (uint32_t)l, [&] { launch_k0_prolongate(L.dev, P, X, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.items.p, S, J); });
                    timed("k1_sort_items", (uint32_t)l, [&] { launch_sort_items(L.dev, slots_a, lw.items.p, lw.items_sorted.p, lw.sort_hist.as<uint32_t>(), lw.sort_start.as<uint32_t>(), S); });
                    timed("k1g_dense_x", (uint32_t)l, [&] { launch_k1g(L.dev, PA, X, lw.items_sorted.p, lw.sort_start.as<uint32_t>(), lw.blk_start.as<uint32_t>(), lw.x_ok.as<uint32_t>(), lw.cand.as<float>(), S); });
                    timed("k2_topk", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S,
                                                                       J, (uint32_t)L.cand_bound(J), lw.prune_done.as<uint32_t>(), nullptr, lw.x_ok.as<uint32_t>()); });
                    timed("k0b_remaining", (uint32_t)l, [&] { launch_k0b_remaining(L.dev, P, X, prev, lw.cand_off.as<uint32_t>(), lw.prune_done.as<uint32_t>(), J, lw.items.p,
                                                                                   lw.prune_cnt.as<uint32_t>(), S); });
                    timed("k1_sort_items_rest", (uint32_t)l, [&] { launch_sort_items(L.dev, slots_b, lw.items.p, lw.items_sorted.p, lw.sort_hist.as<uint32_t>(), lw.sort_start.as<uint32_t>(), S,
                                                                                     lw.prune_cnt.as<uint32_t>()); });
                    timed("k1g_dense_x_rest", (uint32_t)l, [&] { launch_k1g(L.dev, PB, X, lw.items_sorted.p, lw.sort_start.as<uint32_t>(), lw.blk_start.as<uint32_t>(), lw.x_ok.as<uint32_t>(), lw.cand.as<float>(), S); });
                    timed("k2_topk_rest", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S,
                                                                            0, 0, nullptr, lw.prune_done.as<uint32_t>()); });
                    continue;
                }
                timed("k0_prolongate", (uint32_t)l, [&] { launch_k0_prolongate(L.dev, P, X, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.items.p, S); });
                const uint64_t n_slots3 = (uint64_t)nrows * beam_in[l] * L.max_tiles_per_parent;
                timed("k1_sort_items", (uint32_t)l, [&] { launch_sort_items(L.dev, n_slots3, lw.items.p, lw.items_sorted.p, lw.sort_hist.as<uint32_t>(), lw.sort_start.as<uint32_t>(), S); });
                timed("k1g_dense_x", (uint32_t)l, [&] { launch_k1g(L.dev, P, X, lw.items_sorted.p, lw.sort_start.as<uint32_t>(), lw.blk_start.as<uint32_t>(), lw.x_ok.as<uint32_t>(), lw.cand.as<float>(), S); });
                timed("k2_topk", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S); });
                continue;
            }
            auto runs_k1q = [&](size_t ll) { return m.dense_layers && !o.stats_out && !csc && layer_mode(ll, nrows) != 3 && k1q_regs(m.layers[ll]->dev, beam_in[ll], k[ll], X.dense != 0 || m.dense_layers >= 2) != 0; };
            if (runs_k1q(l)) {
                // consecutive dense-format layers run in ONE launch: the wavefront that owns a query carries its beam through them in LDS
                size_t l1 = l;
                // (only narrow layers share a launch: a fused kernel is compiled for -- and holds the registers of -- its widest layer)
                auto narrow = [&](size_t ll) { return k1q_regs(m.layers[ll]->dev, beam_in[ll], k[ll], X.dense != 0 || m.dense_layers >= 2) <= (uint32_t)std::min(3, m.k1q_fuse); };
                while (m.k1q_fuse && narrow(l) && l1 + 1 < T && l1 + 1 - l < 8 && runs_k1q(l1 + 1) && narrow(l1 + 1)) ++l1;
                const LayerDev* Ls[8]; LayerPlan Ps[8];
                for (size_t ll = l; ll <= l1; ++ll) {
                    Ls[ll - l] = &m.layers[ll]->dev;
                    LayerPlan Q = P;
                    Q.layer = (int)ll; Q.beam_in = beam_in[ll]; Q.k = k[ll]; Q.cand_stride = cstride[ll]; Q.pp = pp[ll];
                    Q.first_layer = (ll == 0 && (!has_init || o.no_prev_pred)) ? 1 : 0;
                    Q.implicit_root = (ll == 0 && !has_init) ? 1 : 0;
                    Q.prune = (m.prune && !(ll < (size_t)Model::kFbLayers && unst[ll])) ? 1 : 0;
                    Ps[ll - l] = Q;
                }
                auto beam_out = [&](size_t ll, uint32_t*& qi, float*& qv, uint32_t*& qc, uint32_t& qs) {
                    if (ll == T - 1) { qi = d_out_idx + row0 * out_stride; qv = d_out_val + row0 * out_stride; qc = d_out_cnt + row0; qs = out_stride; }
                    else { const int b = (int)(ll & 1); qi = lw.beam_idx[b].as<uint32_t>(); qv = lw.beam_val[b].as<float>(); qc = lw.beam_cnt[b].as<uint32_t>(); qs = beam_stride; }
                };
                auto launch_name = [&](size_t a, size_t b) {
                    return (b > a) ? std::string(X.dense ? "k1q_fused_x_" : "k1q_fused_") + std::to_string(a) + "_" + std::to_string(b) : std::string(X.dense ? "k1q_dense_x" : "k1q_dense");
                };
                // (a later layer of this batch that is NOT served by K1Q decides its pruning in K2 and needs the guard flags: this launch writes them)
                uint32_t* xok_out = nullptr;
                if (!x_ok_done && m.prune && l1 + 1 < T) { lw.x_ok.reserve((size_t)nb * 4); xok_out = lw.x_ok.as<uint32_t>(); x_ok_done = true; }
                // SORTED launch of the group's last layer (option qsort): a layer of many parents whose matrix is far larger than the L2s is
                // request-bound -- one 64-byte fabric request per (feature, beam parent) and query -- and the queries that share beam
                // parents are scattered over the batch.  The layers before it run as usual; the queries are then counting-sorted by the best
                // parent of the beam they produced and the last layer runs in that order, every XCD on a contiguous range of it.
                const size_t ls = l1;
                const bool sorted_last = m.qsort != 0 && !X.dense && ls > 0 && !(ls == 0 && has_init) && nrows >= (uint32_t)std::max(1, m.qsort_min_rows) &&
                                         m.layers[ls]->dev.n_parents >= (uint32_t)std::max(2, m.qsort_min_parents) && m.layers[ls]->dev.n_parents <= qsort_max_keys();
                if (sorted_last) {
                    BeamDev pv = prev;
                    if (ls > l) {
                        uint32_t *qi, *qc; float* qv; uint32_t qs;
                        beam_out(ls - 1, qi, qv, qc, qs);
                        timed(launch_name(l, ls - 1).c_str(), (uint32_t)l, [&] { launch_k1q(Ls, Ps, (int)(ls - l), X, prev, qi, qv, qc, qs, S, prune_wmax, xok_out); });
                        xok_out = nullptr;
                        pv = BeamDev{qi, qv, qc, qs};
                    }
                    const uint32_t nk = m.layers[ls]->dev.n_parents;
                    lw.qperm.reserve((size_t)nb * 4); lw.qsort_hist.reserve(qsort_hist_bytes((uint32_t)nb, nk)); lw.qsort_start.reserve(((size_t)nk + 1) * 4);
                    timed("k1_sort_queries", (uint32_t)ls, [&] { launch_sort_queries(pv, nrows, nk, lw.qsort_hist.as<uint32_t>(), lw.qsort_start.as<uint32_t>(), lw.qperm.as<uint32_t>(), S); });
                    uint32_t *qi, *qc; float* qv; uint32_t qs;
                    beam_out(ls, qi, qv, qc, qs);
                    timed(X.dense ? "k1q_dense_x" : "k1q_dense", (uint32_t)ls, [&] { launch_k1q(Ls + (ls - l), Ps + (ls - l), 1, X, pv, qi, qv, qc, qs, S, prune_wmax, xok_out, lw.qperm.as<uint32_t>()); });
                    l = l1;
                    continue;
                }
                uint32_t *qi, *qc; float* qv; uint32_t qs;
                beam_out(l1, qi, qv, qc, qs);
                timed(launch_name(l, l1).c_str(), (uint32_t)l, [&] { launch_k1q(Ls, Ps, (int)(l1 - l + 1), X, prev, qi, qv, qc, qs, S, prune_wmax, xok_out); });
                l = l1;
                continue;
            }
            int g = m.k1_group > 0 ? m.k1_group : k1_auto_group(L.dev, L, X.dense);
            int mode = layer_mode(l, nrows);
            const bool fb_unstaged = l < (size_t)Model::kFbLayers && unst[l];
            if (fb_unstaged && mode == 0 && sorts_rest(l)) mode = 1;    // pruning feedback: everything in one pass, on tile-sorted items
            // ---- exact bound pruning, tile format: score the children of the best beam parent first (K0 -> K1 -> K2 on one slot), then
            //      only the remaining slots of the queries whose top-k is not final yet (see K2Args).  Needs a combiner (a child's score
            //      is then <= its parent's) and the register top-k kernel.
            bool pruned = m.prune && !fb_unstaged && !P.implicit_root && !P.first_layer && P.pp.kind != PP_NOOP && beam_in[l] > 1 && k2_wave_path(P);
            uint32_t J = 1;
            if (o.stats_out) {
                // the stats pass walks the tile format whatever kernel the timed pass runs: stage it the way THAT kernel stages the layer, so
                // that the work it counts is the work the timed kernels evaluate (K1G: J parents first; K1Q: the parents of candidate
                // register 0 first, nothing skipped when the whole beam fits one register)
                const bool dense_x = X.dense != 0 || m.dense_layers >= 2;
                if (mode == 3) J = k1g_first_slots(L, beam_in[l], m.k1g_first);      // (the SAME staging the timed K1G path uses: ADVICE r3)
                else if (m.dense_layers && k1q_regs(L.dev, beam_in[l], k[l], dense_x) != 0) {
                    if (k1q_regs(L.dev, beam_in[l], k[l], dense_x) <= 1) pruned = false;
                    else J = std::max<uint32_t>(1, (64u >> L.dev.d_gp_log2) / std::max<uint32_t>(1, L.dev.d_max_tiles));
                }
                if (J >= beam_in[l]) pruned = false;
            } else if (mode != 0) pruned = false;
            if (pruned) {
                // stages of beam slots: [0, J) first, then -- wide beams only (>= 16 parents: Wiki10-31K's 20) -- a MIDDLE stage [J, J2) before
                // "everything else": with 20 parents of ~60 children and k = 20 the first parent alone rarely settles the top-k, the first
                // five usually do, and the last stage then sees few queries.  Every stage: k0b (the unfinished queries' items of its slots) ->
                // sort -> K1 -> K2 over the candidates of all slots scored so far (+ the done flag for the next stage).
                uint32_t stage_end[3]; int n_stage = 0;
                stage_end[n_stage++] = J;
                if (m.prune_mid != 0 && beam_in[l] >= 16u && !o.stats_out) stage_end[n_stage++] = std::min<uint32_t>(beam_in[l] - 1u, J + 4u);
                stage_end[n_stage++] = beam_in[l];
                const uint64_t slots_b = (uint64_t)nrows * (beam_in[l] - J) * L.max_tiles_per_parent;
                lw.prune_done.reserve((size_t)nb * 4); lw.prune_cnt.reserve(256);
                lw.items_sorted.reserve(slots_max * k0_item_bytes());
                need_x_ok();
                LayerPlan PA = P; PA.beam_in = J;                  // (K1 sizes its grid from beam_in x tiles per parent)
                timed("k0_prolongate", (uint32_t)l, [&] { launch_k0_prolongate(L.dev, P, X, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.items.p, S, J); });
                if (lanes == 2 && k1_done) XRL_HIP(hipStreamWaitEvent(S, k1_done, 0));
                timed(X.dense ? "k1_dense" : "k1_sparse", (uint32_t)l, [&] { launch_k1(L.dev, PA, X, lw.items.p, nullptr, lw.cand.as<float>(), g, S); });
                timed("k2_topk", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S,
                                                                   J, (uint32_t)L.cand_bound(J), lw.prune_done.as<uint32_t>(), nullptr, lw.x_ok.as<uint32_t>()); });
                if (o.stats_out) XRL_HIP(hipMemsetAsync(lw.items_sorted.p, 0xFF, slots_b * k0_item_bytes(), S));   // the stats pass walks the whole list: unused slots read as "no tile"
                bool srt = sorts_rest(l);
                // ... unless the previous predicts left (almost) nothing for the later stages: four tiny launches of the sort then cost more than
                // the locality buys (a 61 250-row shard of Amazon-670K: 0.99 -> 0.93 ms per step, profiles/r05_k1q/README.md).  The count is the one the
                // pruning feedback already samples (the last stage's K1 launch writes it to a host-visible word); results never depend on it.
                if (srt && P.fb_host && l < (size_t)Model::kFbLayers && m.sort_rest_min > 0) {
                    const uint32_t seen = static_cast<volatile uint32_t*>(m.fb_host)[2 * Model::kFbLayers + l];
                    // (measured, Amazon-670K shape, 8192 leaf tiles: 78 k items = 9.5 per tile -- K1 runs 0.295 ms sorted, 0.298 unsorted, the sort costs 0.062;
                    //  4.4 M items = 537 per tile on the hard model -- unsorted K1 finds 7 % of its lines in the L2, sorted 94 %: the sort pays once
                    //  a tile's lookup words and entries are re-used a few dozen times)
                    const uint64_t need = std::max<uint64_t>((uint64_t)m.sort_rest_min, 32ull * L.n_tiles);
                    if (seen != 0xFFFFFFFFu && (uint64_t)seen < need) srt = false;
                }
                if (P.fb_host && l < (size_t)Model::kFbLayers) m.fb_tile_slots[l] = (uint64_t)nrows * (beam_in[l] - stage_end[n_stage - 2]) * L.max_tiles_per_parent;
                for (int st = 1; st < n_stage; ++st) {
                    const uint32_t r0 = stage_end[st - 1], r1 = stage_end[st];
                    const bool last = st == n_stage - 1;
                    const uint64_t slots_s = (uint64_t)nrows * (r1 - r0) * L.max_tiles_per_parent;
                    LayerPlan PB = P; PB.beam_in = r1 - r0;
                    PB.tune.wpb = 4;                               // a later stage's grid is sized for "nothing pruned": mostly empty wavefronts, 4 per workgroup to dispatch fewer groups
                    const char* sfx = last ? "_rest" : "_mid";
                    // (sorted: the compacted list goes where the first phase's items were -- K1 has consumed them -- and is sorted into items_sorted)
                    timed(last ? "k0b_remaining" : "k0b_remaining_mid", (uint32_t)l, [&] { launch_k0b_remaining(L.dev, P, X, prev, lw.cand_off.as<uint32_t>(), lw.prune_done.as<uint32_t>(), r0,
                                                                                                                 srt ? lw.items.p : lw.items_sorted.p, lw.prune_cnt.as<uint32_t>(), S, r1); });
                    if (srt) timed((std::string("k1_sort_items") + sfx).c_str(), (uint32_t)l, [&] { launch_sort_items(L.dev, slots_s, lw.items.p, lw.items_sorted.p, lw.sort_hist.as<uint32_t>(),
                                                                                                                      lw.sort_start.as<uint32_t>(), S, lw.prune_cnt.as<uint32_t>()); });
                    timed((std::string(X.dense ? "k1_dense" : "k1_sparse") + sfx).c_str(), (uint32_t)l, [&] {
                        launch_k1(L.dev, PB, X, lw.items_sorted.p, srt ? lw.sort_start.as<uint32_t>() + L.n_tiles : lw.prune_cnt.as<uint32_t>(), lw.cand.as<float>(), g, S); });
                    if (lanes == 2 && last) { k1_done = next_event(); XRL_HIP(hipEventRecord(k1_done, S)); }
                    // (a middle stage selects among the slots scored so far and renews the done flags; queries finished earlier are skipped: one buffer serves both)
                    if (last) timed("k2_topk_rest", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S,
                                                                                      0, 0, nullptr, lw.prune_done.as<uint32_t>()); });
                    else timed("k2_topk_mid", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S,
                                                                                 r1, (uint32_t)L.cand_bound(r1), lw.prune_done.as<uint32_t>(), lw.prune_done.as<uint32_t>(), lw.x_ok.as<uint32_t>()); });
                    if (o.stats_out && last) {
                        launch_stats(L.dev, P, X, prev, lw.ncand.as<uint32_t>(), lw.items.p, ws.stats.as<double>() + kStatsPerLayer * l, S, (uint64_t)nrows * J * L.max_tiles_per_parent);
                        launch_stats(L.dev, PB, X, prev, nullptr, lw.items_sorted.p, ws.stats.as<double>() + kStatsPerLayer * l, S, slots_b);
                    }
                }
                continue;
            }
            timed("k0_prolongate", (uint32_t)l, [&] { launch_k0_prolongate(L.dev, P, X, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.items.p, S); });
            const uint64_t n_slots = (uint64_t)nrows * beam_in[l] * L.max_tiles_per_parent;
            if (mode != 0) timed("k1_sort_items", (uint32_t)l, [&] { launch_sort_items(L.dev, n_slots, lw.items.p, lw.items_sorted.p, lw.sort_hist.as<uint32_t>(), lw.sort_start.as<uint32_t>(), S); });
            if (lanes == 2 && k1_done) XRL_HIP(hipStreamWaitEvent(S, k1_done, 0));   // K1 launches take turns across the lanes
            LayerPlan PU = P; PU.fb_host = nullptr;   // (an unstaged pass's item count is not a feedback sample: a probe's outcome may be pending)
            timed(X.dense ? "k1_dense" : "k1_sparse", (uint32_t)l, [&] {
                    launch_k1(L.dev, PU, X, mode == 1 ? lw.items_sorted.p : lw.items.p, mode == 1 ? lw.sort_start.as<uint32_t>() + L.n_tiles : nullptr,
                              lw.cand.as<float>(), g, S); });
            if (lanes == 2) { k1_done = next_event(); XRL_HIP(hipEventRecord(k1_done, S)); }
            timed("k2_topk", (uint32_t)l, [&] { launch_k2_topk(L.dev, P, prev, lw.cand_off.as<uint32_t>(), lw.ncand.as<uint32_t>(), lw.cand.as<float>(), oi, ov, oc, os, S); });
            if (o.stats_out) launch_stats(L.dev, P, X, prev, lw.ncand.as<uint32_t>(), lw.items.p, ws.stats.as<double>() + kStatsPerLayer * l, S);
        }
    }
    if (lanes == 2) {                                                   // the caller's stream continues after the auxiliary lane
        hipEvent_t e = next_event();
        XRL_HIP(hipEventRecord(e, m.aux_stream));
        XRL_HIP(hipStreamWaitEvent(stream, e, 0));
    }

    if (!m.ws_done) XRL_HIP(hipEventCreateWithFlags(&m.ws_done, hipEventDisableTiming));
    XRL_HIP(hipEventRecord(m.ws_done, stream)); m.ws_stream = stream;

    if (o.stats_out) {
        // per layer (kStatsPerLayer doubles): [0] algorithmic bytes of the reference chunks streamed (8E + 4R + 4(R+1) each,
        // SURVEY.md 8d), [1] candidates evaluated, [2..7] matched work of the (query, tile) items (launch_stats)
        XRL_HIP(hipStreamSynchronize(stream));
        XRL_HIP(hipMemcpy(o.stats_out, ws.stats.p, T * kStatsPerLayer * sizeof(double), hipMemcpyDeviceToHost));
    } else if (sync) {
        XRL_HIP(hipStreamSynchronize(stream));
    }
}

}  // namespace xrl

namespace xrl {
void resolve_profile(Model& m) {
    if (m.pending.empty()) return;
    XRL_HIP(hipSetDevice(m.device));
    XRL_HIP(hipDeviceSynchronize());
    for (auto& ev : m.pending) {
        float ms = 0.f;
        XRL_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
        m.profile[ev.slot].ms += ms;
        m.profile[ev.slot].launches += 1;
        (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b);
    }
    m.pending.clear();
}
}  // namespace xrl
