// hf_create.h — hf_create (include/hmm_flagger_hip.h): what EM_construct + EM_renewParametersAndEstimatorsFromModel set up per chunk and
// iteration in the reference (programs/submodules/hmm/hmm.c:253-298), done ONCE per run here: the windows go up, every window gets its
// packed record (k_setup / window_record), its row of A = T∘e and the position of its pair record; the segment list of the
// workgroup-per-segment forward-backward (hf_seg.h) and the plan of the statistics by emission row (hf_rows.h) are built.
//
// Round 6: five steps with the data they hand to each other named in ONE struct (CreateWork) instead of an 850-line
// function — included by hf_estep.hip behind the context, the kernels and the upload helpers:
//   create_pack_windows      first pass over the windows: packed upload words, packed records, the contig-end ("slow") windows,
//                            which emission keys and (key, transition class) rows occur; the window arrays go up
//   create_device_store      the context's device arrays and pinned result block, k_setup / k_regmask enqueued
//   create_key_tables        slow-window list, emission keys, the per-pass row tables, tiles of the per-chunk statistics
//   create_rows_and_segments rows of A (second pass), sub-passes, segment descriptors
//   create_statistics_plan   groups, row slots, the position of every window's record (third pass), the plan's uploads
//   create_launch_config     one-launch guard, cached row blocks, XCD plan, the job list of the per-pass tables, environment switches
// Everything a step allocates on the device comes out of the context's slab and is released by hf_destroy: a step that fails just
// returns its code and hf_create destroys the context.
#pragma once

namespace {
// Pinned staging buffers of 4 bytes per window for everything hf_create moves between host and device (pageable copies were measured at
// ~1 GB/s): a = P0 packed windows up; b = P1 the rows of A up | P2 the record positions up | P3 (never uploaded) the packed records as the
// host computes them; c = 8 MiB for the small arrays (UploadStage).  They come from the process-wide cache (PinCache: pinning costs ~0.2 ms
// per MB); a first-time b is pinned on a helper thread while the first pass over the windows runs.  Back to the cache when hf_create leaves
// (its uploads are complete by then).
struct PinnedArena {
    char *a = nullptr, *b = nullptr, *c = nullptr; std::thread th;
    ~PinnedArena() { if (th.joinable()) th.join(); pin_cache().release(a); pin_cache().release(b); pin_cache().release(c); }
};

// Which (region, x, x_prev) emission keys and which (key, transition class) rows of A occur at interior windows: one 16-bit word per key, at a
// stride of 256 per coverage value (the largest coverage is only known after the pass), bit c = class c occurs (c < HF_AROW_CLASSES).
// PRIVATE per participating thread and OR-ed afterwards (round 6): rounds 2-5 marked one shared byte table — a few thousand popular cells
// that all sixteen threads set in their first microseconds, 64 of them per cache line: the lines bounced between the cores and the first
// pass ran at ~30 ns per window (2.0-2.4 ms of a 4 ms hf_create on a fresh box, every part 10x slower than the second pass's, which touches
// the same bytes: profiles/r06_parts_trace.txt).
struct KeyMarks {
    size_t words = 0;
    uint64_t gen = 0;
    std::atomic<int> claimed{0};
    std::vector<std::vector<uint16_t>> tabs;             // one per participating thread, claimed on the thread's first part, owned by THIS hf_create
    KeyMarks() : tabs(HF_PARTS + 8) {}                   // (hf_multi's ranks create their contexts concurrently on one pool: nothing here is shared between two calls)
    uint16_t* mine() {
        static thread_local uint64_t tl_gen = 0;
        static thread_local int tl_slot = -1;
        if (tl_gen != gen) {
            tl_slot = claimed.fetch_add(1);
            tl_gen = gen;
            if (tl_slot < (int) tabs.size()) tabs[(size_t) tl_slot].assign(words, 0);
        }
        return tl_slot < (int) tabs.size() ? tabs[(size_t) tl_slot].data() : nullptr;
    }
};
std::atomic<uint64_t> g_create_gen{0};

struct CreateWork {
    hf_ctx* ctx = nullptr; const hf_windows* w = nullptr;
    int n_regions = 1, max_comps = 1, device = 0, algo = HF_ALGO_SCAN;
    size_t N = 0, C = 0; int32_t maxT = 0;
    // phases (hf_create_phases; HF_HOST_TRACE prints them)
    bool ctrace = false;
    std::chrono::steady_clock::time_point t_begin, t_last;
    void phase(const char* name) {
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - t_last).count();
        ctx->create_phases.emplace_back(name, ms);
        if (ctrace) std::fprintf(stderr, "[hf_create] %-34s %7.2f ms\n", name, ms);
        t_last = now;
    }
    // staging
    PinnedArena arena; UploadStage stg;
    uint32_t* P0 = nullptr;                  // packed upload words (pinned)
    int32_t *P1 = nullptr, *P2 = nullptr;    // rows of A / record positions (pinned; set once the helper thread has been joined)
    std::unique_ptr<uint32_t[]> hrec_own; uint32_t* hrec = nullptr;   // the packed records as the host computes them: in P3, or on the heap (first context of a process)
    // upload temporaries (slab memory: released with the context)
    uint32_t* d_packed = nullptr; uint16_t *d_cov = nullptr, *d_mapq = nullptr, *d_clip = nullptr; uint64_t* d_annot = nullptr;
    int32_t *d_cs = nullptr, *d_ce = nullptr, *d_cl = nullptr, *d_spare = nullptr;
    // create_pack_windows ->
    std::vector<std::vector<int64_t>> part_slow; std::vector<int32_t> nslow;   // slow windows per part of the chunk list (ascending), per chunk
    std::vector<uint16_t> keymark;           // OR of the threads' KeyMarks tables
    // create_key_tables ->
    std::vector<int64_t> slow; std::vector<int32_t> soff, keys;
    // create_rows_and_segments ->
    int32_t* h_arow = nullptr; std::vector<int32_t> h_arow_src;             // window -> row of A; row of A -> emission row (both also on the device)
    std::vector<std::vector<int32_t>> pcnt; size_t plan_ppt = 4;             // pairs per (sub-pass, row of A), per part of the chunk list; parts per thread of passes 2 and 3
    int n_sub = 1; std::vector<int32_t> sub_of;                              // sub-pass of every chunk
    std::vector<int32_t> cseg0;                                              // first segment of every chunk
    bool key_seen(size_t k256) const { return keymark[k256] != 0; }
    bool key_class(size_t k256, size_t cl) const { return (keymark[k256] >> cl) & 1u; }
};
#define CTRY(x) do { const int rc_ = (x); if (rc_) return rc_; } while (0)
#define CALLOC(p, bytes) do { (p) = static_cast<decltype(p)>(ctx_alloc(ctx, (bytes))); \
    if (!(p)) return set_err(HF_E_HIP, "hf_create: out of device memory"); } while (0)

// ------------------------------------------------------------------------------------------
// 1. ONE pass over the windows: the packed upload word, the packed record (as k_setup computes it: window_record), the largest
// coverage, the contig-end ("slow") windows, and the key marks; then the window arrays go up (one packed word per window when every value
// fits a byte — window values are at most 250, chunk.c:393-441 — the four arrays as they are otherwise)
// ------------------------------------------------------------------------------------------
int create_pack_windows(CreateWork& cw) {
    hf_ctx* ctx = cw.ctx; const hf_windows* w = cw.w;
    const size_t N = cw.N, C = cw.C;
    const int n_regions = cw.n_regions, device = cw.device;
    cw.arena.c = pin_cache().acquire((size_t) 8 << 20);
    cw.stg.p = cw.arena.c; cw.stg.cap = cw.arena.c ? (size_t) 8 << 20 : 0;
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_off, w->chunk_off, C + 1));
    cw.phase("context, chunk offsets up");
    if (N > 0) {
        cw.arena.a = pin_cache().acquire(N * 4);
        if (!cw.arena.a) return set_err(HF_E_HIP, "hf_create: out of host memory");
        cw.arena.b = pin_cache().try_acquire(3 * N * 4);
        PinnedArena* const ar = &cw.arena;
        if (!cw.arena.b) cw.arena.th = std::thread([ar, device, N] { if (hipSetDevice(device) == hipSuccess) ar->b = pin_cache().acquire(3 * N * 4); });
    }
    cw.phase("pinned staging buffer");
    cw.P0 = reinterpret_cast<uint32_t*>(cw.arena.a);
    cw.part_slow.assign(HF_PARTS, {});
    cw.nslow.assign(C, 0);
    if (cw.arena.b) cw.hrec = reinterpret_cast<uint32_t*>(cw.arena.b + 2 * N * 4);
    else { cw.hrec_own.reset(N ? new uint32_t[N] : nullptr); cw.hrec = cw.hrec_own.get(); }
    std::atomic<int> wide{0}, bad_region{0};
    std::atomic<unsigned> maxx_all{0};
    KeyMarks marks;
    marks.words = (size_t) n_regions << 16;
    marks.gen = g_create_gen.fetch_add(1) + 1;
    if (N > 0) {
        const std::shared_ptr<const ValidityLut> vlut = validity_lut(w->max_high_mapq_ratio, w->min_high_mapq_ratio, w->min_highly_clipped_ratio);
        const bool lut_ok = vlut->monotone;
        uint32_t* const stage = cw.P0; uint32_t* const hrec_w = cw.hrec;
        par_chunks(w->chunk_off, C, [&](size_t c0, size_t c1, size_t part) {
            bool big = false, badr = false;
            unsigned mx = 0;
            std::vector<int64_t>& mine = cw.part_slow[part];
            // (everything the loop touches through LOCAL restrict pointers: the compiler otherwise reloads every base pointer in every iteration)
            uint16_t* const __restrict__ mark_p = marks.mine();
            const uint16_t* const __restrict__ covp = w->cov; const uint16_t* const __restrict__ mqp = w->mapq; const uint16_t* const __restrict__ clp = w->clip;
            const uint64_t* const __restrict__ annp = w->annot;
            uint32_t* const __restrict__ stagep = stage; uint32_t* const __restrict__ hrecp = hrec_w;
            const int16_t* const __restrict__ dup_le = reinterpret_cast<const int16_t*>(vlut->dup_le);
            const uint16_t* const __restrict__ col_ge = vlut->col_ge; const uint16_t* const __restrict__ end_ge = vlut->end_ge;
            const unsigned nreg = (unsigned) n_regions;
            for (size_t c = c0; c < c1; c++) {
                const size_t before = mine.size();
                const size_t t0 = (size_t) w->chunk_off[c], te = (size_t) w->chunk_off[c + 1];
                const int cs_ = w->chunk_s[c], ce_ = w->chunk_e[c], cl_ = w->chunk_ctg_len[c];
                // INTERIOR columns [ia, ib): beta_t is beta_star by construction (hmm.c:301-316: l = mid - L + 1 and u = mid, so u - l = L - 1)
                // — mid is non-decreasing in the column, so the two conditions cut a prefix and a suffix of the chunk.  Those windows take
                // their record from the validity thresholds (no division, no beta arithmetic); the others — and any window with a value
                // above 255 — go through window_record, the function k_setup runs.  The result is the same bits either way (HF_CREATE_VERIFY).
                int64_t ia = 1, ib = (int64_t) (te - t0);
                if (!lut_ok) ib = ia;
                else if (w->adjust_contig_ends) {
                    const int Lr = w->mean_read_len;
                    const int l2 = (int) (-(1 - w->min_read_frac) * Lr), u2 = (int) (cl_ - w->min_read_frac * Lr);
                    auto mid_of = [&](int64_t col) {
                        const int icol = (int) col;
                        const int a1 = (int) (cs_ + (double) w->window_len * (icol + 0.5));
                        const int a2 = (int) ((cs_ + (double) w->window_len * icol + ce_) / 2);
                        return a1 < a2 ? a1 : a2;
                    };
                    while (ia < ib && !(mid_of(ia) - Lr + 1 >= l2)) ia++;
                    while (ib > ia && !(mid_of(ib - 1) <= u2)) ib--;
                }
                unsigned pre_region = 0, xp = 0;
                auto slow_window = [&](size_t t) {              // the general path: window_record
                    const unsigned cv = covp[t], mq = mqp[t], cp = clp[t];
                    const unsigned region = (unsigned) (annp[t] >> 58);
                    const unsigned x = cv & 0xffu;
                    big |= (cv | mq | cp) > 0xffu;
                    if (x > mx) mx = x;
                    stagep[t] = cv | (mq << 8) | (cp << 16) | (region << 24);
                    double bt;
                    const uint32_t r = window_record(cv, mq, cp, region, pre_region, (int64_t) (t - t0), cs_, ce_, cl_, w->window_len,
                                                     w->mean_read_len, w->adjust_contig_ends, w->min_read_frac, w->max_high_mapq_ratio,
                                                     w->min_high_mapq_ratio, w->min_highly_clipped_ratio, ctx->beta_star, &bt);
                    hrecp[t] = r;
                    if (region >= nreg) badr = true;
                    else if (REC_SLOW(r)) mine.push_back((int64_t) t);
                    else mark_p[((size_t) region << 16) | (x << 8) | xp] |= (uint16_t) (1u << (REC_REGCHG(r) ? 8u : REC_VMASK(r)));
                    pre_region = region; xp = x;
                };
                const size_t ta = t0 + (size_t) ia < te ? t0 + (size_t) ia : te, tb = t0 + (size_t) ib > ta ? t0 + (size_t) ib : ta;
                for (size_t t = t0; t < ta; t++) slow_window(t);
                for (size_t t = ta; t < tb; t++) {              // the interior: thresholds, no beta, never slow
                    const unsigned cv = covp[t], mq = mqp[t], cp = clp[t];
                    if ((cv | mq | cp) > 0xffu) { slow_window(t); continue; }
                    const unsigned region = (unsigned) (annp[t] >> 58);
                    if (cv > mx) mx = cv;
                    stagep[t] = cv | (mq << 8) | (cp << 16) | (region << 24);
                    const unsigned vm = ((int) mq <= (int) dup_le[cv] ? 1u : 0u) | (mq >= col_ge[cv] ? 2u : 0u) | (cp >= end_ge[cv] ? 4u : 0u);
                    const bool regchg = pre_region != region;
                    hrecp[t] = cv | (region << 8) | (vm << 16) | (regchg ? 1u << 20 : 0u);
                    if (region >= nreg) badr = true;
                    else mark_p[((size_t) region << 16) | (cv << 8) | xp] |= (uint16_t) (1u << (regchg ? 8u : vm));
                    pre_region = region; xp = cv;
                }
                for (size_t t = tb; t < te; t++) slow_window(t);
                cw.nslow[c] = (int32_t) (mine.size() - before);
            }
            if (big) wide.store(1, std::memory_order_relaxed);
            if (badr) bad_region.store(1, std::memory_order_relaxed);
            unsigned cur = maxx_all.load(std::memory_order_relaxed);
            while (mx > cur && !maxx_all.compare_exchange_weak(cur, mx, std::memory_order_relaxed)) {}
        });
        ctx->M = (int) maxx_all.load() + 1;
    }
    {   // the threads' marks, OR-ed (a thread's table stays its own until its next hf_create: nothing else runs on the pool meanwhile)
        cw.keymark.assign(marks.words, 0);
        uint16_t* const __restrict__ dst = cw.keymark.data();
        for (const auto& t : marks.tabs) if (!t.empty()) for (size_t i = 0; i < marks.words; i++) dst[i] |= t[i];
    }
    cw.phase("one pass: packing, records, keys");
    if (bad_region.load()) return set_err(HF_E_REGION, "a window's region index is >= n_regions");
    if (N > 0) {
        if (w->chunk_off[0] != 0 || (size_t) w->chunk_off[C] != N) wide.store(1);   // (a window outside every chunk: take the plain path)
        if (!wide.load()) {
            cw.d_packed = static_cast<uint32_t*>(ctx_alloc(ctx, N * 4));
            hipError_t e1 = cw.d_packed ? hipSuccess : hipErrorOutOfMemory;
            if (e1 == hipSuccess) e1 = hipMemcpyAsync(cw.d_packed, cw.P0, N * 4, hipMemcpyHostToDevice, nullptr);   // (k_setup follows on the same stream; P0 is not written again)
            if (e1 != hipSuccess) { (void) hipGetLastError(); cw.d_packed = nullptr; }
        }
    }
    if (!cw.d_packed) {
        CTRY(dev_upload(ctx, cw.stg, &cw.d_cov, w->cov, N)); CTRY(dev_upload(ctx, cw.stg, &cw.d_mapq, w->mapq, N)); CTRY(dev_upload(ctx, cw.stg, &cw.d_clip, w->clip, N));
        CTRY(dev_upload(ctx, cw.stg, &cw.d_annot, w->annot, N));
    }
    CTRY(dev_upload(ctx, cw.stg, &cw.d_cs, w->chunk_s, C)); CTRY(dev_upload(ctx, cw.stg, &cw.d_ce, w->chunk_e, C)); CTRY(dev_upload(ctx, cw.stg, &cw.d_cl, w->chunk_ctg_len, C));
    cw.phase("window arrays up");
    return HF_OK;
}

// ------------------------------------------------------------------------------------------
// 2. the context's device arrays, its pinned result block, and the set-up kernels (no wait: the host computed every record itself)
// ------------------------------------------------------------------------------------------
int create_device_store(CreateWork& cw) {
    hf_ctx* ctx = cw.ctx; const hf_windows* w = cw.w;
    const size_t N = cw.N, C = cw.C;
    const int n_regions = cw.n_regions;
    CALLOC(ctx->d_rec, N * 4); CALLOC(ctx->d_beta, N * 8); CALLOC(ctx->d_regmask, C * 8);
    if (cw.algo == HF_ALGO_SEQ) CALLOC(ctx->d_E, N * 16 * 8);
    if (cw.algo == HF_ALGO_SEQ) CALLOC(ctx->d_scale, N * 8);   // window-order scales: the sequential cross-check only (hf_seg.h keeps them by slot)
    CALLOC(ctx->d_label, N + 16);
    CALLOC(ctx->d_chunk_stats, C * (size_t) ctx->V * 8); CALLOC(ctx->d_total, ((size_t) ctx->V + 1) * 8);
    CALLOC(ctx->d_flags, 4);
    CALLOC(ctx->d_done, HF_DONE_BYTES);         // tickets and scratch of the in-launch hand-offs (hf_rows.h)
    hipMemsetAsync(ctx->d_done, 0, HF_DONE_BYTES, nullptr);
    CALLOC(ctx->d_cks, 8);
    hipMemsetAsync(ctx->d_cks, 0, 8, nullptr);
    ctx->params_bytes = sizeof(DevParams) + (size_t) (n_regions - 1) * sizeof(DevRegion);
    CALLOC(ctx->d_params, ctx->params_bytes);
    hipMemsetAsync(ctx->d_params, 0, ctx->params_bytes, nullptr);   // (the kernel-argument path writes the bytes in use only)
    {   // one pinned block: the result vector (+ flag word, stamp, checksums) | the flag word of hf_check | the parameter block
        const size_t tot_bytes = (((size_t) ctx->V + 2 + HF_MAXREGIONS) * 8 + 63) / 64 * 64;   // (what follows stays 64-byte aligned)
        const size_t par_bytes = (ctx->params_bytes + 63) / 64 * 64;
        // (from the process-wide cache: pinning even this small block was 0.1-0.2 ms of hf_create; zeroed — a stale stamp word must not look like a future one)
        char* pin = pin_cache().acquire(tot_bytes + 64 + par_bytes);
        if (!pin) return set_err(HF_E_HIP, "hf_create: out of host memory");
        std::memset(pin, 0, tot_bytes + 64 + par_bytes);
        ctx->h_total = reinterpret_cast<double*>(pin);
        ctx->h_flags = reinterpret_cast<unsigned*>(pin + tot_bytes);
        ctx->h_params = reinterpret_cast<DevParams*>(pin + tot_bytes + 64);
    }
    {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, ctx->h_total, 0) == hipSuccess) ctx->d_total_host = (double*) dp;
        else {
            // (the cache falls back to pageable memory when pinning fails — a block the device cannot address: no zero-copy totals,
            // no stream stamp into it; passes then complete through a copy and hipStreamSynchronize)
            (void) hipGetLastError();
            ctx->d_total_host = nullptr; ctx->stream_stamp_ok = false;
        }
    }
    cw.phase("device + pinned allocations");
    hipEventCreate(&ctx->ev0); hipEventCreate(&ctx->ev1);
    hipMemsetAsync(ctx->d_flags, 0, 4, nullptr);
    hipMemsetAsync(ctx->d_label, 0xff, N ? N : 1, nullptr);
    if (N > 0 && C > 0) {
        dim3 grid((unsigned) ((cw.maxT + 255) / 256), (unsigned) C);
        hipLaunchKernelGGL(k_setup, grid, dim3(256), 0, 0, ctx->d_off, cw.d_packed, cw.d_cov, cw.d_mapq, cw.d_clip, cw.d_annot, cw.d_cs, cw.d_ce, cw.d_cl,
                           w->window_len, w->mean_read_len, w->adjust_contig_ends, w->min_read_frac,
                           w->max_high_mapq_ratio, w->min_high_mapq_ratio, w->min_highly_clipped_ratio, n_regions,
                           ctx->beta_star, ctx->d_rec, ctx->d_beta, ctx->d_flags);
        hipLaunchKernelGGL(k_regmask, dim3((unsigned) C), dim3(256), 0, 0, ctx->d_off, ctx->d_rec, ctx->d_regmask);
    }
    cw.phase("events, memsets, k_setup enqueued");
    // (no wait for k_setup: a region index out of range has been refused already; the device's own flag word is read behind the ONE
    // synchronisation at the end of hf_create)
    if (N && std::getenv("HF_CREATE_VERIFY")) {   // the test suite: the device's records against the host's, bit for bit
        std::vector<uint32_t> dev(N);
        bool same = hipMemcpy(dev.data(), ctx->d_rec, N * 4, hipMemcpyDeviceToHost) == hipSuccess;
        for (size_t c = 0; c < C && same; c++) {
            const size_t t0 = (size_t) w->chunk_off[c], n = (size_t) (w->chunk_off[c + 1] - w->chunk_off[c]);
            same = std::memcmp(dev.data() + t0, cw.hrec + t0, n * 4) == 0;
        }
        if (!same) return set_err(HF_E_HIP, "hf_create: host and device disagree on a window's packed record");
    }
    cw.phase("flags back");
    return HF_OK;
}

// ------------------------------------------------------------------------------------------
// 3. slow windows (chunk-first, or beta differs from beta_star) and the emission keys — bits of the packed records and the key marks of the
// first pass, nothing comes back from the device —, the per-pass row tables, the tiles of the per-chunk statistics
// ------------------------------------------------------------------------------------------
int create_key_tables(CreateWork& cw) {
    hf_ctx* ctx = cw.ctx; const hf_windows* w = cw.w;
    const size_t C = cw.C;
    const int n_regions = cw.n_regions, max_comps = cw.max_comps;
    if (cw.N == 0) ctx->M = 1;
    const size_t MM = (size_t) ctx->M * ctx->M;
    cw.soff.assign(C + 1, 0);
    for (size_t c = 0; c < C; c++) cw.soff[c + 1] = cw.soff[c] + cw.nslow[c];
    cw.slow.reserve((size_t) cw.soff[C]);
    for (auto& v : cw.part_slow) cw.slow.insert(cw.slow.end(), v.begin(), v.end());   // parts are consecutive chunk ranges: ascending
    for (size_t reg = 0; reg < (size_t) n_regions; reg++)                              // keys in (region, x, x_prev) order
        for (size_t x = 0; x < (size_t) ctx->M; x++)
            for (size_t xp = 0; xp < (size_t) ctx->M; xp++)
                if (cw.key_seen((reg << 16) | (x << 8) | xp)) cw.keys.push_back((int32_t) ((reg * ctx->M + x) * ctx->M + xp));
    ctx->n_slow = (int) cw.slow.size();
    ctx->n_keys = (int) cw.keys.size();
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_slow_w, cw.slow.data(), cw.slow.size()));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_slow_off, cw.soff.data(), cw.soff.size()));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_keys, cw.keys.data(), cw.keys.size()));
    // one buffer per table: rows of the (region, x, x_prev) keys first, then the private rows of the slow windows
    // (a row index fits 32 bits; + 1 row of padding)
    ctx->n_lut = (int64_t) n_regions * (int64_t) MM;
    CALLOC(ctx->d_lutE, ((size_t) ctx->n_lut + cw.slow.size() + 1) * 16 * 8);
    CALLOC(ctx->d_lutC, ((size_t) ctx->n_lut + cw.slow.size() + 1) * 4 * (size_t) max_comps * 8);
    ctx->d_Es = ctx->d_lutE + (size_t) ctx->n_lut * 16;
    ctx->d_Cs = ctx->d_lutC + (size_t) ctx->n_lut * 4 * (size_t) max_comps;
    cw.phase("contig-end list, keys, row tables");
    // tiles of 64*HF_SCAN_L windows, enumerated chunk by chunk
    std::vector<int32_t> ctile0(C + 1, 0);
    std::vector<TileDesc> desc;
    const int64_t TW = 64 * HF_SCAN_L;
    size_t si = 0;
    for (size_t c = 0; c < C; c++) {
        const int64_t t0 = w->chunk_off[c], T = w->chunk_off[c + 1] - t0;
        ctile0[c] = (int32_t) desc.size();
        for (int64_t b = 0; b < T; b += TW) {
            while (si < cw.slow.size() && cw.slow[si] < t0 + b) si++;
            TileDesc d;
            d.t0 = t0; d.T = (int) T; d.base = (int) b; d.chunk = (int) c; d.slow0 = (int) si;
            desc.push_back(d);
        }
    }
    ctile0[C] = (int32_t) desc.size();
    ctx->ntiles = (int) desc.size();
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_tile_desc, desc.data(), desc.size()));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_chunk_tile0, ctile0.data(), ctile0.size()));
    const size_t nt = (size_t) ctx->ntiles;
    CALLOC(ctx->d_tile_ll, nt * 8);
    if (cw.algo == HF_ALGO_SEQ) {   // f, b tile-major / lane-minor (hf_device.h fb_slot): only the sequential cross-check keeps them
        CALLOC(ctx->d_f, nt * 64 * HF_SCAN_L * 4 * 8); CALLOC(ctx->d_b, nt * 64 * HF_SCAN_L * 4 * 8);
    }
    ctx->h_off.assign(w->chunk_off, w->chunk_off + C + 1);
    ctx->h_tile0 = ctile0;
    CALLOC(ctx->d_tile_stats, nt * (size_t) n_regions * (16 + 9 + 2 + 3 * 16 + 1) * 8);
    cw.phase("tiles + work arrays");
    return HF_OK;
}

// ------------------------------------------------------------------------------------------
// 4. rows of A = T∘e: the (key, transition class) pairs that occur, then the slow windows; every window's row (second pass over the
// windows) with the pairs per (sub-pass, row of A); the sub-passes; the segments of the workgroup-per-segment forward-backward (hf_seg.h)
// ------------------------------------------------------------------------------------------
int create_rows_and_segments(CreateWork& cw) {
    hf_ctx* ctx = cw.ctx; const hf_windows* w = cw.w;
    const size_t N = cw.N, C = cw.C;
    const int n_regions = cw.n_regions;
    cw.pcnt.assign(HF_PARTS, {});
    cw.sub_of.assign(C, 0);
    if (cw.arena.th.joinable()) cw.arena.th.join();
    if (N > 0 && !cw.arena.b) return set_err(HF_E_HIP, "hf_create: out of host memory");
    cw.P1 = reinterpret_cast<int32_t*>(cw.arena.b); cw.P2 = reinterpret_cast<int32_t*>(cw.arena.b + N * 4);
    cw.h_arow = cw.P1;
    if (!(N > 0 && C > 0 && N < (size_t) INT32_MAX / 2)) return HF_OK;
    constexpr int64_t NL = 64;
    static_assert(HF_SEG_SPLIT <= 64 * HF_SEG_LMAX, "a segment has at most HF_SEG_LMAX windows per lane");
    // (shorter segments for small inputs were measured in rounds 3 and 5, 128..384 windows at 0.19 M .. 1.5 M windows: a workgroup's life is
    // mostly the scans and the carried-in chains — 384 windows gain 6 % at 1/8 of configs[2] and lose at 1/4 and above, profiles/r05_split_sweep.txt)
    constexpr int64_t SMAX = HF_SEG_SPLIT;
    {
        constexpr int NC = HF_AROW_CLASSES;
        auto cls_of = [](uint32_t r) { return REC_REGCHG(r) ? 8 : (int) REC_VMASK(r); };
        std::vector<int32_t> combo_id((size_t) ctx->n_lut * NC, 0);
        int32_t* cid = combo_id.data();
        // (region and coverage from the packed upload words of the first pass when there are any — 4 bytes per window instead of the
        // 10 of annot + cov: these passes run at the speed of the host's memory)
        const uint32_t* const pk = cw.d_packed ? cw.P0 : nullptr;
        auto key_of = [&](size_t t) {
            if (pk) return ((size_t) (pk[t] >> 24) * ctx->M + (pk[t] & 0xffu)) * ctx->M + (pk[t - 1] & 0xffu);
            const size_t reg = (size_t) ((w->annot[t] & 0xFC00000000000000ULL) >> 58);
            return (reg * ctx->M + (w->cov[t] & 0xffu)) * ctx->M + (w->cov[t - 1] & 0xffu);
        };
        std::vector<int32_t>& a_src = cw.h_arow_src;
        std::vector<int32_t> a_cls;
        // the (key, class) pairs the first pass marked, numbered in (region, x, x_prev, class) order
        for (size_t reg = 0; reg < (size_t) n_regions; reg++)
            for (size_t x = 0; x < (size_t) ctx->M; x++)
                for (size_t xp = 0; xp < (size_t) ctx->M; xp++) {
                    const size_t k256 = (reg << 16) | (x << 8) | xp;
                    if (!cw.key_seen(k256)) continue;
                    const size_t key = (reg * ctx->M + x) * ctx->M + xp;
                    for (size_t cl = 0; cl < (size_t) NC; cl++)
                        if (cw.key_class(k256, cl)) {
                            cid[key * NC + cl] = (int32_t) a_src.size() + 1;            // id + 1
                            a_src.push_back((int32_t) key);
                            a_cls.push_back((int32_t) cl | (int32_t) (reg << 8));
                        }
                }
        const int32_t n_combo = (int32_t) a_src.size();
        a_src.resize((size_t) n_combo + cw.slow.size()); a_cls.resize((size_t) n_combo + cw.slow.size());
        const size_t n_ar_all = a_src.size();
        int32_t* const arow = cw.h_arow;
        cw.phase("(key, class) list");
        {   // sub-passes (hf_ctx::SubPass): whole chunks, about equal window counts, <= ~1.6 M windows (~140 MB of records) each
            int S = 1;
            if ((int64_t) N > 2800000) S = (int) (((int64_t) N + 1599999) / 1600000);   // (one launch wins up to ~2.8 M windows, sub-passes of ~1.5 M beyond: profiles/r05_subpass_count2.txt)
            if (const char* e = std::getenv("HF_SUBPASSES")) { const int v = std::atoi(e); if (v >= 1 && v <= 256) S = v; }   // tests, A/B runs
            if ((size_t) S > C) S = (int) C;
            if (S < 1) S = 1;
            cw.n_sub = S;
            const int64_t total = w->chunk_off[C] - w->chunk_off[0];
            for (size_t c = 0; c < C; c++) {
                const int64_t mid = (w->chunk_off[c] + w->chunk_off[c + 1]) / 2 - w->chunk_off[0];
                const int sidx = total > 0 ? (int) (mid * S / total) : 0;
                cw.sub_of[c] = sidx < 0 ? 0 : (sidx >= S ? S - 1 : sidx);
            }
            for (size_t c = 1; c < C; c++) if (cw.sub_of[c] < cw.sub_of[c - 1]) cw.sub_of[c] = cw.sub_of[c - 1];   // (monotone: contiguous chunk ranges)
        }
        // second pass: every window's row of A, and the pairs per (sub-pass, row of A) (x >= 2: hmm.c:638-642) as one histogram per
        // part of the chunk list (popular rows: no contended atomics; few parts when the histogram is long)
        const uint32_t* const hrec = cw.hrec;
        const int n_sub = cw.n_sub;
        cw.plan_ppt = n_ar_all * (size_t) n_sub > 65536 ? 1 : 4;      // parts per thread of the second AND third pass (the same parts: pcnt)
        par_chunks(w->chunk_off, C, [&](size_t c0, size_t c1, size_t part) {
            std::vector<int32_t>& h = cw.pcnt[part];
            h.assign(n_ar_all * (size_t) n_sub, 0);
            for (size_t c = c0; c < c1; c++) {
                const int64_t t0 = w->chunk_off[c], T = w->chunk_off[c + 1] - t0;
                int32_t sp = cw.soff[c];
                for (int64_t x = 0; x < T; x++) {
                    const size_t t = (size_t) (t0 + x);
                    const uint32_t r = hrec[t];
                    int32_t id;
                    if (REC_SLOW(r)) {
                        id = n_combo + sp;
                        a_src[(size_t) id] = (int32_t) (ctx->n_lut + sp);
                        a_cls[(size_t) id] = (x == 0 ? 9 : cls_of(r)) | (int32_t) (REC_REGION(r) << 8);
                        arow[t] = id | (x == 0 ? (int32_t) 0x80000000 : 0);
                        sp++;
                    } else arow[t] = id = cid[key_of(t) * NC + cls_of(r)] - 1;
                    if (x >= 2) h[(size_t) cw.sub_of[c] * n_ar_all + (size_t) id]++;
                }
            }
        }, cw.plan_ppt);
        cw.phase("rows of A (second pass)");
        ctx->n_combo = n_combo; ctx->n_arows = (int) a_src.size();
        if (a_src.size() >= ((size_t) 1 << 25))     // the segment kernels address a row by a 32-bit BYTE offset (index << 7)
            return set_err(HF_E_ARG, "hf_create: more than 2^25 distinct rows of A (contig-end windows included): shard the chunk list (hmm_flagger_multi.h)");
        CALLOC(ctx->d_arow, N * 4);
        if (hipMemcpyAsync(ctx->d_arow, arow, N * 4, hipMemcpyHostToDevice, nullptr) != hipSuccess) return set_err(HF_E_HIP, "hf_create: rows of A up");   // (P1: pinned, not written again)
        CTRY(dev_upload(ctx, cw.stg, &ctx->d_arow_src, a_src.data(), a_src.size()));
        CTRY(dev_upload(ctx, cw.stg, &ctx->d_arow_cls, a_cls.data(), a_cls.size()));
        CALLOC(ctx->d_lutA, (a_src.size() + 1) * 16 * 8);
        {   // one row behind the rows of A: the IDENTITY (hf_seg.h: what a lane multiplies by past its last window); no kernel writes it
            double ident[16];
            for (int k = 0; k < 16; k++) ident[k] = (k % 5 == 0) ? 1.0 : 0.0;
            // (through the staging buffer: a synchronous copy here waited for the 6 MB of rows of A enqueued just above — 0.25 ms)
            hipError_t ei;
            if (cw.stg.p && cw.stg.used + 128 <= cw.stg.cap) {
                std::memcpy(cw.stg.p + cw.stg.used, ident, sizeof ident);
                ei = hipMemcpyAsync(ctx->d_lutA + a_src.size() * 16, cw.stg.p + cw.stg.used, sizeof ident, hipMemcpyHostToDevice, nullptr);
                cw.stg.used += 128;
            } else ei = hipMemcpy(ctx->d_lutA + a_src.size() * 16, ident, sizeof ident, hipMemcpyHostToDevice);
            if (ei != hipSuccess) return set_err(HF_E_HIP, "hf_create: the identity row");
        }
        if (cw.ctrace) std::fprintf(stderr, "[hf_create] %d emission keys, %d (key, transition class) rows, %d slow windows\n",
                                    ctx->n_keys, n_combo, ctx->n_slow);
    }
    // a chunk of T windows is cut into ceil(T / SMAX) equal segments; window w of a segment (w = j*L + i) lives in slot slot0 + i*64 + j
    std::vector<SegDesc>& segs = ctx->h_segs;
    std::vector<int32_t>& cseg0 = cw.cseg0;
    cseg0.assign(C + 1, 0);
    int64_t nslots = 0;
    for (size_t c = 0; c < C; c++) {
        const int64_t t0 = w->chunk_off[c], T = w->chunk_off[c + 1] - t0;
        cseg0[c] = (int32_t) segs.size();
        if (T <= 0) continue;
        const int64_t nsg = (T + SMAX - 1) / SMAX, sz = (T + nsg - 1) / nsg;
        const int first = (int) segs.size();
        for (int64_t k = 0; k * sz < T; k++) {
            SegDesc d;
            std::memset(&d, 0, sizeof d);
            const int64_t w0 = k * sz, n = T - w0 < sz ? T - w0 : sz;
            d.t0 = t0 + w0; d.n = (int) n; d.L = (int) ((n + NL - 1) / NL);
            d.slot0 = (int32_t) nslots; nslots += (int64_t) d.L * NL;
            d.chunk_slow0 = ctx->n_combo + cw.soff[c];               // the A row of the chunk's first window
            d.seg0 = first; d.k = (int) k; d.chunk = (int) c; d.ident_row = ctx->n_arows;
            d.reg_first = (int32_t) ((w->annot[t0] & 0xFC00000000000000ULL) >> 58);
            d.reg_last = (int32_t) ((w->annot[t0 + T - 1] & 0xFC00000000000000ULL) >> 58);
            segs.push_back(d);
        }
        const int nsc = (int) segs.size() - first;
        const int32_t spare = (int32_t) nslots++;          // takes f of the chunk's last window
        for (int k = 0; k < nsc; k++) {
            segs[(size_t) (first + k)].nseg = nsc;
            segs[(size_t) (first + k)].next_slot = k + 1 < nsc ? segs[(size_t) (first + k + 1)].slot0 : spare;
        }
    }
    cseg0[C] = (int32_t) segs.size();
    ctx->h_cseg0 = cseg0;
    if (nslots < INT32_MAX) {
        ctx->nseg = (int) segs.size(); ctx->n_slots = nslots;
        CTRY(dev_upload(ctx, cw.stg, &ctx->d_chunk_seg0, cseg0.data(), cseg0.size()));   // (the descriptors go up with the plan's positions in them)
        CALLOC(ctx->d_seg_ready, segs.size() * 4);
        hipMemsetAsync(ctx->d_seg_ready, 0, segs.size() * 4, nullptr);
        CALLOC(ctx->d_seg_ll, segs.size() * 8);
        CALLOC(ctx->d_Pseg, segs.size() * 16 * 8);
    } else segs.clear();
    cw.phase("segments");
    return HF_OK;
}

// ------------------------------------------------------------------------------------------
// 5. plan of the statistics by emission row (hf_rows.h) and the POSITION of every window's pair record.
// Pairs (x-1, x), x = 2..T-1 of every chunk (hmm.c:638-642), grouped by the row of A of window x (emission key x transition class, or a
// contig-end window's own row): up to 64 pairs per group, record of the group's i-th pair at position group*64 + i — k_pair_sums STREAMS
// the records, k_seg_fb scatters them (whole 64-byte records).  Windows without a pair (the first two of a chunk) and the f of a chunk's
// last window get positions after the groups.
// ------------------------------------------------------------------------------------------
struct StatsPlan {            // what the plan's two halves hand to each other
    bool compact = false, planned = false;
    int64_t n_pos = 0;
    size_t np = 0;                                        // pairs
    std::vector<int32_t> cnt;                             // pairs per (sub-pass, row of A)
    std::vector<int32_t> g_pos0;                          // position of the first pair of every (sub-pass, row of A)
    std::vector<int32_t> grp_ar, grp_n, grp_off;
    std::vector<RowSlot> rslots; std::vector<int32_t> rwreg, rwoff;
    std::vector<int64_t> extra0;                          // first of the chunk's positions behind its sub-pass's groups: its windows 0, 1 and its spare record
    std::vector<int32_t> h_spare;
};
// extras and spare records of sub-pass sp from position `from` on: windows 0, 1 of its chunks (with a plan) and the chunk's spare record,
// then one spare record per segment
int64_t plan_place_extras(CreateWork& cw, StatsPlan& pl, size_t sp, int64_t from, bool with_first_two) {
    hf_ctx* ctx = cw.ctx; const hf_windows* w = cw.w;
    int64_t p = from;
    const auto& sb = ctx->subs[sp];
    for (int c = sb.c0; c < sb.c1; c++) {
        const int64_t T = w->chunk_off[c + 1] - w->chunk_off[c];
        pl.extra0[(size_t) c] = p;
        p += T <= 0 ? 0 : (with_first_two ? (T < 2 ? T : 2) : 0) + 1;
        pl.extra0[(size_t) c + 1] = p;      // (read as "end of chunk c" below: the chunks of a sub-pass are consecutive)
    }
    for (int k = sb.seg0; k < sb.seg1; k++) ctx->h_segs[(size_t) k].trash_pos = (int32_t) p++;
    return p;
}

// groups and row slots by INDEX (a push_back per group cost 0.5 ms): first the bases — the groups and their positions SUB-PASS BY
// SUB-PASS (a sub-pass's records are contiguous: [groups | windows 0, 1 and spare record of its chunks | one spare record per segment]),
// then the row slots REGION BY REGION (k_row_stats' blocks belong to one region): within a (region, sub-pass) every run of rows of one
// EMISSION row (its transition classes are adjacent) fills slots of up to HF_ROWSLOT_GROUPS consecutive groups — then the arrays are sized
// once and filled by plain stores.
void plan_groups_and_slots(CreateWork& cw, StatsPlan& pl, int bpw) {
    hf_ctx* ctx = cw.ctx; const hf_windows* w = cw.w;
    const int n_regions = cw.n_regions;
    const size_t n_ar = cw.h_arow_src.size(), S = (size_t) cw.n_sub;
    const bool compact = pl.compact;
    std::vector<hf_ctx::SubPass>& subs = ctx->subs;
    // rows of A that occur, ordered by (region, row of A): combos are numbered by (emission key, class), keys are region-major; the
    // contig-end windows' rows follow in window order
    struct Occ { int32_t region, ar; };
    std::vector<Occ> occ;
    const size_t MMr = (size_t) ctx->M * ctx->M;
    for (size_t r = 0; r < n_ar; r++) {
        bool any = false;
        for (size_t sp = 0; sp < S && !any; sp++) any = pl.cnt[sp * n_ar + r] != 0;
        if (!any) continue;
        const int64_t er = cw.h_arow_src[r];            // emission row: a key, or n_lut + slow index
        int32_t reg;
        if (er < ctx->n_lut) reg = (int32_t) ((size_t) er / MMr);
        else reg = (int32_t) ((w->annot[(size_t) cw.slow[(size_t) (er - ctx->n_lut)]] & 0xFC00000000000000ULL) >> 58);
        occ.push_back({reg, (int32_t) r});
    }
    if (n_regions > 1) std::stable_sort(occ.begin(), occ.end(), [](const Occ& a, const Occ& b) { return a.region < b.region; });
    const size_t nocc = occ.size();
    pl.g_pos0.assign(S * n_ar, 0);
    const size_t unit = (size_t) 16 * (size_t) bpw;     // row slots per wavefront of k_row_stats
    struct RowBase { int32_t g0, slot0, go; int64_t p0; };   // first group, first slot of the row's run, groups of the run before this row, first position
    std::vector<RowBase> rb(S * nocc, RowBase{0, 0, 0, 0});
    size_t n_grp = 0, n_slot = 0;
    int64_t next_pos = 0;
    for (size_t sp = 0; sp < S; sp++) {
        subs[sp].g0 = (int) n_grp; subs[sp].p0 = next_pos;
        for (size_t oi = 0; oi < nocc; oi++) {
            const size_t r = (size_t) occ[oi].ar;
            const int32_t c_ = pl.cnt[sp * n_ar + r];
            if (!c_) continue;
            RowBase& B = rb[sp * nocc + oi];
            B.g0 = (int32_t) n_grp;
            B.p0 = compact ? next_pos : subs[sp].p0 + (int64_t) (n_grp - (size_t) subs[sp].g0) * HF_GRP_PAIRS;
            n_grp += (size_t) ((c_ + HF_GRP_PAIRS - 1) / HF_GRP_PAIRS);
            if (compact) next_pos += c_;
        }
        if (!compact) next_pos = subs[sp].p0 + (int64_t) (n_grp - (size_t) subs[sp].g0) * HF_GRP_PAIRS;
        subs[sp].g1 = (int) n_grp;
        next_pos = plan_place_extras(cw, pl, sp, next_pos, true);
        subs[sp].p1 = next_pos;
    }
    pl.n_pos = next_pos;
    {
        size_t oi0 = 0;
        for (int reg = 0; reg < n_regions; reg++) {
            pl.rwoff[(size_t) reg] = (int32_t) (n_slot / unit);
            size_t oi1 = oi0;
            while (oi1 < nocc && occ[oi1].region == reg) oi1++;
            for (size_t sp = 0; sp < S; sp++) {
                int64_t open_row = -1; size_t run_slot0 = n_slot; int32_t run_groups = 0;
                for (size_t oi = oi0; oi < oi1; oi++) {
                    const size_t r = (size_t) occ[oi].ar;
                    const int32_t c_ = pl.cnt[sp * n_ar + r];
                    if (!c_) continue;
                    const int64_t er = cw.h_arow_src[r];
                    const int32_t ng = (c_ + HF_GRP_PAIRS - 1) / HF_GRP_PAIRS;
                    if (er != open_row) { n_slot = run_slot0 + (size_t) ((run_groups + HF_ROWSLOT_GROUPS - 1) / HF_ROWSLOT_GROUPS); run_slot0 = n_slot; run_groups = 0; open_row = er; }
                    rb[sp * nocc + oi].slot0 = (int32_t) run_slot0; rb[sp * nocc + oi].go = run_groups;
                    run_groups += ng;
                }
                n_slot = run_slot0 + (size_t) ((run_groups + HF_ROWSLOT_GROUPS - 1) / HF_ROWSLOT_GROUPS);
            }
            n_slot = (n_slot + HF_RS_WPB * unit - 1) / (HF_RS_WPB * unit) * (HF_RS_WPB * unit);   // whole blocks of k_row_stats per region
            for (size_t k = (size_t) pl.rwoff[(size_t) reg]; k < n_slot / unit; k++) pl.rwreg.push_back(reg);
            oi0 = oi1;
        }
    }
    pl.grp_ar.assign((n_grp + 3) / 4 * 4, 0); pl.grp_n.assign((n_grp + 3) / 4 * 4, 0);   // (k_pair_sums: four groups per wavefront)
    pl.grp_off.assign(n_grp + 1, 0);
    pl.rslots.assign(n_slot, RowSlot{-1, 0, 0, 0});
    for (size_t sp = 0; sp < S; sp++)
        for (size_t oi = 0; oi < nocc; oi++) {
            const size_t r = (size_t) occ[oi].ar;
            const int32_t c_ = pl.cnt[sp * n_ar + r];
            if (!c_) continue;
            const int64_t er = cw.h_arow_src[r];
            const RowBase B = rb[sp * nocc + oi];
            pl.g_pos0[sp * n_ar + r] = (int32_t) B.p0;
            int32_t xpx;
            if (er < ctx->n_lut) xpx = (int32_t) (((size_t) er / (size_t) ctx->M) % (size_t) ctx->M) | ((int32_t) ((size_t) er % (size_t) ctx->M) << 8);
            else { const size_t t = (size_t) cw.slow[(size_t) (er - ctx->n_lut)]; xpx = (int32_t) (w->cov[t] & 0xffu) | ((int32_t) (w->cov[t - 1] & 0xffu) << 8); }
            int64_t p = B.p0;
            for (int32_t j = 0, left = c_; left > 0; j++, left -= HF_GRP_PAIRS) {
                const size_t g = (size_t) (B.g0 + j);
                const int32_t here = left < HF_GRP_PAIRS ? left : HF_GRP_PAIRS;
                pl.grp_ar[g] = (int32_t) r; pl.grp_n[g] = here; pl.grp_off[g] = (int32_t) p;
                p += compact ? here : HF_GRP_PAIRS;
                RowSlot& sl = pl.rslots[(size_t) B.slot0 + (size_t) ((B.go + j) / HF_ROWSLOT_GROUPS)];
                if ((B.go + j) % HF_ROWSLOT_GROUPS == 0) { sl.row = (int32_t) er; sl.g0 = (int32_t) g; sl.ng = 1; sl.xpx = xpx; }
                else sl.ng++;
            }
        }
    pl.rwoff[(size_t) n_regions] = (int32_t) (pl.rslots.size() / unit);
    ctx->n_parts = 1;
    for (int reg = 0; reg < n_regions; reg++) if (pl.rwoff[(size_t) reg + 1] > pl.rwoff[(size_t) reg]) ctx->n_parts++;
    ctx->n_groups = (int) n_grp;
    pl.grp_off[n_grp] = (int32_t) (n_grp ? pl.grp_off[n_grp - 1] + (compact ? pl.grp_n[n_grp - 1] : HF_GRP_PAIRS) : 0);
    ctx->plan_compact = compact;
    pl.planned = true;
}

// the plan's device side: the negative-binomial count-data bins, groups, row slots, work arrays
int plan_upload(CreateWork& cw, StatsPlan& pl) {
    hf_ctx* ctx = cw.ctx;
    const int n_regions = cw.n_regions;
    const size_t C = cw.C;
    {   // negative_binomial count data: the row slots of every (region, min(x, 249)) bin, in plan order
        std::vector<int32_t> boff((size_t) n_regions * 256 + 1, 0), blist;
        auto bin_of = [&](size_t k) {
            const int x = pl.rslots[k].xpx & 0xff;
            return (size_t) pl.rwreg[k / ((size_t) 16 * (size_t) ctx->rs_bpw)] * 256 + (size_t) (x < HF_NB_MAX_COVERAGE ? x : HF_NB_MAX_COVERAGE - 1);
        };
        for (size_t k = 0; k < pl.rslots.size(); k++) if (pl.rslots[k].row >= 0) boff[bin_of(k) + 1]++;
        for (size_t b = 0; b + 1 < boff.size(); b++) boff[b + 1] += boff[b];
        blist.resize((size_t) boff.back());
        std::vector<int32_t> fill(boff.begin(), boff.end() - 1);
        for (size_t k = 0; k < pl.rslots.size(); k++) if (pl.rslots[k].row >= 0) blist[(size_t) fill[bin_of(k)]++] = (int32_t) k;
        CTRY(dev_upload(ctx, cw.stg, &ctx->d_bin_off, boff.data(), boff.size()));
        CTRY(dev_upload(ctx, cw.stg, &ctx->d_bin_list, blist.data(), blist.size()));
        CALLOC(ctx->d_slot_h, pl.rslots.size() * 4 * 8);
        CALLOC(ctx->d_H, (size_t) n_regions * 4 * 256 * 8);
    }
    ctx->n_rowwaves = (int) (pl.rslots.size() / ((size_t) 16 * (size_t) ctx->rs_bpw));   // 16 * bpw slots per wavefront, whole blocks per region
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_grp_ar, pl.grp_ar.data(), pl.grp_ar.size()));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_grp_n, pl.grp_n.data(), pl.grp_n.size()));
    if (pl.compact) CTRY(dev_upload(ctx, cw.stg, &ctx->d_grp_off, pl.grp_off.data(), pl.grp_off.size()));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_rowslots, pl.rslots.data(), pl.rslots.size()));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_rw_region, pl.rwreg.data(), pl.rwreg.size()));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_rw_off, pl.rwoff.data(), pl.rwoff.size()));
    ctx->h_rw_off = pl.rwoff;
    CALLOC(ctx->d_grp_sums, (size_t) pl.grp_ar.size() * 16 * 8);
    CALLOC(ctx->d_chunk_ll, C * 8);
    CALLOC(ctx->d_rw_stats, (size_t) ctx->n_rowwaves * (16 + 9 + 2 + 3 * 16 + 1) * 8);
    ctx->rows_ready = true;
    return HF_OK;
}

int create_statistics_plan(CreateWork& cw) {
    hf_ctx* ctx = cw.ctx; const hf_windows* w = cw.w;
    const size_t N = cw.N, C = cw.C;
    const int n_regions = cw.n_regions;
    if (!(N > 0 && C > 0 && ctx->nseg > 0)) return HF_OK;
    const size_t n_ar = cw.h_arow_src.size();
    const size_t S = (size_t) cw.n_sub;
    StatsPlan pl;
    pl.cnt.assign(S * n_ar + 1, 0);
    pl.rwoff.assign((size_t) n_regions + 1, 0);
    pl.extra0.assign(C + 1, 0);
    pl.h_spare.assign(C, 0);
    for (size_t c = 0; c < C; c++) {
        const int64_t T = w->chunk_off[c + 1] - w->chunk_off[c];
        pl.np += (size_t) (T > 2 ? T - 2 : 0);
    }
    // pairs per (sub-pass, row of A): the sum of the parts' histograms; the per-part counts become the parts' starting ranks for the
    // positions below (exclusive prefix over the parts)
    for (auto& h : cw.pcnt)
        if (!h.empty())
            for (size_t r = 0; r < S * n_ar; r++) { const int32_t here = h[r]; h[r] = pl.cnt[r]; pl.cnt[r] += here; }
    // Two layouts of the groups' records.  PADDED: group g at positions g*64.. — k_pair_sums streams whole groups with a fixed geometry
    // (the layout of inputs whose rows are popular: BASELINE configs[2], [4]).  When most pairs sit in rows of their own (coverage spread
    // over the whole range, or reads longer than the contigs: every window a contig-end window) that would cost up to 64 positions per
    // pair; then COMPACT: the groups back to back, group g at grp_off[g] — k_pair_sums_compact, four lanes per group.
    int64_t n_groups_all = 0;
    for (size_t r = 0; r < S * n_ar; r++) n_groups_all += (pl.cnt[r] + HF_GRP_PAIRS - 1) / HF_GRP_PAIRS;
    const bool dense = n_groups_all * HF_GRP_PAIRS <= 4 * (int64_t) pl.np + (4 << 20);   // 32 MiB of slack
    pl.compact = !dense;
    const char* const force = std::getenv("HF_STATS_PLAN");     // tests: "compact" / "padded" [",bpw=N"] on inputs of any size
    if (force && std::strstr(force, "compact")) pl.compact = true;
    if (force && std::strstr(force, "padded") && n_groups_all * HF_GRP_PAIRS + 3 * (int64_t) C + 4 * HF_GRP_PAIRS < INT32_MAX) pl.compact = false;
    // sparse plans have as many row slots as pairs, give or take: a wavefront of k_row_stats then takes `bpw` batches of 16 slots before it
    // reduces — a block's hand-off (partial vector, ticket) costs as much as a batch.  Measured on config 5 (260 k slots,
    // profiles/r03k_cfg5.txt): bpw 1 / 2 / 4 / 8 / 16 = 82 / 62 / 56 / 52 / 81 us: at least ~1000 wavefronts stay
    int bpw = 1;
    while (bpw < 8 && n_groups_all / 16 / (2 * bpw) >= 1000) bpw *= 2;   // (below that a second batch per wavefront costs more than it saves: cfg-2 +4 us)
    if (force) { const char* b = std::strstr(force, "bpw="); if (b) { const int v = std::atoi(b + 4); if (v >= 1 && v <= 64) bpw = v; } }
    ctx->rs_bpw = bpw;
    cw.phase("plan: pairs");
    int32_t* const pos = cw.P2;                   // record position of every window (b half); the position of the record with its f: k_pos_f
    if (w->chunk_off[0] != 0 || (size_t) w->chunk_off[C] != N) std::memset(pos, 0, N * 4);   // windows outside every chunk
    // the sub-passes' chunk and segment ranges
    std::vector<hf_ctx::SubPass>& subs = ctx->subs;
    subs.assign(S, hf_ctx::SubPass{0, 0, 0, 0, 0, 0, 0, 0});
    for (size_t sp = 0; sp < S; sp++) { subs[sp].c0 = (int) C; subs[sp].c1 = 0; }
    for (size_t c = 0; c < C; c++) {
        hf_ctx::SubPass& sb = subs[(size_t) cw.sub_of[c]];
        if ((int) c < sb.c0) sb.c0 = (int) c;
        if ((int) c + 1 > sb.c1) sb.c1 = (int) c + 1;
    }
    for (size_t sp = 0; sp < S; sp++) {
        if (subs[sp].c1 < subs[sp].c0) { subs[sp].c0 = subs[sp].c1 = sp ? subs[sp - 1].c1 : 0; }   // (an empty sub-pass)
        subs[sp].seg0 = cw.cseg0[(size_t) subs[sp].c0]; subs[sp].seg1 = cw.cseg0[(size_t) subs[sp].c1];
    }
    const bool can_plan = pl.np > 0 && (pl.compact ? (int64_t) pl.np : n_groups_all * HF_GRP_PAIRS) + 3 * (int64_t) C + 4 * HF_GRP_PAIRS + (int64_t) ctx->nseg < INT32_MAX && N < (size_t) INT32_MAX;
    if (can_plan) {
        plan_groups_and_slots(cw, pl, bpw);
        cw.phase("plan: groups, row slots");
    } else {
        // no plan (sparse rows past the position range): the per-chunk statistics read the records by window; positions in slot order,
        // one "sub-pass" holds everything
        cw.n_sub = 1;
        subs.assign(1, hf_ctx::SubPass{0, (int) C, 0, ctx->nseg, 0, 0, 0, 0});
        std::fill(cw.sub_of.begin(), cw.sub_of.end(), 0);
        pl.n_pos = plan_place_extras(cw, pl, 0, ctx->n_slots, false);
        subs[0].p1 = pl.n_pos;
    }
    // third pass: the position of every window's record.  Pairs of a (sub-pass, row of A) in window order; the windows without a pair of
    // their own (x = 0, 1) and the f of every chunk's last window behind the sub-pass's groups, chunk by chunk
    const bool planned = pl.planned;
    par_chunks(w->chunk_off, C, [&](size_t c0, size_t c1, size_t part) {
        int32_t* const fill = cw.pcnt[part].data();      // rank of the part's next pair of every (sub-pass, row of A)
        for (size_t c = c0; c < c1; c++) {
            const int64_t t0 = w->chunk_off[c], T = w->chunk_off[c + 1] - t0;
            if (T <= 0) continue;
            const int64_t ex = pl.extra0[c];
            if (planned) {
                const size_t sb = (size_t) cw.sub_of[c] * n_ar;
                for (int64_t x = 0; x < T && x < 2; x++) pos[(size_t) (t0 + x)] = (int32_t) (ex + x);
                for (int64_t x = 2; x < T; x++) {
                    const size_t r = sb + (size_t) (cw.h_arow[(size_t) (t0 + x)] & 0x7fffffff);
                    const int64_t k = fill[r]++;
                    pos[(size_t) (t0 + x)] = pl.g_pos0[r] + (int32_t) k;
                }
            } else {
                for (int k = cw.cseg0[c]; k < cw.cseg0[c + 1]; k++) {
                    const SegDesc& d = ctx->h_segs[(size_t) k];
                    for (int64_t x = 0; x < d.n; x++) pos[(size_t) (d.t0 + x)] = seg_slot(d, x);
                }
            }
            const int32_t spare = (int32_t) (ex + (planned ? (T < 2 ? T : 2) : 0));   // takes the f of the chunk's last window
            pl.h_spare[c] = spare;
            for (int k = cw.cseg0[c]; k < cw.cseg0[c + 1]; k++) ctx->h_segs[(size_t) k].spare_pos = spare;
        }
    }, cw.plan_ppt);
    cw.phase("plan: positions (third pass)");
    if (planned) CTRY(plan_upload(cw, pl));
    CTRY(dev_upload(ctx, cw.stg, &ctx->d_seg, ctx->h_segs.data(), ctx->h_segs.size()));
    ctx->n_pos = pl.n_pos;
    CALLOC(ctx->d_pos, N * 4); CALLOC(ctx->d_pos_f, N * 4);
    if (hipMemcpyAsync(ctx->d_pos, pos, N * 4, hipMemcpyHostToDevice, nullptr) != hipSuccess) return set_err(HF_E_HIP, "hf_create: record positions up");
    {   // pos_f[t] = the position of window t + 1's record (which holds f_t), the chunk's spare record for its last window: on the device
        CTRY(dev_upload(ctx, cw.stg, &cw.d_spare, pl.h_spare.data(), C));
        if (w->chunk_off[0] != 0 || (size_t) w->chunk_off[C] != N) hipMemsetAsync(ctx->d_pos_f, 0, N * 4, nullptr);
        hipLaunchKernelGGL(k_pos_f, dim3((unsigned) ((cw.maxT + 255) / 256), (unsigned) C), dim3(256), 0, 0, ctx->d_off, ctx->d_pos, cw.d_spare, ctx->d_pos_f);
    }
    if (pl.n_pos >= INT32_MAX) return set_err(HF_E_ARG, "hf_create: more than 2^31 record positions (shard the chunk list: hmm_flagger_multi.h)");
    int64_t cap = 0;                          // positions the pass buffer holds: the largest sub-pass (everything, with one)
    for (const auto& sb : ctx->subs) if (sb.p1 - sb.p0 > cap) cap = sb.p1 - sb.p0;
    CALLOC(ctx->d_recs, (size_t) (cap + 1) * 64);
    if (ctx->subs.size() == 1) ctx->d_recs_all = ctx->d_recs;      // (several sub-passes: the all-windows buffer on first use, all_records_buffer)
    if (cw.ctrace) std::fprintf(stderr, "[hf_create] statistics plan: %s, %d groups, %d row-slot wavefronts of %d x 16 slots, %lld positions for %lld pairs; %zu sub-pass(es), record buffer %.0f MB\n",
                                !planned ? "none (per-chunk statistics)" : ctx->plan_compact ? "compact" : "padded", ctx->n_groups, ctx->n_rowwaves,
                                ctx->rs_bpw, (long long) pl.n_pos, (long long) pl.np, ctx->subs.size(), (double) (cap + 1) * 64 / 1e6);
    cw.phase("plan: uploads, allocations");
    return HF_OK;
}

// ------------------------------------------------------------------------------------------
// 6. how the segment kernel will be launched, the job list of the per-pass tables, the environment's switches
// ------------------------------------------------------------------------------------------
// XCD plan (VERDICT r05 #2b): within every sub-pass the chunks are dealt to eight lists, each chunk to the list with the fewest segments so
// far (ties: the lowest list), and list x's r-th segment runs as block b0 + 8 r + x — all segments of a chunk on blocks congruent mod 8
// (observed: block b runs on XCD b % 8), the lists within one chunk's segments of each other (padding blocks: -1, they leave at once).  A
// chunk's segments then span 8 x nseg block indices instead of nseg: the plan is only built where twice that span is resident, so the
// one-launch guard still holds for it.  MEASURED AND NOT THE DEFAULT (profiles/r06_ab_handoff.txt): on configs[2] the plan makes k_seg_fb
// 13 us SLOWER (60 us against 47, same box, three alternations), at 1/4 and 1/8 of the size it changes nothing.  Block b runs segment b, as
// in rounds 3-5; HF_SEG_XCD=1 switches the plan on.
int create_xcd_plan(CreateWork& cw, int64_t resident, int max_nseg) {
    hf_ctx* ctx = cw.ctx;
    const char* e = std::getenv("HF_SEG_XCD");
    const bool want = e && e[0] == '1';
    if (!(want && ctx->seg_fused && resident > 0 && (int64_t) max_nseg * 8 * 2 <= resident && !ctx->subs.empty())) return HF_OK;
    std::vector<int32_t>& tab = ctx->h_seg_of_block;
    tab.clear();
    for (auto& sb : ctx->subs) {
        std::vector<int32_t> lane[8];
        for (int c = sb.c0; c < sb.c1; c++) {
            const int s0 = cseg0_of(ctx, c), s1 = cseg0_of(ctx, c + 1);
            if (s1 <= s0) continue;
            int best = 0;
            for (int x = 1; x < 8; x++) if (lane[x].size() < lane[best].size()) best = x;
            for (int k = s0; k < s1; k++) lane[best].push_back(k);
        }
        size_t rows = 0;
        for (int x = 0; x < 8; x++) rows = std::max(rows, lane[x].size());
        sb.b0 = (int) tab.size();
        tab.resize(tab.size() + rows * 8, -1);
        for (int x = 0; x < 8; x++)
            for (size_t r = 0; r < lane[x].size(); r++) tab[(size_t) sb.b0 + r * 8 + (size_t) x] = lane[x][r];
        sb.b1 = (int) tab.size();
    }
    return dev_upload(ctx, cw.stg, &ctx->d_seg_of_block, tab.data(), tab.size());
}

int create_launch_config(CreateWork& cw) {
    hf_ctx* ctx = cw.ctx;
    const int device = cw.device, algo = cw.algo;
    { const char* e = std::getenv("HF_HOST_TRACE"); ctx->host_trace = e && e[0] == '1'; }
    { const char* e = std::getenv("HF_SEG_LAUNCHES"); ctx->seg_fused = !(e && e[0] == '2'); }   // HF_SEG_LAUNCHES=2: k_seg_prod + k_seg_fb
    if (ctx->seg_fused && ctx->nseg > 0) {
        // Static guard of the one-launch hand-off: the segments of a chunk wait for each other INSIDE the launch, so all
        // of them have to be resident together.  A chunk with more segments than the device holds workgroups of k_seg_fb (small
        // --windowLen with a large --chunkLen: > 1.57 M windows in one chunk on 256 CUs x 12) could only time out (65 536 polls per
        // waiting lane) and re-run every first pass in two launches: such a context starts in two-launch mode.
        int per_cu = 0, cus = 0, max_nseg = 0;
        for (const SegDesc& d : ctx->h_segs) if (d.nseg > max_nseg) max_nseg = d.nseg;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_seg_fb<true, true>, 64, seg_lds_bytes()) != hipSuccess) { (void) hipGetLastError(); per_cu = 0; }
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void) hipGetLastError(); cus = 0; }
        int64_t resident = (int64_t) per_cu * cus;
        if (const char* e = std::getenv("HF_SEG_RESIDENT")) resident = std::atoll(e);   // tests: pretend a smaller device
        if (resident > 0 && max_nseg > resident) ctx->seg_fused = false;
        CTRY(create_xcd_plan(cw, resident, max_nseg));
        // Cached row blocks (hf_seg.h, round 5): the LDS that a device with FEWER segments than it could hold leaves idle goes to the
        // workgroups — the largest nc at which every segment is still resident at once.  Decided from the occupancy the runtime reports for
        // that much dynamic LDS; HF_SEG_CACHED_STEPS forces a value (tests, A/B runs).
        auto raise_lds = [](size_t lds) {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_fb<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_fb<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) == hipSuccess;
        };
        int nc = 0;
        if (ctx->seg_fused && cus > 0) {
            for (int c = HF_SEG_LMAX; c >= 1; c--) {
                const size_t lds = seg_lds_bytes(c);
                if (lds > ctx->lds_max && lds > 64 * 1024 && !raise_lds(lds)) { (void) hipGetLastError(); continue; }
                int pc = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, k_seg_fb<true, true, true>, 64, lds) != hipSuccess) { (void) hipGetLastError(); continue; }
                int64_t res_c = (int64_t) pc * cus;
                if (std::getenv("HF_SEG_RESIDENT") && per_cu > 0) res_c = resident * pc / per_cu;     // the pretended device, scaled alike
                // 15 % of room to spare: where the segments filled the device to the last workgroup the API allows (1 531 segments at six
                // per CU) k_seg_fb took 59 us instead of 41 — one workgroup per CU fewer was actually resident, and the segments of a chunk
                // wait for each other (profiles/r05_scale_nc.txt)
                if (res_c * 85 / 100 >= ctx->nseg) { nc = c; break; }
            }
            // all of a lane's steps or none: with only some of them cached the kernel's run-time block arithmetic costs what the saved fetches
            // bring (same box, 3 of 8 steps at 1/4 of configs[2]: 29.2 us against 28.9 without; all 8 at 1/8: 25.6 against 26.5 —
            // profiles/r05g_scale_nc.txt); HF_SEG_CACHED_STEPS still forces any number
            if (nc < HF_SEG_LMAX) nc = 0;
            if (const char* e = std::getenv("HF_SEG_CACHED_STEPS")) {
                const int v = std::atoi(e);
                if (v >= 0 && v <= HF_SEG_LMAX) nc = v;
                const size_t lds = seg_lds_bytes(nc);
                if (lds > 64 * 1024 && !raise_lds(lds)) { (void) hipGetLastError(); nc = 0; }
            }
        }
        ctx->seg_nc = nc;
        if (cw.ctrace || ctx->host_trace)
            std::fprintf(stderr, "[hf_create] segment kernel: %d segments, longest chunk %d; %lld workgroups resident (%d per CU x %d CUs): %s; %d of %d row blocks cached in LDS (%zu B per workgroup)%s\n",
                         ctx->nseg, max_nseg, (long long) resident, per_cu, cus,
                         ctx->seg_fused ? "one launch" : "TWO launches (a chunk has more segments than the device holds workgroups)", nc, HF_SEG_LMAX, seg_lds_bytes(nc),
                         ctx->d_seg_of_block ? "; XCD plan" : "");
    }
    {   // the job list of the per-pass tables (hf_scan.h), built once: the (key, class) list of the rows of A when the segment kernels run,
        // the emission keys otherwise (HF_ALGO_SEQ); then the slow windows
        const bool arows = ctx->nseg > 0 && ctx->d_lutA && algo == HF_ALGO_SCAN;
        const int nk = arows ? ctx->n_combo : ctx->n_keys;
        const int jobs = nk + ctx->n_slow;
        CALLOC(ctx->d_jobs, (size_t) (jobs + 1) * sizeof(TableJob));
        if (jobs > 0) {
            hipLaunchKernelGGL(k_build_jobs, dim3((unsigned) ((jobs + 255) / 256)), dim3(256), 0, 0, nk, arows ? ctx->d_arow_src : ctx->d_keys,
                               arows ? ctx->d_arow_cls : (const int32_t*) nullptr, ctx->n_slow, ctx->d_slow_w, ctx->d_rec, ctx->d_beta, ctx->M,
                               ctx->n_lut, ctx->beta_star, ctx->d_jobs);
            if (hipGetLastError() != hipSuccess) return set_err(HF_E_HIP, "hf_create: k_build_jobs");
        }
        TabWork& tw = ctx->tabwork;
        tw.n_jobs = jobs; tw.K = ctx->K; tw.jobs = ctx->d_jobs; tw.lutE = ctx->d_lutE; tw.lutC = ctx->d_lutC; tw.lutA = arows ? ctx->d_lutA : nullptr;
    }
    ctx->seg_test_timeout = std::getenv("HF_SEG_TEST_TIMEOUT") != nullptr;
    { const char* e = std::getenv("HF_STREAM_STAMP"); if (e && e[0] == '0') ctx->stream_stamp_ok = false; }
    { const char* e = std::getenv("HF_PARAMS_COPY"); if (e && e[0] == '1') ctx->kp_ok = false; }                    // the parameter block by a copy ahead of every pass
    { const char* e = std::getenv("HF_TOTAL"); if (e && !std::strcmp(e, "device")) ctx->host_total_ok = false; }   // HF_TOTAL=device: the pass's last blocks sum the partials (rounds 3-4)
    {
        const char* e = std::getenv("HF_STATS");
        ctx->stats_mode = (e && std::strcmp(e, "chunks") == 0) ? HF_STATS_CHUNKS : HF_STATS_ROWS;
    }
#ifdef HF_SEG_TRACE
    if (ctx->nseg > 0 && std::getenv("HF_SEG_TRACE_FILE")) {
        CALLOC(ctx->d_seg_trace, (size_t) ctx->nseg * HF_SEG_TRACE_N * 8);
        hipMemsetAsync(ctx->d_seg_trace, 0, (size_t) ctx->nseg * HF_SEG_TRACE_N * 8, nullptr);
        hipMemcpyToSymbol(HIP_SYMBOL(g_seg_trace), &ctx->d_seg_trace, sizeof(void*));
    }
#endif
    cw.phase("switches, job list");
    return HF_OK;
}
#undef CALLOC
}  // namespace

extern "C" int hf_create(const hf_windows* w, int n_regions, int max_comps, int device, int algo, hf_ctx** out) {
    if (!w || !out || w->n_windows < 0 || w->n_chunks < 0 || n_regions < 1 || n_regions > HF_MAXREGIONS ||
        max_comps < 1 || max_comps > HF_MAXCOMP || (algo != HF_ALGO_SCAN && algo != HF_ALGO_SEQ))
        return set_err(HF_E_ARG, "hf_create: bad argument");
    if (hf_device_count() <= 0) return set_err(HF_E_NOGPU, "hf_create: no HIP device (there is no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    hf_ctx* ctx = new hf_ctx();
    ctx->device = device; ctx->algo = algo;
    CreateWork cw;
    cw.ctx = ctx; cw.w = w; cw.n_regions = n_regions; cw.max_comps = max_comps; cw.device = device; cw.algo = algo;
    cw.ctrace = std::getenv("HF_HOST_TRACE") != nullptr;
    cw.t_begin = cw.t_last = std::chrono::steady_clock::now();
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0) ctx->lds_max = (size_t) v;
    }
    ctx->N = w->n_windows; ctx->C = w->n_chunks; ctx->R = n_regions; ctx->K = max_comps;
    ctx->V = hf_stats_len(n_regions, max_comps);
    ctx->meta = *w;
    ctx->meta.chunk_off = nullptr; ctx->meta.cov = ctx->meta.mapq = ctx->meta.clip = nullptr;
    ctx->meta.annot = nullptr; ctx->meta.chunk_s = ctx->meta.chunk_e = ctx->meta.chunk_ctg_len = nullptr;
    for (int c = 0; c < w->n_chunks; c++) {
        const int64_t T = w->chunk_off[c + 1] - w->chunk_off[c];
        if (T < 0 || T > INT32_MAX) { delete ctx; return set_err(HF_E_ARG, "hf_create: bad chunk_off"); }
        if (T > cw.maxT) cw.maxT = (int32_t) T;
    }
    ctx->maxT = cw.maxT;
    {   // beta of every interior window (hmm.c:301-316): u - l = L - 1 there
        const int L = w->mean_read_len;
        double bs = 1.0;
        if (w->adjust_contig_ends) { bs = (double) (L - 1) / L; if (!(bs > 0.25)) bs = 0.25; }
        ctx->beta_star = bs;
    }
    cw.N = (size_t) ctx->N; cw.C = (size_t) ctx->C;
    // (the first slab holds what this function allocates — ~(29 + 2.4 R) bytes per window of arrays, ~85 per window of pair
    // records, and a context in sub-passes keeps ONE sub-pass's records, not all windows': 4 x configs[2] reserved 814 MB, now 470)
    {
        const size_t rec_windows = cw.N > 2800000 ? (size_t) 1800000 : cw.N;
        ctx->slab_first = cw.N * (size_t) (32 + 3 * n_regions) + rec_windows * 90 + ((size_t) 16 << 20);
    }
    int rc = create_pack_windows(cw);
    if (!rc) rc = create_device_store(cw);
    if (!rc) rc = create_key_tables(cw);
    if (!rc) rc = create_rows_and_segments(cw);
    if (!rc && algo == HF_ALGO_SCAN && cw.N > 0 && cw.C > 0 && ctx->nseg == 0)
        rc = set_err(HF_E_ARG, "hf_create: HF_ALGO_SCAN holds at most 2^30 windows per context (shard the chunk list: hmm_flagger_multi.h)");
    if (!rc) rc = create_statistics_plan(cw);
    if (!rc) rc = create_launch_config(cw);
    if (rc) { hf_destroy(ctx); return rc; }
    {   // the ONE synchronisation of this function: uploads and set-up kernels done, the staging buffers may go back to the cache
        const hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { hf_destroy(ctx); return set_err(HF_E_HIP, std::string("setup: ") + hipGetErrorString(e)); }
        unsigned fl = 0;
        if (cw.N > 0 && cw.C > 0) hipMemcpy(&fl, ctx->d_flags, 4, hipMemcpyDeviceToHost);
        if (fl & HF_FLAG_REGION) { hf_destroy(ctx); return set_err(HF_E_REGION, "a window's region index is >= n_regions"); }
    }
    cw.phase("uploads and set-up kernels done");
    ctx->create_phases.emplace_back("total", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - cw.t_begin).count());
    *out = ctx;
    return HF_OK;
}
