// hf_estep.hip — MI355X (gfx950) E-step of HMM-Flagger: the host side of the C ABI (include/hmm_flagger_hip.h), the
// context, the one-time set-up kernels and the launch sequence of a pass; the kernels of a pass live in the headers below.
//
// Replaces EM_runOneIterationForList / EM_runForwardForList (programs/submodules/hmm/hmm.c:739,790).
// One pass (per EM iteration), HF_ALGO_SCAN:
//   hf_scan.h    k_tables      emission rows of this iteration (per occurring (region, x, x_prev) + per contig-end window)
//                              and the rows of A_t = T_t∘e_t           (hmm_utils.c:753-793, 941-947, 2278-2292)
//   hf_seg.h     k_seg_fb      one workgroup per chunk segment, ONE launch: lane products of A_t, scans, the segments' products
//                              handed over inside the launch, scaled forward + log-likelihood, scaled backward + posterior
//                              argmax, pair records   (k_seg_prod + k_seg_fb<., false>: the two-launch fall-back)
//                                                                        (hmm.c:333-434, 452-545, 671-692)
//   hf_rows.h    k_pair_sums / k_pair_sums_compact, k_row_stats (its last blocks sum the pass's total): xi sufficient statistics summed by emission row (default)
//                                                                        (hmm.c:563-650, hmm_utils.c:812-839, 1027-1034)
//   hf_chunks.h  k_stats_tile, k_chunk_stats, k_reduce: the same statistics as one vector per chunk, summed in chunk-list
//                order (hmm.c:759-763) — HF_STATS_CHUNKS, the per-chunk multi-GPU exchange, HF_ALGO_SEQ
//   hf_nb.h, hf_nb_rows.h   the negative_binomial model's tables and count data (per chunk / by emission row)
//   hf_seq.h     HF_ALGO_SEQ, an independent on-device check: k_emit_rows (direct evaluation of every window's row),
//                k_fwd_seq / k_bwd_seq (one wavefront per chunk, sequential recurrences)
// There is no CPU fallback: without a HIP device hf_create fails with HF_E_NOGPU.
#include "hf_device.h"
#include <hip/hip_ext.h>
#include "../../include/hmm_flagger_model.h"
#ifndef HF_SCAN_L
#define HF_SCAN_L 4   // consecutive windows per lane in the scan kernels
#endif
#include <string>
#include <algorithm>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <memory>
#include <chrono>
#include <cstdlib>

#include <thread>
#include <immintrin.h>
static thread_local std::string g_err;

// hf_create's host loops over the windows are independent per chunk: run fn(first chunk, last chunk + 1, part) on up to
// HF_PARTS threads, the chunk list cut into parts of about equal window count (1.5 M windows: 6-8 ms per loop on one thread).
// The threads are a process-wide pool (hf_warmup starts it): spawning 16 threads per loop was measured at 1-2 ms a loop, as much
// as the loops themselves.  A caller that finds the pool busy (hf_multi's ranks create their contexts concurrently) spawns its own.
constexpr size_t HF_PARTS = 64;      // parts at most; host_threads() of them are used
namespace {
struct HostPool {
    std::mutex run_m, m;
    std::condition_variable cv;
    std::vector<std::thread> th;
    const std::function<void(size_t)>* job = nullptr;
    size_t gen = 0, parts = 0;
    std::atomic<size_t> next{0}, done{0}, active{0};
    bool stop = false;
    void start(size_t n) {
        std::lock_guard<std::mutex> g(m);
        while (th.size() < n) th.emplace_back([this] { worker(); });
    }
    void worker() {
        size_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* f;
            size_t P;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job; P = parts;
                if (!f) continue;                      // woke after the job was over
                active.fetch_add(1, std::memory_order_relaxed);   // under m: run() does not reuse the counters while a worker holds a job
            }
            for (size_t k; (k = next.fetch_add(1, std::memory_order_relaxed)) < P;) { (*f)(k); done.fetch_add(1, std::memory_order_release); }
            active.fetch_sub(1, std::memory_order_release);
        }
    }
    // f(k) for k = 0..P-1 on the caller and up to `workers` pool threads (parts are handed out one at a time: a thread that wakes late
    // takes fewer); false if another caller holds the pool
    bool run(size_t P, const std::function<void(size_t)>& f, size_t workers) {
        std::unique_lock<std::mutex> only(run_m, std::try_to_lock);
        if (!only.owns_lock()) return false;
        start(std::min(workers, P - 1));
        next.store(0); done.store(0);
        { std::lock_guard<std::mutex> g(m); job = &f; parts = P; gen++; }
        cv.notify_all();
        for (size_t k; (k = next.fetch_add(1, std::memory_order_relaxed)) < P;) { f(k); done.fetch_add(1, std::memory_order_release); }
        while (done.load(std::memory_order_acquire) < P) std::this_thread::yield();
        { std::lock_guard<std::mutex> g(m); job = nullptr; parts = 0; }   // a worker that wakes from here on leaves the counters alone
        while (active.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        return true;
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};
HostPool& host_pool() { static HostPool p; return p; }

// Pinned staging buffers of hf_create, kept for the life of the process (round 5): pinning costs ~0.2 ms per MB (1.2-1.3 ms of a 7 ms
// hf_create on BASELINE configs[2]) and un-pinning as much again on a helper thread.  A context takes the smallest free buffer that
// is large enough, or a new one (a free smaller one is given back first); hf_warmup pins a first set while the input loads.
struct PinCache {
    struct Slot { char* p = nullptr; size_t cap = 0; bool busy = false, pinned = false; };
    std::mutex m;
    std::vector<Slot> slots;
    char* try_acquire(size_t bytes) {                     // a cached buffer, or nullptr
        std::lock_guard<std::mutex> g(m);
        int best = -1;
        for (size_t i = 0; i < slots.size(); i++)
            if (!slots[i].busy && slots[i].cap >= bytes && (best < 0 || slots[i].cap < slots[(size_t) best].cap)) best = (int) i;
        if (best < 0) return nullptr;
        slots[(size_t) best].busy = true;
        return slots[(size_t) best].p;
    }
    char* acquire(size_t bytes) {                         // ... or a new one (pinned when possible; nullptr: out of host memory)
        if (char* p = try_acquire(bytes)) return p;
        Slot s;
        s.cap = bytes + bytes / 8 + 4096;
        {   // make room: one free buffer that was too small goes back
            std::lock_guard<std::mutex> g(m);
            for (size_t i = 0; i < slots.size(); i++)
                if (!slots[i].busy) { if (slots[i].pinned) (void) hipHostFree(slots[i].p); else std::free(slots[i].p); slots.erase(slots.begin() + (long) i); break; }
        }
        // Portable AND coherent: with a non-zero flag word and no Coherent / Mapped bit the runtime lets HIP_HOST_COHERENT decide, and
        // its default is non-coherent — the kernels' zero-copy writes into the result block (totals, partial vectors, labels) would then be visible
        // at the end of the kernel only, and the HF_POLL stamp never while it runs.  Plain flags as the fall-back, pageable memory as the last one
        // (a context whose result block is not device-addressable completes its passes through a copy: create_device_store).
        if (hipHostMalloc((void**) &s.p, s.cap, hipHostMallocPortable | hipHostMallocCoherent) == hipSuccess) s.pinned = true;
        else if ((void) hipGetLastError(), hipHostMalloc((void**) &s.p, s.cap, hipHostMallocDefault) == hipSuccess) s.pinned = true;
        else { (void) hipGetLastError(); s.p = (char*) std::malloc(s.cap); }
        if (!s.p) return nullptr;
        s.busy = true;
        std::lock_guard<std::mutex> g(m);
        slots.push_back(s);
        return s.p;
    }
    void release(char* p) {
        if (!p) return;
        std::lock_guard<std::mutex> g(m);
        for (auto& s : slots) if (s.p == p) s.busy = false;
    }
};
PinCache& pin_cache() { static PinCache* c = new PinCache(); return *c; }   // (never destroyed: the process leaves without unwinding HIP)

// Validity masks of hmm_utils.c:2229-2254 by THRESHOLD per coverage value (every value below 256): the ratio v / (0.1 + cv) is monotone in v,
// so "Dup valid" = !(ratio > max_mapq) holds for v up to a largest value, "Col valid" = !(ratio < min_mapq) and "End valid" = !(ratio <
// min_clip) from a smallest value on — found with window_record's own divisions and comparisons, once per set of thresholds; the first pass
// of hf_create then needs three byte-sized table entries of its window's coverage (768 bytes: L1) instead of two divisions.
struct ValidityLut {
    double max_mapq = -1.0, min_mapq = -1.0, min_clip = -1.0;
    uint16_t dup_le[256], col_ge[256], end_ge[256];     // Dup valid iff mq <= dup_le[cv] (0xffff: never... see below); Col iff mq >= col_ge[cv]; End iff cp >= end_ge[cv]
    bool monotone = true;                               // (false: a NaN threshold or the like broke the monotone pattern — hf_create then takes window_record everywhere)
};
std::shared_ptr<const ValidityLut> validity_lut(double max_mapq, double min_mapq, double min_clip) {
    static std::mutex mu;
    static std::shared_ptr<const ValidityLut> cached;
    std::lock_guard<std::mutex> g(mu);
    if (cached && cached->max_mapq == max_mapq && cached->min_mapq == min_mapq && cached->min_clip == min_clip) return cached;
    auto L = std::make_shared<ValidityLut>();
    L->max_mapq = max_mapq; L->min_mapq = min_mapq; L->min_clip = min_clip;
    for (unsigned cv = 0; cv < 256; cv++) {
        int n_dup = 0, n_col = 0, n_end = 0;            // values v for which the bit is set; then check that they form a prefix / suffixes
        bool dup[256], col[256], end[256];
        for (unsigned v = 0; v < 256; v++) {
            const double ratio = (double) v / (0.1 + cv);                 // window_record's own expressions
            dup[v] = !(ratio > max_mapq); col[v] = !(ratio < min_mapq); end[v] = !(ratio < min_clip);
            n_dup += dup[v]; n_col += col[v]; n_end += end[v];
        }
        for (unsigned v = 0; v < 256; v++)
            if (dup[v] != ((int) v < n_dup) || col[v] != ((int) v >= 256 - n_col) || end[v] != ((int) v >= 256 - n_end)) L->monotone = false;
        L->dup_le[cv] = (uint16_t) (n_dup - 1 < 0 ? 0xffff : n_dup - 1);   // 0xffff: no value is valid (compared as signed below)
        L->col_ge[cv] = (uint16_t) (256 - n_col);                          // 256: no value
        L->end_ge[cv] = (uint16_t) (256 - n_end);
    }
    cached = L;
    return cached;
}
}
// threads of hf_create's passes over the windows: 16 (what rounds 2-4 used), fewer on a smaller host; HF_HOST_THREADS=n (<= 64) overrides
static size_t host_threads() {
    static const size_t n = [] {
        const unsigned hw = std::thread::hardware_concurrency();
        size_t t = std::min<size_t>(16, hw ? hw : 1);
        if (const char* e = std::getenv("HF_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) t = std::min<size_t>((size_t) v, HF_PARTS); }
        return t;
    }();
    return n;
}
template <class Fn>
static void par_chunks(const int64_t* chunk_off, size_t C, Fn fn, size_t parts_per_thread = 4) {
    const unsigned hw = std::thread::hardware_concurrency();
    (void) hw;
    // Round 6: FOUR parts per thread, handed out dynamically (HostPool::run), instead of one static part each — on a box the driver leased
    // fresh, hf_create took 8.35 ms against 2.9-4.3 on others: a pass over the windows lasted as long as the thread that woke last.  Parts stay
    // consecutive chunk ranges in order (what the slow-window lists and the plan's per-part ranks rely on), and the cut depends on the chunk
    // list and the thread count only: the second and third pass of hf_create see the same parts.
    const size_t NT = C < 16 ? 1 : std::min<size_t>(host_threads(), C);
    if (NT <= 1) { fn((size_t) 0, C, (size_t) 0); return; }
    const size_t T = std::min<size_t>(std::min<size_t>(HF_PARTS, parts_per_thread * NT), C);
    std::vector<size_t> cut(T + 1, 0);
    const int64_t total = chunk_off[C] - chunk_off[0];
    size_t c = 0;
    for (size_t k = 1; k < T; k++) {
        const int64_t target = chunk_off[0] + total * (int64_t) k / (int64_t) T;
        while (c < C && chunk_off[c] < target) c++;
        cut[k] = c;
    }
    cut[T] = C;
    // HF_PARTS_TRACE=1 (profiles/tools): when every part started and ended, and on which thread — how late the pool's threads wake
    static const bool ptrace = std::getenv("HF_PARTS_TRACE") != nullptr;
    struct PartLog { double t0, t1; size_t tid; };
    std::vector<PartLog> plog(ptrace ? T : 0);
    const auto pstart = std::chrono::steady_clock::now();
    const std::function<void(size_t)> part = [&](size_t k) {
        if (!ptrace) { fn(cut[k], cut[k + 1], k); return; }
        const double a = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - pstart).count();
        fn(cut[k], cut[k + 1], k);
        plog[k] = {a, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - pstart).count(),
                   std::hash<std::thread::id>()(std::this_thread::get_id()) % 997};
    };
    struct PartDump {
        const std::vector<PartLog>& l; bool on;
        ~PartDump() {
            if (!on) return;
            std::fprintf(stderr, "[parts]");
            for (size_t k = 0; k < l.size(); k++) std::fprintf(stderr, " %zu:t%zu:%.0f-%.0f", k, l[k].tid, l[k].t0, l[k].t1);
            std::fprintf(stderr, " us\n");
        }
    } pdump{plog, ptrace};
    if (host_pool().run(T, part, NT - 1)) return;
    std::atomic<size_t> next{0};                       // the pool is busy (hf_multi's ranks create their contexts concurrently): threads of our own
    auto work = [&] { for (size_t k; (k = next.fetch_add(1, std::memory_order_relaxed)) < T;) part(k); };
    std::vector<std::thread> th;
    for (size_t k = 1; k < NT; k++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}
struct hf_ctx;
static int cseg0_of(const hf_ctx* ctx, int c);      // first segment of chunk c (c = C: the number of segments)
// slot of a segment's x-th window: lane x / L holds it as its x % L-th (hf_seg.h)
static inline int32_t seg_slot(const SegDesc& d, int64_t x) { return d.slot0 + (int32_t) ((x % d.L) * 64 + x / d.L); }
static int set_err(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return set_err(HF_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct hf_ctx {
    int device = 0, algo = HF_ALGO_SCAN;
    // device memory of hf_create comes out of a few SLABS (round 5: ~45 hipMalloc calls and as many hipFree were ~0.5 ms of a 4 ms hf_create):
    // bump allocation, 256-byte aligned; what hf_create frees again (temporaries) simply stays until hf_destroy.  ctx_free knows which
    // pointers are the context's own later hipMallocs.
    std::vector<std::pair<char*, size_t>> slabs; char* slab_cur = nullptr; size_t slab_left = 0, slab_first = 0;
    int64_t N = 0; int32_t C = 0; int32_t maxT = 0;
    int R = 1, K = 2;
    int64_t V = 0;                 // per-chunk stats vector length
    hf_windows meta{};             // scalar options only (pointers cleared)
    // device-resident window store
    int64_t* d_off = nullptr;      // [C+1]
    uint32_t* d_rec = nullptr;     // [N]
    double* d_beta = nullptr;      // [N]
    uint64_t* d_regmask = nullptr; // [C] bit r set if region r occurs in the chunk
    // per-pass work arrays
    double* d_E = nullptr;         // [N][16] emission rows, HF_ALGO_SEQ only
    double* d_f = nullptr;         // [ntiles][L][2][64] double2: tile-major, lane-minor (hf_scan.h fb_slot)
    double* d_b = nullptr;         // same layout
    std::vector<int64_t> h_off; std::vector<int32_t> h_tile0;   // host copies for hf_get_forward_backward
    double* d_scale = nullptr;     // [N]
    int8_t* d_label = nullptr;     // [N]
    double* d_chunk_stats = nullptr; // [C][V]; own_chunk_stats: allocated here (else bound to an exchange buffer, hf_bind_chunk_stats)
    bool own_chunk_stats = true;
    double* d_total = nullptr;     // [V]
    double* d_total_host = nullptr; // device address of the pinned h_total: k_reduce writes the result straight to the host
    // scan algorithm: tile tables and per-tile work arrays
    TileDesc* d_tile_desc = nullptr;
    int ntiles = 0; int32_t* d_chunk_tile0 = nullptr;
    bool host_trace = false; double ht[5] = {0, 0, 0, 0, 0}; long ht_n = 0;   // HF_HOST_TRACE=1: where an EM step's host time goes
    double* d_tile_ll = nullptr;
    double* d_tile_stats = nullptr; // [ntiles][R][NA(16)]
    unsigned* d_flags = nullptr;
    TableJob* d_jobs = nullptr; TabWork tabwork{};   // the job list of the per-pass tables (hf_scan.h k_build_jobs), built once
    DevParams* d_params = nullptr; DevParams* h_params = nullptr; size_t params_bytes = 0;
    KParams kparams{};             // the parameter block as kernel arguments of k_tables (one region: pack_kparams)
    bool kp_ok = true, kp_now = false;   // HF_PARAMS_COPY=1 switches the kernel-argument path off; this pass uses it
    unsigned* h_flags = nullptr;
    int8_t* h_label = nullptr;     // pinned [N]: staging of hf_get_labels, taken from the process's pinned cache on first use (1.5 MB of pinning = 0.15 ms of hf_create until round 5)
    int8_t* d_label_host = nullptr; // its device address (a kernel writes the labels there)
    double* h_total = nullptr;     // pinned [V+1]: reduced vector + flag word, one copy per pass
    hipEvent_t ev0 = nullptr, ev1 = nullptr; bool ev_valid = false;
    double ksum[HF_NKERNELS] = {}; int64_t kcount[HF_NKERNELS] = {};   // accumulated by hf_finish while profiling is on
    float klast[HF_NKERNELS] = {}; bool klast_ok[HF_NKERNELS] = {};      // the last pass's durations as hf_finish read them (hf_kernel_times returns these)
    unsigned prof_mask = 0;        // bit k: kernel k (HF_K_*) is bracketed by kev[2k], kev[2k+1]
    int prof_stride = 1; long prof_pass = 0; bool prof_now = true;   // events only in every prof_stride-th pass (hf_set_profiling_stride)
    hipEvent_t kev[2 * HF_NKERNELS] = {}; bool kran[HF_NKERNELS] = {};
    bool have_full = false;
    size_t lds_max = 64 * 1024;    // LDS one workgroup may use (hipDeviceAttributeMaxSharedMemoryPerBlock)
    bool launch_failed = false;
    double beta_star = 1.0;
    // per-iteration emission rows (k_tables): keys = occurring (region, x, x_prev) of interior windows,
    // slow = chunk-first and contig-end windows (beta != beta_star), ascending; slow_off[c] = chunk c's first entry
    int M = 1; double* d_lutE = nullptr; double* d_lutC = nullptr; int64_t n_lut = 0;
    int n_keys = 0; int32_t* d_keys = nullptr;
    // negative_binomial model: device copies of hf_params.nb_* and the per-tile count data (allocated on first use)
    double *d_nbE = nullptr, *d_nbP = nullptr, *d_nbDig = nullptr, *d_nbR = nullptr, *d_nbBeta = nullptr, *d_tile_hist = nullptr;   // d_nbE owns ONE buffer: E | P | dig | r | beta
    char* h_nb[2] = {nullptr, nullptr}; hipEvent_t nb_ev[2] = {nullptr, nullptr}; bool nb_ev_used[2] = {false, false}; unsigned nb_turn = 0;   // pinned staging of those tables
    // statistics by emission row (hf_rows.h): the static plan and its work arrays
    int stats_mode = HF_STATS_CHUNKS; bool rows_ready = false, pass_rows = false; int pass_kc = 0, pass_wpb = 4;
    double poll_seq = 0.0;         // completion stamp of the last polled pass (hf_finish)
    // the total of a one-GPU pass summed by the HOST (hf_rows.h part_host): pinned partials [n_rw_blocks][NA] | [C] | flag word, their device address
    double* h_part = nullptr; double* d_part_host = nullptr; size_t part_cap = 0; bool host_total_ok = true, pass_host_total = false; int pass_rw_blocks = 0;
    std::vector<int32_t> h_rw_off;
    double* d_rank_out = nullptr; double* d_rank_flag = nullptr;   // hf_bind_rank_total: where a rows-mode pass writes its total / flag word
    bool pass_bound = false;       // the last pass wrote them there
    bool stream_stamp_ok = true; uint32_t stream_stamp = 0;   // wait_total: completion through hipStreamWriteValue32 (HF_STREAM_STAMP=0: off)
    bool pass_polled = false;      // the last rows-mode pass carried a stamp (decided at launch: k_row_stats writes the total itself)
    unsigned* d_done = nullptr;    // k_reduce / k_row_stats: blocks finished (the last one sums the parts and stamps the host block)
    unsigned long long* d_cks = nullptr;   // k_reduce: XOR of the words written (checksum of a polled pass)
    int poll_kind = 0;             // what the last polled kernel was: 1 k_row_stats' total (a checksum per region), 2 k_reduce (one)
    int n_groups = 0, n_rowwaves = 0, n_parts = 1;   // n_parts: regions that have a row, + 1 (the log-likelihood blocks): hf_rows.h hand-offs
    bool pass_nb = false;          // the last rows-mode pass ran the negative_binomial kernels (hf_nb_rows.h)
    int32_t* d_bin_off = nullptr; int32_t* d_bin_list = nullptr; double* d_slot_h = nullptr; double* d_H = nullptr;   // count-data plan
    double* d_chunk_ll = nullptr;  // [C] log-likelihood per chunk (rows mode)
    double* d_recs = nullptr;      // pair records { f_{t-1}, b_t } of k_seg_fb, plan order; fb_recs: the last full pass wrote them
    bool fb_recs = false, pass_pairs_done = false;   // pass_pairs_done: enqueue_pass ran k_pair_sums itself (per sub-pass)
    // SUB-PASSES (round 5): a context whose records would not fit the Infinity Cache (256 MB; BASELINE configs[2] writes 130 MB) cuts its
    // chunk list into sub-passes of whole chunks; a pass runs k_seg_fb then k_pair_sums per sub-pass through ONE buffer that holds a
    // sub-pass's positions [p0, p1) at a time (d_recs; the plan's groups are numbered sub-pass by sub-pass, so a sub-pass's records are
    // contiguous) — the records never travel to HBM and back.  Everything that needs the records of ALL windows at once (the getters,
    // per-chunk statistics) runs the segment kernel into d_recs_all instead (allocated on first use; = d_recs with one sub-pass).
    struct SubPass { int c0, c1, seg0, seg1, g0, g1; int64_t p0, p1; int b0 = 0, b1 = 0; };   // b0, b1: its blocks in the XCD plan (d_seg_of_block)
    std::vector<SubPass> subs;
    double* d_recs_all = nullptr; bool recs_all = false;   // recs_all: d_recs_all holds the records of every window of the last full pass
    bool scales_all = false;       // ... and d_scale_s its scales (an EM pass writes none: 8 of its 72 bytes per window that only
                                   // hf_get_forward_backward reads — the getter runs the segment kernel again, with the array)
    int32_t* d_grp_off = nullptr;     // compact plan: first position of every group (+ the end)
    bool plan_compact = false;        // the groups' records back to back (sparse rows) instead of 64 positions per group
    int rs_bpw = 1;                   // batches of 16 row slots per wavefront of k_row_stats
    int32_t* d_grp_ar = nullptr; int32_t* d_grp_n = nullptr; double* d_grp_sums = nullptr;   // per group: its row of A, its pairs
    int32_t* d_pos_f = nullptr;    // [N] position of the record that holds every window's f (getters)
    int32_t* d_slot_of = nullptr;  // window -> slot, built from h_segs and uploaded by the first getter call
    int32_t* d_pos = nullptr; int64_t n_pos = 0;   // [N] position of every window's pair record in d_recs (plan order; slot order without a plan)
    RowSlot* d_rowslots = nullptr; int32_t* d_rw_region = nullptr; int32_t* d_rw_off = nullptr; double* d_rw_stats = nullptr;
    // segment kernels (hf_seg.h): the forward-backward of the statistics-by-row path
    SegDesc* d_seg = nullptr; int nseg = 0; int32_t* d_chunk_seg0 = nullptr; double* d_seg_ll = nullptr; double* d_Pseg = nullptr;
    double* d_segQ = nullptr;          // [nseg][8][64] double2: lane products, lane-minor (two-launch mode only: allocated on first use)
    // one-launch mode (hf_seg.h): k_seg_fb computes the lane products itself and the segments of a chunk hand their products to
    // each other inside the launch (flags stamped with the launch's epoch); a timed-out wait switches the context to two launches
    unsigned* d_seg_ready = nullptr; unsigned seg_epoch = 0; bool seg_fused = true, seg_test_timeout = false;
    int seg_nc = 0;                    // row blocks a segment workgroup keeps in LDS across its three walks (hf_seg.h: chosen so that all segments stay resident)
    // XCD plan (round 6): block b of k_seg_fb runs segment seg_of_block[b]; the segments of a chunk sit on block indices congruent mod 8
    // (observed: block b runs on XCD b % 8 — for speed only, the hand-off protocol is system-scope whatever the placement).  Null: identity.
    int32_t* d_seg_of_block = nullptr; std::vector<int32_t> h_seg_of_block;
    hf_params last_p{}; int last_mode = HF_MODE_FULL;   // what the last hf_estep was given (the fallback re-runs the pass)
    hipStream_t last_stream = nullptr;                  // ... and the stream it ran on (the getters' re-run of the segment kernel goes there)
    // rows of A_t = T_t∘e_t (hf_seg.h): one per (emission key, transition class) that occurs at an interior window, then one
    // per slow window; d_arow[t] = the row of window t (bit 31: chunk-first), d_arow_src / d_arow_cls = where a row comes from
    int32_t* d_arow = nullptr; int32_t* d_arow_src = nullptr; int32_t* d_arow_cls = nullptr; double* d_lutA = nullptr;
    int n_arows = 0, n_combo = 0;
    double* d_scale_s = nullptr; int64_t n_slots = 0;
    std::vector<SegDesc> h_segs;                // the segment descriptors (the getters derive a window's slot from them)
    std::vector<int32_t> h_cseg0;               // first segment of every chunk (+ the end)
    std::vector<std::pair<const char*, double>> create_phases;   // hf_create_phases
    bool pass_seg = false;             // the last pass ran the segment kernels (log-likelihood partials per segment)
    unsigned long long* d_seg_trace = nullptr;   // -DHF_SEG_TRACE builds only
    int n_slow = 0; int64_t* d_slow_w = nullptr; int32_t* d_slow_off = nullptr; double* d_Es = nullptr; double* d_Cs = nullptr;
};

static int cseg0_of(const hf_ctx* ctx, int c) { return ctx->h_cseg0[(size_t) c]; }
static void* ctx_alloc(hf_ctx* ctx, size_t bytes) {
    bytes = (bytes ? bytes : 8);
    bytes = (bytes + 255) & ~(size_t) 255;
    if (bytes > ctx->slab_left) {
        const size_t want = std::max(bytes, ctx->slabs.empty() ? ctx->slab_first : (size_t) 64 << 20);
        char* p = nullptr;
        if (hipMalloc((void**) &p, want) != hipSuccess) {
            (void) hipGetLastError();
            if (want == bytes || hipMalloc((void**) &p, bytes) != hipSuccess) return nullptr;   // (a smaller slab: just this request)
            ctx->slabs.emplace_back(p, bytes);
            return p;                                                                         // (the current slab keeps its remainder)
        }
        ctx->slabs.emplace_back(p, want);
        ctx->slab_cur = p; ctx->slab_left = want;
    }
    char* r = ctx->slab_cur;
    ctx->slab_cur += bytes; ctx->slab_left -= bytes;
    return r;
}
static void ctx_free(hf_ctx* ctx, void* p) {          // hipFree unless the pointer lives in a slab
    if (!p) return;
    for (const auto& sl : ctx->slabs)
        if ((char*) p >= sl.first && (char*) p < sl.first + sl.second) return;
    hipFree(p);
}

// ------------------------------------------------------------------------------------------
// setup: packed records + contig-end factor beta (hmm.c:301-316), once per run
// ------------------------------------------------------------------------------------------
// packed != nullptr: the four per-window inputs as one word cov | mapq << 8 | clip << 16 | region << 24 (hf_create packs them on the
// host when every value fits a byte — window values are at most 250, chunk.c:393-441: a quarter of the bytes to upload)
// The packed record of one window (hf_device.h REC_*) and its beta: ONE function for the device (k_setup) and for the host
// (hf_create's plan reads the same bits without fetching them back; -ffp-contract=off on both sides, plain IEEE divisions and
// comparisons, so the two agree bit for bit -- HF_CREATE_VERIFY=1, set by the test suite, downloads the device's and compares).
__host__ __device__ static inline uint32_t window_record(unsigned cv, unsigned mq, unsigned cp, unsigned region, unsigned pre_region,
                                                         int64_t col, int s, int e, int ctg_len, int window_len, int mean_read_len,
                                                         int adjust, double min_frac, double max_mapq, double min_mapq,
                                                         double min_clip, double beta_star, double* beta_out) {
    // validity mask, hmm_utils.c:2229-2254
    const double ratio_m = (double) mq / (0.1 + cv);
    const double ratio_c = (double) cp / (0.1 + cv);
    unsigned vm = 0;
    if (!(ratio_m > max_mapq)) vm |= 1u;  // Dup valid
    if (!(ratio_m < min_mapq)) vm |= 2u;  // Col valid
    if (!(ratio_c < min_clip)) vm |= 4u;  // End/Msj column valid
    uint32_t r = (cv & 0xffu) | (region << 8) | (vm << 16);
    if (col == 0) r |= 1u << 19;
    else if (pre_region != region) r |= 1u << 20;
    // beta, hmm.c:301-316; min/max are the int functions of common.c:142-148
    double bt = 1.0;
    if (adjust) {
        const int icol = (int) col;
        const int a1 = (int) (s + (double) window_len * (icol + 0.5));
        const int a2 = (int) ((s + (double) window_len * icol + e) / 2);
        const int mid = a1 < a2 ? a1 : a2;
        const int L = mean_read_len;
        const int l1 = mid - L + 1, l2 = (int) (-(1 - min_frac) * L);
        const int l = l2 < l1 ? l1 : l2;
        const int u2 = (int) (ctg_len - min_frac * L);
        const int u = mid < u2 ? mid : u2;
        bt = (double) (u - l) / L;
        if (bt <= 0.25) bt = 0.25;
    }
    *beta_out = bt;
    if (col == 0 || bt != beta_star) r |= 1u << 21;   // private emission row (REC_SLOW)
    return r;
}

__global__ void k_setup(const int64_t* __restrict__ off, const uint32_t* __restrict__ packed, const uint16_t* __restrict__ cov,
                        const uint16_t* __restrict__ mapq, const uint16_t* __restrict__ clip,
                        const uint64_t* __restrict__ annot, const int32_t* __restrict__ cs,
                        const int32_t* __restrict__ ce, const int32_t* __restrict__ cl, int window_len,
                        int mean_read_len, int adjust, double min_frac, double max_mapq, double min_mapq,
                        double min_clip, int n_regions, double beta_star, uint32_t* __restrict__ rec,
                        double* __restrict__ beta, unsigned* __restrict__ flags) {
    const int c = blockIdx.y;
    const int64_t t0 = off[c], T = off[c + 1] - t0;
    const int64_t col = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= T) return;
    const int64_t t = t0 + col;
    const uint32_t pw = packed ? packed[t] : 0u;
    const unsigned cv = packed ? (pw & 0xffu) : cov[t];
    const unsigned mq = packed ? ((pw >> 8) & 0xffu) : mapq[t], cp = packed ? ((pw >> 16) & 0xffu) : clip[t];
    const unsigned region = packed ? (pw >> 24) : (unsigned) ((annot[t] & 0xFC00000000000000ULL) >> 58);
    if ((int) region >= n_regions) atomicOr(flags, HF_FLAG_REGION);
    unsigned pre_region = region;
    if (col > 0) pre_region = packed ? (packed[t - 1] >> 24) : (unsigned) ((annot[t - 1] & 0xFC00000000000000ULL) >> 58);
    double bt;
    rec[t] = window_record(cv, mq, cp, region, pre_region, col, cs[c], ce[c], cl[c], window_len, mean_read_len, adjust, min_frac,
                           max_mapq, min_mapq, min_clip, beta_star, &bt);
    beta[t] = bt;
}

__global__ void k_regmask(const int64_t* __restrict__ off, const uint32_t* __restrict__ rec,
                          uint64_t* __restrict__ regmask) {
    const int c = blockIdx.x;
    const int64_t t0 = off[c], T = off[c + 1] - t0;
    unsigned long long m = 0;
    for (int64_t i = threadIdx.x; i < T; i += blockDim.x) m |= 1ull << (REC_REGION(rec[t0 + i]) & 63u);
    for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o);
    __shared__ unsigned long long sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = 0;
        for (int w = 0; w < (int) (blockDim.x >> 6); w++) a |= sm[w];
        regmask[c] = a;
    }
}

// position of the record that holds every window's forward vector (the getters): the record of the NEXT window of its chunk, the chunk's
// spare record for the last one
__global__ void k_pos_f(const int64_t* __restrict__ off, const int32_t* __restrict__ pos, const int32_t* __restrict__ spare, int32_t* __restrict__ pos_f) {
    const int c = blockIdx.y;
    const int64_t t0 = off[c], T = off[c + 1] - t0;
    const int64_t col = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= T) return;
    pos_f[t0 + col] = col + 1 < T ? pos[t0 + col + 1] : spare[c];
}

#include "hf_scan.h"
#include "hf_seq.h"
#include "hf_nb.h"
#include "hf_chunks.h"
#include "hf_rows.h"
#include "hf_nb_rows.h"
#include "hf_seg.h"


// ------------------------------------------------------------------------------------------
// host side of the C ABI
// ------------------------------------------------------------------------------------------
// the event pair around a whole pass (hf_last_kernel_ms) costs two extra packets per step: only while profiling is on
static bool pass_events(const hf_ctx* ctx) {
    return (ctx->prof_mask & HF_PROF_PASS) != 0 || ctx->host_trace;
}

// small uploads of hf_create: out of the context's slab, through a pinned staging area when there is room (memcpy + ASYNCHRONOUS copy on the
// null stream: hf_create synchronises once, at its end) — a hipMalloc and a synchronous copy out of pageable memory per array were ~30 us each
struct UploadStage { char* p = nullptr; size_t cap = 0, used = 0; };
template <typename T>
static int dev_upload(hf_ctx* ctx, UploadStage& stg, T** dst, const T* src, size_t n) {
    *dst = static_cast<T*>(ctx_alloc(ctx, (n ? n : 1) * sizeof(T)));
    if (!*dst) return set_err(HF_E_HIP, "hf_create: out of device memory");
    if (!n) return 0;
    const size_t bytes = n * sizeof(T), need = (bytes + 63) & ~(size_t) 63;
    if (stg.p && stg.used + need <= stg.cap) {
        std::memcpy(stg.p + stg.used, src, bytes);
        HIPCHK(hipMemcpyAsync(*dst, stg.p + stg.used, bytes, hipMemcpyHostToDevice, nullptr));
        stg.used += need;
    } else HIPCHK(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

// event pair around one kernel launch, only for the kernels selected by hf_set_profiling
struct KTimer {
    hf_ctx* c; hipStream_t st; int k; bool on;
    KTimer(hf_ctx* c_, hipStream_t s_, int k_) : c(c_), st(s_), k(k_), on(((c_->prof_mask >> k_) & 1u) && c_->prof_now) {
        if (on) hipEventRecord(c->kev[2 * k], st);
    }
    ~KTimer() { if (on) { hipEventRecord(c->kev[2 * k + 1], st); c->kran[k] = true; } }
};

// Launch geometry of a one-wavefront-per-tile kernel: 4 wavefronts per block, fewer when the per-region transition
// tables (1088 B per region) plus the per-wavefront LDS do not fit (many regions / components); dynamic LDS above the
// default 64 KiB is requested explicitly.
struct TileGeom { unsigned blocks = 0, threads = 256; size_t lds = 0; bool ok = false; };
template <class Kern>
static TileGeom tile_geom(const hf_ctx* ctx, Kern kernel, size_t per_wave_bytes, bool with_tab = true, int wmax = 4) {
    TileGeom g;
    const size_t tab = with_tab ? (size_t) ctx->R * HF_TAB_STRIDE * 8 : 0;
    int w = wmax;
    while (w >= 1 && tab + (size_t) w * per_wave_bytes > ctx->lds_max) w--;
    if (w < 1) return g;
    g.threads = 64u * (unsigned) w;
    g.blocks = (unsigned) ((ctx->ntiles + w - 1) / w);
    g.lds = tab + (size_t) w * per_wave_bytes;
    g.ok = true;
    if (g.lds > 64 * 1024)
        g.ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) g.lds) == hipSuccess;
    return g;
}
#define TILE_GEOM_OR_FAIL(g) do { if (!(g).ok) { set_err(HF_E_ARG, "the per-region tables do not fit the LDS of one workgroup"); ctx->launch_failed = true; return; } } while (0)

static RowSrc row_src(const hf_ctx* ctx) {
    RowSrc S;
    S.lutE = ctx->d_lutE; S.lutC = ctx->d_lutC; S.Es = ctx->d_Es; S.Cs = ctx->d_Cs; S.M = ctx->M; S.K = ctx->K; S.n_lut = (int) ctx->n_lut;
    return S;
}

// where the log-likelihood partials of the last forward pass are: per segment (hf_seg.h) or per tile (hf_scan.h)
static const int32_t* ll_off(const hf_ctx* ctx) { return ctx->pass_seg ? ctx->d_chunk_seg0 : ctx->d_chunk_tile0; }
static const double* ll_part(const hf_ctx* ctx) { return ctx->pass_seg ? ctx->d_seg_ll : ctx->d_tile_ll; }

// HF_ALGO_SCAN runs the segment kernels (hf_seg.h) whatever the statistics path
static bool seg_pass(const hf_ctx* ctx) { return ctx->algo == HF_ALGO_SCAN && ctx->nseg > 0; }

// does a full pass of the Gaussian models take the statistics-by-row path?
static bool rows_pass(const hf_ctx* ctx) { return ctx->stats_mode == HF_STATS_ROWS && ctx->rows_ready && ctx->algo == HF_ALGO_SCAN; }

// polled completion of a pass: see wait_total
static int poll_mode() {
    static const int mode = [] {
        const char* e = std::getenv("HF_POLL");
        if (!e) return 0;
        if (!std::strcmp(e, "debug")) return 2;
        return e[0] == '1' ? 1 : 0;
    }();
    return mode;
}
static bool poll_ok(const hf_ctx* ctx, int last_kernel) {
    return poll_mode() != 0 && ctx->d_total_host && !((ctx->prof_mask >> last_kernel) & 1u) && !(ctx->prof_mask & HF_PROF_PASS) &&
           !ctx->host_trace;
}
static double next_stamp(hf_ctx* ctx) { ctx->poll_seq += 1.0; ctx->h_total[ctx->V + 1] = 0.0; return ctx->poll_seq; }

// the groups' sums of f (x) b, times their row of A: padded plan (whole groups streamed) or compact plan (hf_create)
// the groups of ONE sub-pass (hf_ctx::SubPass): padded plan — whole groups streamed from the sub-pass's first position on; compact plan — the
// groups' own (global) positions.  recs_eff: the address position 0 would have (a pass buffer that holds one sub-pass starts p0 records later)
static void launch_pair_sums(hf_ctx* ctx, hipStream_t st, const hf_ctx::SubPass& sb, const double* recs_eff) {
    const int n = sb.g1 - sb.g0;
    if (n <= 0) return;
    if (ctx->plan_compact)
        hipLaunchKernelGGL(k_pair_sums_compact, dim3((unsigned) ((n + 63) / 64)), dim3(256), 0, st, n, ctx->d_grp_ar + sb.g0, ctx->d_grp_off + sb.g0,
                           ctx->d_grp_n + sb.g0, ctx->d_lutA, recs_eff, ctx->d_grp_sums + (size_t) sb.g0 * 16);
    else
        hipLaunchKernelGGL(k_pair_sums, dim3((unsigned) ((n + 15) / 16)), dim3(256), 0, st, n, ctx->d_grp_ar + sb.g0, ctx->d_grp_n + sb.g0,
                           ctx->d_lutA, recs_eff + sb.p0 * 8, ctx->d_grp_sums + (size_t) sb.g0 * 16);
}

// k_seg_fb over the segments [g0, g0 + n) (hf_seg.h); timed: by the dispatch's own start / stop timestamps (hipExtLaunchKernelGGL hands the two
// events to the launch: what rocprofv3 reports for the kernel, without the two marker packets of an event pair around it — those measured 3 us
// more than the kernel and cost the step ~15 us)
// sub >= 0: the segments of that sub-pass; < 0: all segments
static void launch_seg_fb(hf_ctx* ctx, hipStream_t st, bool full, double* recs_eff, int sub, unsigned epoch, unsigned wait_epoch, bool timed, bool with_scales = true) {
    int g0 = sub >= 0 ? ctx->subs[(size_t) sub].seg0 : 0, n = sub >= 0 ? ctx->subs[(size_t) sub].seg1 - g0 : ctx->nseg;
    const int32_t* const sob = ctx->seg_fused ? ctx->d_seg_of_block : nullptr;   // (the XCD plan: one-launch mode only — two launches have no hand-off inside the launch)
    if (sob) {       // blocks of the plan instead of segments (a sub-pass's blocks are contiguous, and so are all of them)
        g0 = sub >= 0 ? ctx->subs[(size_t) sub].b0 : 0;
        n = (sub >= 0 ? ctx->subs[(size_t) sub].b1 : ctx->subs.back().b1) - g0;
    }
    if (n <= 0) return;
    const int nc = ctx->seg_fused ? ctx->seg_nc : 0;     // cached row blocks: one-launch mode only (the lane products are computed in the same kernel)
    const size_t lds = seg_lds_bytes(nc);
#define HF_SEG_FB_ARGS ctx->d_seg, ctx->d_arow, ctx->d_lutA, ctx->d_params, ctx->d_segQ, ctx->d_Pseg, ctx->d_seg_ready, epoch, wait_epoch, ctx->d_pos, recs_eff, \
                        with_scales ? ctx->d_scale_s : (double*) nullptr, ctx->d_label, ctx->d_seg_ll, ctx->d_flags, (int32_t) g0, nc, sob
#define HF_SEG_FB_LAUNCH(B, F, CA) do { \
        if (timed) hipExtLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_fb<B, F, CA>), dim3((unsigned) n), dim3(64), (uint32_t) lds, st, \
                                         ctx->kev[2 * HF_K_SEG_FB], ctx->kev[2 * HF_K_SEG_FB + 1], 0, HF_SEG_FB_ARGS); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seg_fb<B, F, CA>), dim3((unsigned) n), dim3(64), lds, st, HF_SEG_FB_ARGS); } while (0)
    if (nc > 0) { if (full) HF_SEG_FB_LAUNCH(true, true, true); else HF_SEG_FB_LAUNCH(false, true, true); }
    else if (full) { if (ctx->seg_fused) HF_SEG_FB_LAUNCH(true, true, false); else HF_SEG_FB_LAUNCH(true, false, false); }
    else { if (ctx->seg_fused) HF_SEG_FB_LAUNCH(false, true, false); else HF_SEG_FB_LAUNCH(false, false, false); }
#undef HF_SEG_FB_LAUNCH
#undef HF_SEG_FB_ARGS
}

// the buffer that holds the records of ALL windows (one sub-pass: the pass buffer itself)
static int all_records_buffer(hf_ctx* ctx) {
    if (ctx->d_recs_all) return HF_OK;
    int64_t n_pos = 0;
    for (const auto& sb : ctx->subs) if (sb.p1 > n_pos) n_pos = sb.p1;
    HIPCHK(hipMalloc((void**) &ctx->d_recs_all, (size_t) (n_pos + 1) * 64));
    return HF_OK;
}

template <int KT>
static void launch_stats(hf_ctx* ctx, hipStream_t st, int full, int ncol) {
    ctx->pass_rows = false; ctx->pass_host_total = false;
    if (full && rows_pass(ctx)) {
        // statistics by emission row (hf_rows.h); the total of the pass comes out of k_row_stats' last blocks
        // (k_pair_sums ran behind every sub-pass's k_seg_fb: enqueue_pass)
        KTimer t(ctx, st, HF_K_ROW_STATS);
        constexpr size_t NA = 16 + 9 + 2 + 3 * KT + 1;
        // the wavefronts' sums: HF_RS_WPB wavefronts per block (a block stays inside one region: hf_create pads the regions to that multiple) — four
        // for a sparse plan, whose wavefronts take several batches of slots each: eight of those per block ran config 5 in 83 us instead of 51
        const int want_wpb = ctx->rs_bpw > 1 ? 4 : HF_RS_WPB;
        TileGeom g = tile_geom(ctx, k_row_stats<KT>, NA * 8, false, want_wpb);
        TILE_GEOM_OR_FAIL(g);
        if ((int) g.threads != 64 * want_wpb) { set_err(HF_E_ARG, "k_row_stats: the block's partial sums do not fit the LDS"); ctx->launch_failed = true; return; }
        const int wpb = (int) g.threads / 64;
        const int n_rw_blocks = (ctx->n_rowwaves + wpb - 1) / wpb, n_ll_blocks = (ctx->C + wpb - 1) / wpb;
        // the launch's last block also sums the partials (rows_total): into d_total and straight into the pinned host block
        const bool bound = ctx->d_rank_out != nullptr;    // multi-GPU `ranks` exchange: the total goes into the exchange buffer, not to the host
        const bool polled = !bound && poll_ok(ctx, HF_K_ROW_STATS);
        const double seq = polled ? next_stamp(ctx) : 0.0;
        ctx->pass_polled = polled; ctx->pass_bound = bound;
        // one GPU, nobody polling, nothing bound: the blocks' partial vectors go to pinned host memory and the HOST sums them (host_rows_total)
        bool host_total = !bound && !polled && ctx->host_total_ok && ctx->d_total_host != nullptr;
        if (host_total) {
            const size_t need = (size_t) n_rw_blocks * NA + (size_t) ctx->C + 1;
            if (need > ctx->part_cap) {
                if (ctx->h_part) hipHostFree(ctx->h_part);
                ctx->h_part = nullptr; ctx->d_part_host = nullptr; ctx->part_cap = 0;
                void* dp = nullptr;
                if (hipHostMalloc((void**) &ctx->h_part, (need + 64) * 8) == hipSuccess && hipHostGetDevicePointer(&dp, ctx->h_part, 0) == hipSuccess) {
                    ctx->d_part_host = (double*) dp; ctx->part_cap = need + 64;
                } else { (void) hipGetLastError(); if (ctx->h_part) hipHostFree(ctx->h_part); ctx->h_part = nullptr; ctx->host_total_ok = false; host_total = false; }
            }
        }
        ctx->pass_host_total = host_total; ctx->pass_rw_blocks = n_rw_blocks;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_row_stats<KT>), dim3((unsigned) (n_rw_blocks + n_ll_blocks)), dim3(g.threads), g.lds, st,
                           ctx->n_rowwaves, n_rw_blocks, ctx->d_rw_region, ctx->d_rowslots, ctx->d_grp_sums, row_src(ctx), ctx->d_params,
                           ctx->d_rw_stats, ctx->C, ll_off(ctx), ll_part(ctx), ctx->d_chunk_stats, ctx->V, ctx->d_chunk_ll,
                           ctx->d_rw_off, ctx->K, bound ? ctx->d_rank_out : ctx->d_total, bound ? (double*) nullptr : ctx->d_total_host,
                           bound ? ctx->d_rank_flag : (double*) nullptr, ctx->d_flags, seq, ctx->d_done, ctx->n_parts, ctx->rs_bpw,
                           host_total ? ctx->d_part_host : (double*) nullptr);
        ctx->pass_wpb = wpb;
        ctx->pass_rows = true;
        ctx->pass_kc = ncol;
        return;
    }
    if (full) {
        KTimer t(ctx, st, HF_K_STATS_TILE);
        const TileGeom g = tile_geom(ctx, k_stats_tile<KT>, (size_t) (3 * ncol > 28 ? 3 * ncol : 28) * 65 * 8);
        TILE_GEOM_OR_FAIL(g);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stats_tile<KT>), dim3(g.blocks), dim3(g.threads), g.lds, st, ctx->ntiles,
                           ctx->d_tile_desc, ctx->d_rec, row_src(ctx), ctx->d_params, ctx->pass_seg ? ctx->d_recs_all : ctx->d_f, ctx->d_b, ctx->d_regmask,
                           ctx->d_tile_stats, ctx->pass_seg ? ctx->d_pos : (const int32_t*) nullptr);
    }
    KTimer t(ctx, st, HF_K_CHUNK_STATS);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chunk_stats<KT>), dim3((unsigned) ctx->C), dim3(128), 0, st, ctx->d_chunk_tile0, ll_off(ctx),
                       ctx->d_regmask, ctx->d_tile_stats, ll_part(ctx), ctx->d_params, ctx->d_chunk_stats, ctx->V, ctx->K,
                       full);
}

// the last kernel of a negative_binomial pass in HF_STATS_ROWS mode: total vector (+ flag word) into `out`
// (the Gaussian models' total is written by the last block of k_row_stats itself: hf_rows.h rows_total)
static int launch_nb_total(hf_ctx* ctx, hipStream_t st, double* out, bool with_flags = true, double seq = 0.0) {
    KTimer t(ctx, st, HF_K_ROWS_TOTAL);
    NbTables nt;
    nt.E = ctx->d_nbE; nt.P = ctx->d_nbP; nt.dig = ctx->d_nbDig; nt.r = ctx->d_nbR; nt.beta = ctx->d_nbBeta;
    hipLaunchKernelGGL(k_nb_total, dim3((unsigned) ctx->R), dim3(1024), 0, st, ctx->d_rw_off, ctx->pass_wpb, ctx->d_rw_stats, ctx->d_H,
                       ctx->d_params, nt, ctx->d_chunk_ll, (int64_t) ctx->C, ctx->V, ctx->K, out,
                       with_flags ? ctx->d_flags : (const unsigned*) nullptr, seq, ctx->d_done);
    HIPCHK(hipGetLastError());
    return 0;
}

__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// host getters (hf_get_forward_backward): f, b and the scale of windows first .. first + n, gathered from the pair records
__global__ void k_gather_fb(int64_t first, int64_t n, const int32_t* __restrict__ pos, const int32_t* __restrict__ pos_f,
                            const int32_t* __restrict__ slot_of, const double* __restrict__ recs, const double* __restrict__ scale_s,
                            double* __restrict__ f, double* __restrict__ b, double* __restrict__ sc) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t t = first + i;
    const double* __restrict__ rf = recs + (int64_t) pos_f[t] * 8;
    const double* __restrict__ rb = recs + (int64_t) pos[t] * 8 + 4;
#pragma unroll
    for (int s = 0; s < 4; s++) { f[i * 4 + s] = rf[s]; b[i * 4 + s] = rb[s]; }
    sc[i] = scale_s[slot_of[t]];
}

#include "hf_create.h"

extern "C" {

const char* hf_version(void) { return "flagger_amd 0.1 (gfx950)"; }
const char* hf_last_error(void) { return g_err.c_str(); }

int hf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


__global__ void k_selftest_div(int64_t n, const double* __restrict__ a, const double* __restrict__ d, double* __restrict__ fast,
                               double* __restrict__ exact, int32_t* __restrict__ safe);

int hf_warmup(int device) {
    if (hf_device_count() <= 0) return set_err(HF_E_NOGPU, "hf_warmup: no HIP device");
    {   // the host threads of hf_create's passes over the windows
        const unsigned hw = std::thread::hardware_concurrency();
        (void) hw;
        host_pool().start(host_threads() - 1);
    }
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(nullptr));      // forces the context
    hipLaunchKernelGGL(k_selftest_div, dim3(1), dim3(64), 0, 0, (int64_t) 0, (const double*) nullptr, (const double*) nullptr,
                       (double*) nullptr, (double*) nullptr, (int32_t*) nullptr);   // loads this library's code object
    HIPCHK(hipDeviceSynchronize());
    {   // the first pinned allocation and the first copy in each direction cost several milliseconds each (queues, staging): pay them here
        void *d = nullptr, *h = nullptr;
        if (hipMalloc(&d, 8 << 20) == hipSuccess && hipHostMalloc(&h, 8 << 20) == hipSuccess) {   // (large: a 1 MiB copy takes another path)
            (void) hipMemcpy(d, h, 8 << 20, hipMemcpyHostToDevice);
            (void) hipMemcpy(h, d, 8 << 20, hipMemcpyDeviceToHost);
            (void) hipMemset(d, 0, 64);
        }
        if (h) hipHostFree(h);
        if (d) hipFree(d);
        (void) hipGetLastError();
    }
    {   // hf_create's pinned staging buffers (PinCache), for up to 2 M windows: pinning 32 MB takes ~6 ms — here, while the caller reads its input
        PinCache& pc = pin_cache();
        char* a = pc.acquire((size_t) 4 * (2u << 20));
        char* b = pc.acquire((size_t) 12 * (2u << 20));
        char* c = pc.acquire((size_t) 8 << 20);
        char* d = pc.acquire((size_t) 64 << 10);     // a context's result / parameter block
        pc.release(a); pc.release(b); pc.release(c); pc.release(d);
    }
    return HF_OK;
}

void hf_destroy(hf_ctx* ctx) {
#ifdef HF_KSTAMP
    if (ctx && std::getenv("HF_KSTAMP_FILE")) {
        unsigned long long h[64 * 4 + 1];
        hipSetDevice(ctx->device); hipDeviceSynchronize();
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_kstamp), sizeof(h)) == hipSuccess)
            if (FILE* fp = std::fopen(std::getenv("HF_KSTAMP_FILE"), "ab")) { std::fwrite(h, 8, 64 * 4 + 1, fp); std::fclose(fp); }
    }
#endif
    if (!ctx) return;
    hipSetDevice(ctx->device);
#ifdef HF_SEG_TRACE
    if (ctx->d_seg_trace) {
        hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t) ctx->nseg * HF_SEG_TRACE_N);
        hipMemcpy(h.data(), ctx->d_seg_trace, h.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* fp = std::fopen(std::getenv("HF_SEG_TRACE_FILE"), "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
        unsigned long long* null_p = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_seg_trace), &null_p, sizeof(void*));
        ctx_free(ctx, ctx->d_seg_trace);
    }
#endif
    if (ctx->host_trace && ctx->ht_n)
        std::fprintf(stderr, "[hf host trace] %ld EM steps: parameter view (hfm_params: the negative-binomial tables) %.1f us, enqueue %.1f us, wait %.1f us, m-step %.1f us, gpu span (first launch .. reduction) %.1f us\n",
                     ctx->ht_n, ctx->ht[4] / ctx->ht_n, ctx->ht[0] / ctx->ht_n, ctx->ht[1] / ctx->ht_n, ctx->ht[2] / ctx->ht_n, ctx->ht[3] / ctx->ht_n);
    ctx_free(ctx, ctx->d_off); ctx_free(ctx, ctx->d_rec); ctx_free(ctx, ctx->d_beta); ctx_free(ctx, ctx->d_regmask); ctx_free(ctx, ctx->d_E);
    ctx_free(ctx, ctx->d_f); ctx_free(ctx, ctx->d_b); ctx_free(ctx, ctx->d_scale); ctx_free(ctx, ctx->d_label); if (ctx->own_chunk_stats) ctx_free(ctx, ctx->d_chunk_stats);
    ctx_free(ctx, ctx->d_total); ctx_free(ctx, ctx->d_flags); ctx_free(ctx, ctx->d_params);
    ctx_free(ctx, ctx->d_lutE); ctx_free(ctx, ctx->d_lutC); ctx_free(ctx, ctx->d_slow_w); ctx_free(ctx, ctx->d_slow_off); ctx_free(ctx, ctx->d_keys);
    ctx_free(ctx, ctx->d_nbE); ctx_free(ctx, ctx->d_tile_hist);   // (d_nbP .. d_nbBeta point into d_nbE's buffer)
    for (int b = 0; b < 2; b++) { if (ctx->h_nb[b]) hipHostFree(ctx->h_nb[b]); if (ctx->nb_ev[b]) hipEventDestroy(ctx->nb_ev[b]); }
    ctx_free(ctx, ctx->d_tile_desc); ctx_free(ctx, ctx->d_chunk_tile0);
    ctx_free(ctx, ctx->d_done); ctx_free(ctx, ctx->d_cks); ctx_free(ctx, ctx->d_bin_off); ctx_free(ctx, ctx->d_bin_list); ctx_free(ctx, ctx->d_slot_h); ctx_free(ctx, ctx->d_H); if (ctx->d_recs_all != ctx->d_recs) ctx_free(ctx, ctx->d_recs_all); ctx_free(ctx, ctx->d_recs); ctx_free(ctx, ctx->d_chunk_ll); ctx_free(ctx, ctx->d_grp_ar); ctx_free(ctx, ctx->d_grp_n); ctx_free(ctx, ctx->d_grp_off); ctx_free(ctx, ctx->d_pos); ctx_free(ctx, ctx->d_pos_f); ctx_free(ctx, ctx->d_slot_of); ctx_free(ctx, ctx->d_grp_sums); ctx_free(ctx, ctx->d_rowslots); ctx_free(ctx, ctx->d_rw_region);
    ctx_free(ctx, ctx->d_seg_of_block); ctx_free(ctx, ctx->d_seg); ctx_free(ctx, ctx->d_chunk_seg0); ctx_free(ctx, ctx->d_seg_ll); ctx_free(ctx, ctx->d_Pseg); ctx_free(ctx, ctx->d_segQ); ctx_free(ctx, ctx->d_seg_ready); ctx_free(ctx, ctx->d_scale_s);
    ctx_free(ctx, ctx->d_jobs); ctx_free(ctx, ctx->d_arow); ctx_free(ctx, ctx->d_arow_src); ctx_free(ctx, ctx->d_arow_cls); ctx_free(ctx, ctx->d_lutA);
    ctx_free(ctx, ctx->d_rw_off); ctx_free(ctx, ctx->d_rw_stats);
    ctx_free(ctx, ctx->d_tile_ll); ctx_free(ctx, ctx->d_tile_stats);
    if (ctx->h_part) hipHostFree(ctx->h_part);
    if (ctx->h_label) pin_cache().release(reinterpret_cast<char*>(ctx->h_label));
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    for (int i = 0; i < 2 * HF_NKERNELS; i++) if (ctx->kev[i]) hipEventDestroy(ctx->kev[i]);
    for (auto& sl : ctx->slabs) hipFree(sl.first);
    if (ctx->h_total) { hipDeviceSynchronize(); pin_cache().release(reinterpret_cast<char*>(ctx->h_total)); }   // one pinned block (h_flags and h_params live in it), back to the cache once nothing can write it any more
    delete ctx;
}

// pack one iteration's model: conditional transition tables (hmm_utils.c:2278-2292) and the
// distinct alpha values of each column
static int pack_params(hf_ctx* ctx, const hf_params* p) {
    if (p->n_regions != ctx->R) return set_err(HF_E_ARG, "hf_estep: n_regions differs from hf_create");
    for (int s = 0; s < 4; s++)
        if (p->ncomp[s] < 1 || p->ncomp[s] > ctx->K) return set_err(HF_E_ARG, "hf_estep: ncomp out of range");
    if (p->ncomp[1] != 1 || p->ncomp[2] != 1 || p->ncomp[0] != 1)
        return set_err(HF_E_ARG, "hf_estep: Err/Dup/Hap must have one component (hmm_flagger.c:180-182)");
    DevParams* h = ctx->h_params;
    h->model_type = p->model_type; h->n_regions = p->n_regions;
    h->beta_star = ctx->beta_star;
    for (int s = 0; s < 4; s++) h->ncomp[s] = p->ncomp[s];
    for (int pre = 0; pre < 4; pre++)
        for (int s = 0; s < 4; s++) h->alpha[pre * 4 + s] = p->alpha[pre][s];
    for (int s = 0; s < 4; s++) {
        int nu = 0;
        for (int pre = 0; pre < 4; pre++) {
            const double a = p->alpha[pre][s];
            int u = -1;
            for (int k = 0; k < nu; k++) if (h->ualpha[s][k] == a) { u = k; break; }
            if (u < 0) { u = nu; h->ualpha[s][nu++] = a; }
            h->umap[pre * 4 + s] = u;
        }
        for (int k = nu; k < 4; k++) h->ualpha[s][k] = 0.0;
        h->nuniq[s] = nu;
    }
    {   // k_tables' items: every (state, distinct alpha, component) of a row, Err as truncated exponential is one item
        int n = 0;
        for (int st = 0; st < 4; st++) {
            const bool te = st == 0 && p->model_type == HF_MODEL_TRUNC_EXP_GAUSSIAN;
            const int nu = te ? 1 : h->nuniq[st], nc = te ? 1 : p->ncomp[st];
            for (int u = 0; u < 4; u++) h->item_base[st * 4 + u] = u < nu ? n + u * nc : n;
            for (int u = 0; u < nu; u++)
                for (int c = 0; c < nc; c++) { h->item_s[n] = (uint8_t) st; h->item_u[n] = (uint8_t) u; h->item_c[n] = (uint8_t) c; n++; }
        }
        h->n_items = n;
    }
    for (int r = 0; r < ctx->R; r++) {
        DevRegion* g = &h->reg[r];
        std::memcpy(g->trans, p->trans + (size_t) r * 25, sizeof(double) * 25);
        g->lambda = p->lambda ? p->lambda[r] : 1.0;
        g->trunc_point = p->trunc_point ? p->trunc_point[r] : 0.0;
        std::memcpy(g->mean, p->mean + (size_t) r * 4 * HF_MAXCOMP, sizeof(g->mean));
        std::memcpy(g->var, p->var + (size_t) r * 4 * HF_MAXCOMP, sizeof(g->var));
        std::memcpy(g->weight, p->weight + (size_t) r * 4 * HF_MAXCOMP, sizeof(g->weight));
        const double bs = ctx->beta_star;
        for (int s = 0; s < 4; s++)
            for (int c = 0; c < p->ncomp[s]; c++) {             // (per EM step and region: the components in use, not HF_MAXCOMP)
                double var = g->var[s][c];
                var *= bs;
                g->gvar[s][c] = var;
                g->gnorm[s][c] = g->weight[s][c] / (std::sqrt(var * 2 * HF_PI));
                for (int u = 0; u < 4; u++) g->m1[s][u][c] = (1 - h->ualpha[s][u]) * g->mean[s][c];
            }
        g->te_lam = g->lambda / bs;
        g->te_den = 1 - std::exp(-g->te_lam * (bs * g->trunc_point));
        for (int vm = 0; vm < 8; vm++) {
            bool valid[5] = { true, (vm & 1) != 0, true, (vm & 2) != 0, (vm & 4) != 0 };
            for (int pre = 0; pre < 4; pre++) {
                double tot = 0.0;
                for (int s = 0; s < 5; s++) if (valid[s]) tot += g->trans[pre][s];
                for (int s = 0; s < 4; s++) g->tcond[vm][pre * 4 + s] = valid[s] ? g->trans[pre][s] / tot : 0.0;
            }
        }
    }
    return 0;
}


// the 8-byte words of the packed block that are in use, each with its place in the image (hf_device.h KParams); false: they do not fit
static bool pack_kparams(hf_ctx* ctx, const hf_params* p) {
    if (ctx->R != 1) return false;
    const DevParams* h = ctx->h_params;
    KParams& kp = ctx->kparams;
    const char* const base = reinterpret_cast<const char*>(h);
    int nw = 0;
    bool fits = true;
    auto add = [&](const void* first, size_t bytes) {
        const size_t o = (size_t) (reinterpret_cast<const char*>(first) - base);
        if (!fits || bytes == 0) return;
        if (o % 8 || bytes % 8 || nw + (int) (bytes / 8) > HF_KP_MAX_WORDS) { fits = false; return; }
        std::memcpy(kp.data + nw, first, bytes);
        for (size_t i = 0; i < bytes / 8; i++) kp.idx[nw + (int) i] = (uint16_t) (o / 8 + i);
        nw += (int) (bytes / 8);
    };
    const DevRegion* g = &h->reg[0];
    add(h, offsetof(DevParams, reg));                                                    // the header, the item list
    add(g->trans, sizeof g->trans + sizeof g->tcond + 2 * sizeof(double));               // trans, tcond, lambda, trunc_point: contiguous
    static_assert(offsetof(DevRegion, tcond) == sizeof(double) * 25 && offsetof(DevRegion, lambda) == sizeof(double) * (25 + 128) &&
                  offsetof(DevRegion, mean) == sizeof(double) * (25 + 128 + 2), "DevRegion starts with trans | tcond | lambda | trunc_point");
    for (int s = 0; s < HF_NSTATES; s++) {
        const size_t nc = (size_t) p->ncomp[s] * sizeof(double);
        add(g->mean[s], nc); add(g->var[s], nc); add(g->weight[s], nc); add(g->gvar[s], nc); add(g->gnorm[s], nc);
        for (int u = 0; u < 4; u++) add(g->m1[s][u], nc);
    }
    add(&g->te_lam, 2 * sizeof(double));
    if (!fits) return false;
    kp.n_words = nw; kp.pad = 0;
    return true;
}

// everything one pass enqueues on `st` after the parameters were packed into the pinned block
// retry: the pass again after HF_E_RETRY (hf_finish) — the caller's negative_binomial tables went up with the first attempt and
// may have been freed or rewritten since (hf_estep and hf_finish are separate calls): they are NOT read again
static int enqueue_pass(hf_ctx* ctx, const hf_params* p, int mode, hipStream_t st, bool retry = false) {
    // the parameter block: through the kernel arguments of k_tables when it fits them (one region, HF_ALGO_SCAN, Gaussian models:
    // hf_device.h KParams), a copy ahead of the pass otherwise
    ctx->kp_now = ctx->kp_ok && ctx->C > 0 && ctx->algo == HF_ALGO_SCAN && p->model_type != HF_MODEL_NEGATIVE_BINOMIAL && pack_kparams(ctx, p);
    if (!ctx->kp_now) HIPCHK(hipMemcpyAsync(ctx->d_params, ctx->h_params, ctx->params_bytes, hipMemcpyHostToDevice, st));
    for (int i = 0; i < HF_NKERNELS; i++) { ctx->kran[i] = false; ctx->klast_ok[i] = false; }
    ctx->prof_now = ctx->prof_stride <= 1 || (ctx->prof_pass++ % ctx->prof_stride) == 0;
    ctx->pass_rows = false; ctx->pass_bound = false;
    ctx->pass_seg = false;
    if (ctx->C == 0) HIPCHK(hipMemsetAsync(ctx->d_flags, 0, 4, st));
    if (ctx->C > 0) {
        const RowSrc S = row_src(ctx);
        const bool full = mode == HF_MODE_FULL;
        const bool nbm = p->model_type == HF_MODEL_NEGATIVE_BINOMIAL;
        if (nbm) {   // the caller's per-x tables go up with the parameters
            if (!retry && (!p->nb_E || !p->nb_P || !p->nb_dig || !p->nb_r || !p->nb_beta))
                return set_err(HF_E_ARG, "hf_estep: negative_binomial needs hf_params.nb_E/nb_P/nb_dig/nb_r/nb_beta");
            if (p->nb_max_x > 0 && ctx->M - 1 > p->nb_max_x)
                return set_err(HF_E_ARG, "hf_estep: the windows hold coverage values above hf_params.nb_max_x (hfm_set_max_coverage)");
            const size_t nE = (size_t) ctx->R * 4 * HF_NB_NX * 8, nP = (size_t) ctx->R * 4 * ctx->K * HF_NB_NX * 8,
                         nR = (size_t) ctx->R * 4 * ctx->K * 8;
            // ONE device buffer and ONE copy per pass for the five tables (round 4; five copies out of pageable memory were ~50 us of a
            // 0.24 ms step): E | P | dig | r | beta, gathered into one of two pinned staging buffers (an event per buffer says when its
            // last copy has left it: a caller may enqueue the next pass before this one has run)
            const size_t nAll = nE + 2 * nP + 2 * nR;
            if (!ctx->d_nbE) {
                // everything into locals first: the context is touched only when ALL of it exists (a failure half way used to leave
                // d_nbE set and a staging buffer null — the next pass skipped this block and copied through the null pointer)
                double *dE = nullptr, *dH = nullptr; char* hb[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr};
                bool ok = hipMalloc((void**) &dE, nAll) == hipSuccess &&
                          hipMalloc((void**) &dH, ((size_t) ctx->ntiles * ctx->R * HF_NB_TILE_VEC + 1) * 8) == hipSuccess;
                for (int b = 0; ok && b < 2; b++)
                    ok = hipHostMalloc((void**) &hb[b], nAll) == hipSuccess && hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) == hipSuccess;
                if (!ok) {
                    const hipError_t e_ = hipGetLastError();
                    for (int b = 0; b < 2; b++) { if (hb[b]) hipHostFree(hb[b]); if (ev[b]) hipEventDestroy(ev[b]); }
                    hipFree(dE); hipFree(dH);
                    return set_err(HF_E_HIP, std::string("hf_estep: negative-binomial staging: ") + hipGetErrorString(e_));
                }
                char* base = reinterpret_cast<char*>(dE);
                ctx->d_nbE = dE; ctx->d_tile_hist = dH;
                ctx->d_nbP = reinterpret_cast<double*>(base + nE); ctx->d_nbDig = reinterpret_cast<double*>(base + nE + nP);
                ctx->d_nbR = reinterpret_cast<double*>(base + nE + 2 * nP); ctx->d_nbBeta = reinterpret_cast<double*>(base + nE + 2 * nP + nR);
                for (int b = 0; b < 2; b++) { ctx->h_nb[b] = hb[b]; ctx->nb_ev[b] = ev[b]; }
            }
            if (!retry) {
                const int b = (int) (ctx->nb_turn++ & 1u);
                if (ctx->nb_ev_used[b]) HIPCHK(hipEventSynchronize(ctx->nb_ev[b]));
                char* h = ctx->h_nb[b];
                std::memcpy(h, p->nb_E, nE); std::memcpy(h + nE, p->nb_P, nP); std::memcpy(h + nE + nP, p->nb_dig, nP);
                std::memcpy(h + nE + 2 * nP, p->nb_r, nR); std::memcpy(h + nE + 2 * nP + nR, p->nb_beta, nR);
                HIPCHK(hipMemcpyAsync(ctx->d_nbE, h, nAll, hipMemcpyHostToDevice, st));
                HIPCHK(hipEventRecord(ctx->nb_ev[b], st));
                ctx->nb_ev_used[b] = true;
            }
        }
        {   // also clears the flag word: first kernel of every pass
            KTimer t(ctx, st, HF_K_TABLES);
            // statistics by emission row: the job list of the Gaussian tables is the (key, class) list of the rows of A (hf_seg.h)
            const bool arows = !nbm && seg_pass(ctx);
            const int nk = arows ? ctx->n_combo : ctx->n_keys;
            const int jobs = nk + ctx->n_slow;
            if (nbm)
                hipLaunchKernelGGL(k_tables_nb, dim3((unsigned) (jobs / 256 + 1)), dim3(256), 0, st, ctx->n_keys, ctx->d_keys, ctx->n_slow,
                                   ctx->d_slow_w, ctx->d_rec, ctx->M, ctx->d_nbE, ctx->d_lutE, ctx->d_Es, ctx->d_flags);
            else
            {
#define HF_LAUNCH_TABLES(J, KA, KP) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tables<J, KA>), dim3((unsigned) ((jobs + J - 1) / J + (jobs == 0))), dim3(256), 0, st, \
                               ctx->tabwork, ctx->d_params, ctx->d_flags, KP, ctx->d_params)
                if (ctx->kp_now) {     // the parameter block travels in the kernel arguments (pack_kparams): no copy was enqueued
                    if (jobs < 64 * 1024) HF_LAUNCH_TABLES(HF_TABLE_JOBS_SMALL, true, ctx->kparams); else HF_LAUNCH_TABLES(HF_TABLE_JOBS_LARGE, true, ctx->kparams);
                } else {
                    const NoKParams none{0};
                    if (jobs < 64 * 1024) HF_LAUNCH_TABLES(HF_TABLE_JOBS_SMALL, false, none); else HF_LAUNCH_TABLES(HF_TABLE_JOBS_LARGE, false, none);
                }
#undef HF_LAUNCH_TABLES
            }
        }
        if (ctx->ntiles > 0) {
            if (ctx->algo == HF_ALGO_SEQ) {
                {
                    KTimer t(ctx, st, HF_K_EMIT_ROWS);
                    hipLaunchKernelGGL(k_emit_rows, dim3((unsigned) ((ctx->N + 255) / 256)), dim3(256), 0, st, ctx->N, ctx->d_rec,
                                       ctx->d_beta, ctx->d_params, nbm ? ctx->d_nbE : (const double*) nullptr, ctx->d_E, ctx->d_flags);
                }
                {
                    KTimer t(ctx, st, HF_K_FWD_SEQ);
                    hipLaunchKernelGGL(k_fwd_seq, dim3((unsigned) ctx->C), dim3(64), 0, st, ctx->d_off, ctx->d_chunk_tile0, ctx->d_rec,
                                       ctx->d_E, ctx->d_params, ctx->d_f, ctx->d_scale, ctx->d_tile_ll, ctx->d_flags);
                }
                if (full) ctx->fb_recs = false;
                if (full) {
                    KTimer t(ctx, st, HF_K_BWD_SEQ);
                    hipLaunchKernelGGL(k_bwd_seq, dim3((unsigned) ctx->C), dim3(64), 0, st, ctx->d_off, ctx->d_chunk_tile0, ctx->d_rec,
                                       ctx->d_E, ctx->d_params, ctx->d_f, ctx->d_scale, ctx->d_b, ctx->d_label, ctx->d_flags);
                }
            } else if (seg_pass(ctx)) {
                // one workgroup per chunk segment does the whole forward-backward (hf_seg.h)
                const int nc = ctx->seg_fused ? ctx->seg_nc : 0;
                const size_t lds = seg_lds_bytes(nc);
                if (ctx->host_trace && !ctx->ht_n) {
                    int o1 = 0, o2 = 0;
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&o1, k_seg_prod, 64, seg_lds_bytes());
                    if (ctx->seg_fused) hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, k_seg_fb<true, true>, 64, lds);
                    else hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, k_seg_fb<true, false>, 64, lds);
                    std::fprintf(stderr, "[hf host trace] segment kernels (%s): %d workgroups of 64 threads, %zu B of LDS; resident per CU: k_seg_prod %d, k_seg_fb %d\n",
                                 ctx->seg_fused ? "one launch" : "two launches", ctx->nseg, lds, o1, o2);
                }
                if (nbm) {   // k_tables_nb leaves the emission rows; the Gaussian k_tables writes the rows of A itself
                    KTimer t(ctx, st, HF_K_AROWS);
                    hipLaunchKernelGGL(k_arows, dim3((unsigned) (((int64_t) ctx->n_arows * 16 + 255) / 256)), dim3(256), 0, st, ctx->n_arows,
                                       ctx->d_arow_src, ctx->d_arow_cls, ctx->d_lutE, ctx->d_params, ctx->d_lutA);
                }
                if (!ctx->seg_fused) {
                    if (!ctx->d_segQ) HIPCHK(hipMalloc((void**) &ctx->d_segQ, (size_t) ctx->nseg * 64 * 16 * 8));   // lane products: two-launch mode only
                    KTimer t(ctx, st, HF_K_SEG_PROD);
                    hipLaunchKernelGGL(k_seg_prod, dim3((unsigned) ctx->nseg), dim3(64), seg_lds_bytes(), st, ctx->d_seg, ctx->d_arow, ctx->d_lutA, ctx->d_segQ, ctx->d_Pseg);
                }
                const bool tfb = ((ctx->prof_mask >> HF_K_SEG_FB) & 1u) && ctx->prof_now;
                if (tfb) ctx->kran[HF_K_SEG_FB] = true;
                const unsigned epoch = ++ctx->seg_epoch;
                // HF_SEG_TEST_TIMEOUT=1 (tests/test_estep_gpu.py): the first one-launch pass waits for flags nobody writes, so that
                // the time-out, HF_E_RETRY and the fall-back to two launches are exercised
                const unsigned wait_epoch = (ctx->seg_test_timeout && epoch == 1) ? 0xffffffffu : epoch;
                // Sub-passes (hf_ctx::SubPass).  A full pass whose statistics go by emission row runs sub-pass by sub-pass through the pass
                // buffer: k_seg_fb writes a sub-pass's records, k_pair_sums reads them back while they are still in the Infinity Cache, the next
                // sub-pass overwrites them.  Any other pass (per-chunk statistics read the records by window afterwards; a forward-only pass writes
                // none) is ONE launch over all segments into the all-windows buffer.
                const bool want_rows = full && rows_pass(ctx);
                ctx->pass_pairs_done = false;
                if (want_rows) {
                    const bool alias = ctx->subs.size() > 1;
                    bool first = true;
                    for (size_t si = 0; si < ctx->subs.size(); si++) {
                        const auto& sb = ctx->subs[si];
                        double* const recs_eff = alias ? ctx->d_recs - sb.p0 * 8 : ctx->d_recs;
                        launch_seg_fb(ctx, st, true, recs_eff, (int) si, epoch, wait_epoch, tfb && first, false);   // (timed: the first sub-pass's launch, hf_sub_pass_windows; no scales)
                        first = false;
                        KTimer t(ctx, st, HF_K_PAIR_SUMS);              // (event pairs around k_pair_sums: the last sub-pass's is what is read)
                        launch_pair_sums(ctx, st, sb, recs_eff);
                    }
                    ctx->pass_pairs_done = true;
                    ctx->recs_all = !alias;
                    ctx->scales_all = false;
                } else {
                    double* recs = ctx->d_recs;
                    if (full && ctx->subs.size() > 1) { const int rc_ = all_records_buffer(ctx); if (rc_) return rc_; recs = ctx->d_recs_all; }
                    launch_seg_fb(ctx, st, full, recs, -1, epoch, wait_epoch, tfb, false);
                    if (full) { ctx->recs_all = true; ctx->scales_all = false; }
                }
                ctx->pass_seg = true;
                if (full) ctx->fb_recs = true;
            } else
                return set_err(HF_E_ARG, "hf_estep: HF_ALGO_SCAN holds at most 2^30 windows per context (shard the chunk list: hmm_flagger_multi.h)");
        }
        const int kc = p->ncomp[3];
        const int fl = full && ctx->ntiles > 0;
        ctx->pass_nb = false;
        if (nbm && fl && rows_pass(ctx)) {   // statistics by emission row, negative_binomial (hf_nb_rows.h)
            {   // (k_pair_sums ran behind every sub-pass's k_seg_fb, above)
                KTimer t(ctx, st, HF_K_ROW_STATS);
                const int n_rw_blocks = (ctx->n_rowwaves + 3) / 4, n_ll_blocks = (ctx->C + 3) / 4;
                hipLaunchKernelGGL(k_row_stats_nb, dim3((unsigned) (n_rw_blocks + n_ll_blocks)), dim3(256), 0, st, ctx->n_rowwaves, n_rw_blocks,
                                   ctx->d_rowslots, ctx->d_grp_sums, ctx->d_slot_h, ctx->d_rw_stats, ctx->C, ll_off(ctx), ll_part(ctx),
                                   ctx->d_chunk_stats, ctx->V, ctx->d_chunk_ll, ctx->rs_bpw);
                const int n_bins = ctx->R * 256;
                hipLaunchKernelGGL(k_nb_hist, dim3((unsigned) ((n_bins + 3) / 4)), dim3(256), 0, st, n_bins, ctx->d_bin_off, ctx->d_bin_list,
                                   ctx->d_slot_h, ctx->d_H);
            }
            ctx->pass_rows = true; ctx->pass_nb = true; ctx->pass_wpb = 4;
        }
        else if (nbm) {
            if (fl) {
                KTimer t(ctx, st, HF_K_STATS_TILE);
                const TileGeom g = tile_geom(ctx, k_stats_tile_nb<HF_SCAN_L>, (size_t) HF_NB_WAVE_LDS * 8);
                if (!g.ok) return set_err(HF_E_ARG, "the per-region tables do not fit the LDS of one workgroup");
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stats_tile_nb<HF_SCAN_L>), dim3(g.blocks), dim3(g.threads), g.lds, st, ctx->ntiles, ctx->d_tile_desc,
                                   ctx->d_rec, S, ctx->d_params, ctx->pass_seg ? ctx->d_recs_all : ctx->d_f, ctx->d_b, ctx->d_regmask, ctx->d_tile_hist,
                                   ctx->pass_seg ? ctx->d_pos : (const int32_t*) nullptr);
            }
            NbTables nt;
            nt.E = ctx->d_nbE; nt.P = ctx->d_nbP; nt.dig = ctx->d_nbDig; nt.r = ctx->d_nbR; nt.beta = ctx->d_nbBeta;
            KTimer t(ctx, st, HF_K_CHUNK_STATS);
            hipLaunchKernelGGL(k_chunk_stats_nb, dim3((unsigned) ctx->C), dim3(256), 0, st, ctx->d_chunk_tile0, ll_off(ctx), ctx->d_regmask,
                               ctx->d_tile_hist, ll_part(ctx), ctx->d_params, nt, ctx->d_chunk_stats, ctx->V, ctx->K, fl);
        }
        else if (kc <= 4) launch_stats<4>(ctx, st, fl, kc);
        else if (kc <= 8) launch_stats<8>(ctx, st, fl, kc);
        else launch_stats<16>(ctx, st, fl, kc);
        if (ctx->launch_failed) { ctx->launch_failed = false; return HF_E_ARG; }
    }
    HIPCHK(hipGetLastError());
    return HF_OK;
}

int hf_estep(hf_ctx* ctx, const hf_params* p, int mode, void* stream) {
    if (!ctx || !p || (mode != HF_MODE_FULL && mode != HF_MODE_FORWARD_ONLY)) return set_err(HF_E_ARG, "hf_estep: bad argument");
    hipStream_t st = (hipStream_t) stream;
    HIPCHK(hipSetDevice(ctx->device));
    int rc = pack_params(ctx, p);
    if (rc) return rc;
    ctx->last_p = *p; ctx->last_mode = mode; ctx->last_stream = st;
    if (pass_events(ctx)) HIPCHK(hipEventRecord(ctx->ev0, st));
    rc = enqueue_pass(ctx, p, mode, st);
    if (rc) return rc;
    if (pass_events(ctx)) HIPCHK(hipEventRecord(ctx->ev1, st));
    ctx->ev_valid = pass_events(ctx);
    ctx->have_full = (mode == HF_MODE_FULL);
    return HF_OK;
}

int32_t hf_n_chunks(const hf_ctx* ctx) { return ctx ? ctx->C : 0; }
int64_t hf_n_windows(const hf_ctx* ctx) { return ctx ? ctx->N : 0; }
int64_t hf_chunk_stats_len(const hf_ctx* ctx) { return ctx ? ctx->V : 0; }
double* hf_chunk_stats_dev(hf_ctx* ctx) { return ctx ? ctx->d_chunk_stats : nullptr; }
int8_t* hf_labels_dev(hf_ctx* ctx) { return ctx ? ctx->d_label : nullptr; }

int hf_set_stats_mode(hf_ctx* ctx, int mode) {
    if (!ctx || (mode != HF_STATS_CHUNKS && mode != HF_STATS_ROWS)) return set_err(HF_E_ARG, "hf_set_stats_mode: bad argument");
    ctx->stats_mode = mode;
    return HF_OK;
}
int hf_get_stats_mode(const hf_ctx* ctx) {
    return ctx && ctx->stats_mode == HF_STATS_ROWS && ctx->rows_ready && ctx->algo == HF_ALGO_SCAN ? HF_STATS_ROWS : HF_STATS_CHUNKS;
}

int hf_seg_launches(const hf_ctx* ctx) { return !ctx || !seg_pass(ctx) ? 0 : (ctx->seg_fused ? 1 : 2); }
int hf_sub_passes(const hf_ctx* ctx) { return !ctx || !seg_pass(ctx) ? 0 : (int) ctx->subs.size(); }
int64_t hf_sub_pass_windows(const hf_ctx* ctx, int k) {
    if (!ctx || k < 0 || (size_t) k >= ctx->subs.size()) return 0;
    return ctx->h_off[(size_t) ctx->subs[(size_t) k].c1] - ctx->h_off[(size_t) ctx->subs[(size_t) k].c0];
}
int hf_seg_cached_steps(const hf_ctx* ctx) { return !ctx || !seg_pass(ctx) || !ctx->seg_fused ? 0 : ctx->seg_nc; }
int hf_seg_xcd_plan(const hf_ctx* ctx) { return ctx && seg_pass(ctx) && ctx->seg_fused && ctx->d_seg_of_block ? 1 : 0; }
int64_t hf_seg_block_table(const hf_ctx* ctx, int32_t* seg_of_block, int64_t n) {
    if (!ctx) return 0;
    const int64_t have = (int64_t) ctx->h_seg_of_block.size();
    if (seg_of_block) for (int64_t i = 0; i < n && i < have; i++) seg_of_block[i] = ctx->h_seg_of_block[(size_t) i];
    return have;
}
int hf_create_phases(const hf_ctx* ctx, int max, double* ms, const char** names) {
    if (!ctx) return 0;
    const int n = (int) ctx->create_phases.size();
    for (int i = 0; i < n && i < max; i++) { if (ms) ms[i] = ctx->create_phases[(size_t) i].second; if (names) names[i] = ctx->create_phases[(size_t) i].first; }
    return n;
}

int hf_copy_chunk_stats(hf_ctx* ctx, double* dst_dev, void* stream) {
    if (!ctx || !dst_dev) return set_err(HF_E_ARG, "hf_copy_chunk_stats: bad argument");
    if (ctx->pass_rows) return set_err(HF_E_ARG, "hf_copy_chunk_stats: the last pass ran in HF_STATS_ROWS mode (no per-chunk vectors)");
    HIPCHK(hipSetDevice(ctx->device));
    if (dst_dev == ctx->d_chunk_stats) return HF_OK;   // bound to the exchange buffer: the kernels already wrote in place
    HIPCHK(hipMemcpyAsync(dst_dev, ctx->d_chunk_stats, (size_t) ctx->C * ctx->V * 8, hipMemcpyDeviceToDevice,
                          (hipStream_t) stream));
    return HF_OK;
}

int hf_rank_total(hf_ctx* ctx, double* out_dev, void* stream) {
    if (!ctx || !out_dev) return set_err(HF_E_ARG, "hf_rank_total: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->pass_rows && ctx->pass_nb) return launch_nb_total(ctx, (hipStream_t) stream, out_dev, false);
    if (ctx->pass_rows && ctx->pass_bound && out_dev == ctx->d_rank_out) return HF_OK;   // hf_bind_rank_total: already there
    if (ctx->pass_rows && ctx->pass_host_total) {   // the pass sent its partials to the host: the total on the device from the copies in blk_stats / chunk_ll
        const int wpb = ctx->pass_wpb;
        const dim3 grid((unsigned) (ctx->R + 1)), blk((unsigned) (64 * wpb));
#define HF_LATE(KT) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rows_total_late<KT>), grid, blk, 0, (hipStream_t) stream, ctx->d_rw_off, wpb, ctx->d_rw_stats, ctx->d_params, \
                                       ctx->K, ctx->d_chunk_ll, (int64_t) ctx->C, out_dev, done_scratch(ctx->d_done))
        if (ctx->pass_kc <= 4) HF_LATE(4); else if (ctx->pass_kc <= 8) HF_LATE(8); else HF_LATE(16);
#undef HF_LATE
        HIPCHK(hipGetLastError());
        return HF_OK;
    }
    if (ctx->pass_rows) {   // the pass left its total in d_total (or where it was bound to)
        hipLaunchKernelGGL(k_copy_total, dim3(1), dim3(256), 0, (hipStream_t) stream, ctx->pass_bound ? ctx->d_rank_out : ctx->d_total, out_dev, ctx->V);
        HIPCHK(hipGetLastError());
        return HF_OK;
    }
    return hf_reduce_chunks_indexed(ctx, ctx->d_chunk_stats, nullptr, ctx->C, out_dev, stream);
}

struct ExchangeGeom { int n_ranks = 0, rows_per_rank = 0, flag_row = 0; };   // n_ranks == 0: no flag rows to merge
static int reduce_chunks_seq(hf_ctx* ctx, const double* chunk_stats_dev, const int32_t* row_index_dev, int64_t n_chunks,
                             double* out_dev, void* stream, double seq, ExchangeGeom xg = ExchangeGeom()) {
    if (!ctx || !chunk_stats_dev || !out_dev || n_chunks < 0) return set_err(HF_E_ARG, "hf_reduce_chunks: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    const bool own = chunk_stats_dev == ctx->d_chunk_stats;
    const unsigned keep = ctx->prof_mask;
    if (!own) ctx->prof_mask = 0;
    {
        KTimer t(ctx, (hipStream_t) stream, HF_K_REDUCE);
        hipLaunchKernelGGL(k_reduce, dim3((unsigned) (ctx->V + 1)), dim3(64), 0, (hipStream_t) stream, chunk_stats_dev, row_index_dev,
                           n_chunks, ctx->V, out_dev,
                           (out_dev == ctx->d_total || out_dev == ctx->d_total_host) ? ctx->d_flags : (const unsigned*) nullptr,
                           xg.n_ranks, xg.rows_per_rank, xg.flag_row, seq, ctx->d_done, ctx->d_cks);
    }
    ctx->prof_mask = keep;
    HIPCHK(hipGetLastError());
    return HF_OK;
}
int hf_reduce_chunks_indexed(hf_ctx* ctx, const double* chunk_stats_dev, const int32_t* row_index_dev, int64_t n_chunks,
                             double* out_dev, void* stream) {
    return reduce_chunks_seq(ctx, chunk_stats_dev, row_index_dev, n_chunks, out_dev, stream, 0.0);
}

int hf_reduce_chunks(hf_ctx* ctx, const double* chunk_stats_dev, int64_t n_chunks, double* out_dev, void* stream) {
    return hf_reduce_chunks_indexed(ctx, chunk_stats_dev, nullptr, n_chunks, out_dev, stream);
}

// running sums for hf_kernel_time_sums; the stream has been synchronised
static void accumulate_kernel_times(hf_ctx* ctx) {
    if (!ctx->prof_mask) return;
    for (int i = 0; i < HF_NKERNELS; i++)
        if (ctx->kran[i]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->kev[2 * i], ctx->kev[2 * i + 1]) == hipSuccess) { ctx->ksum[i] += ms; ctx->kcount[i]++; ctx->klast[i] = ms; ctx->klast_ok[i] = true; }
        }
}

static int flags_to_code(unsigned fl) {
    if (fl & HF_FLAG_REGION) return set_err(HF_E_REGION, "a window's region index is >= n_regions");
    if (fl & HF_FLAG_SYNC) return set_err(HF_E_HIP, "a chunk segment waited too long for another segment's product (one-launch segment kernel)");
    if (fl & HF_FLAG_NAN) return set_err(HF_E_NAN, "[Error] prob is NAN");
    if (fl & HF_FLAG_SCALE) return set_err(HF_E_SCALE, "scale is very low!");
    return HF_OK;
}

int hf_check(hf_ctx* ctx, void* stream) {
    if (!ctx) return set_err(HF_E_ARG, "hf_check: bad argument");
    hipStream_t st = (hipStream_t) stream;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    accumulate_kernel_times(ctx);
    if ((*ctx->h_flags & HF_FLAG_SYNC) && ctx->seg_fused) {   // as hf_finish: two launches from now on, the caller repeats the pass
        ctx->seg_fused = false;
        std::fprintf(stderr, "[hmm_flagger_hip] one-launch segment kernel: a hand-off timed out; this context falls back to k_seg_prod + k_seg_fb\n");
        return set_err(HF_E_RETRY, "a hand-off of the one-launch segment kernel timed out: the context now runs two launches, repeat hf_estep");
    }
    return flags_to_code(*ctx->h_flags);
}

// Completion of a pass.  Default: hipStreamSynchronize on the launch stream — the HIP-defined way to see the statistics
// block that the last kernel wrote into pinned host memory.
// Opt-in (environment HF_POLL=1): the last kernel also writes a completion stamp behind a system-scope fence and the host
// spins on it instead of waiting for the stream's completion signal (3-12 us less per EM step, depending on the box).  The
// order in which device writes to host memory become visible is NOT defined by the HIP memory model: round 1 measured a
// stale element in one of ~2 000 passes with the stamp alone, so a polled block is only accepted when a position-weighted
// checksum of what the host READ matches the one the kernel wrote (polled_block_consistent; none in 800 000 passes,
// profiles/tools/poll_soak.py), and the memory must be coherent host memory (HIP_HOST_COHERENT=0 falls back to the
// stream after the 2 s bail-out).  HF_POLL=debug additionally synchronises after acceptance and reports any word
// that still changed.  It stays off by default: no EM step should depend on a probabilistic check.
// The stamp alone is not enough: the device's writes to host memory may become visible out of order (a soak test saw a
// stale element once in ~2 000 passes).  So the kernel also writes a checksum of everything it wrote (hf_cks_term: bit
// pattern x position weight, summed mod 2^64) plus the pass's stamp value; the host accepts the block only when what it
// READ has that checksum — stale data, a stale checksum or both fail the comparison and the host keeps polling.  (A plain
// XOR was not enough either: the estimator layout repeats values, and two equal stale words cancel — seen once in 300 000.)
static bool polled_block_consistent(const hf_ctx* ctx) {
    const double* h = ctx->h_total;
    const int64_t V = ctx->V;
    auto bits = [](double d) { unsigned long long u; std::memcpy(&u, &d, 8); return u; };
    const unsigned long long s = bits(ctx->poll_seq);
    if (ctx->poll_kind == 2) {
        unsigned long long x = s;
        for (int64_t v = 0; v <= V; v++) x += hf_cks_term(bits(h[v]), v);
        return x == bits(h[V + 2]);
    }
    const int64_t rstride = 24 * (int64_t) ctx->K + 16;
    for (int r = 0; r < ctx->R; r++) {
        unsigned long long x = s;
        for (int64_t v = 0; v < rstride; v++) { const int64_t at = 1 + r * rstride + v; x += hf_cks_term(bits(h[at]), at); }
        if (r == 0) x += hf_cks_term(bits(h[0]), 0) + hf_cks_term(bits(h[V]), V);
        if (x != bits(h[V + 2 + r])) return false;
    }
    return true;
}
// The total of a pass whose blocks wrote their partial vectors to the host (hf_rows.h part_host): rows_total_region and rows_total_ll restated —
// the same additions in the same order (nq interleaved accumulators per element over the region's blocks, then their sum; the log-likelihood
// by 64 strided sums and a halving tree), so h_total holds the bits the launch's own last blocks would have written.
extern "C++" {
template <int KT>
static void host_rows_total_t(hf_ctx* ctx) {
    constexpr int NA = 16 + 9 + 2 + 3 * KT + 1;
    const double* part = ctx->h_part;
    const int nblk = ctx->pass_rw_blocks, wpb = ctx->pass_wpb, nt = 64 * wpb, Kctx = ctx->K, nreg = ctx->R;
    const int64_t V = ctx->V, C = ctx->C;
    const DevParams* P = ctx->h_params;
    const int ncol = P->ncomp[3];
    const bool te = P->model_type == HF_MODEL_TRUNC_EXP_GAUSSIAN;
    const int rstride = 24 * Kctx + 16;
    double* out = ctx->h_total;
    int nq = nt / NA;
    nq = nq < 1 ? 1 : (nq > 16 ? 16 : nq);
    for (int r = 0; r < nreg; r++) {
        double* dst = out + 1 + (int64_t) r * rstride;
        for (int v = 0; v < rstride; v++) dst[v] = 0.0;
        const int w0 = ctx->h_rw_off[(size_t) r] / wpb, w1 = ctx->h_rw_off[(size_t) r + 1] / wpb;
        if (w1 <= w0) continue;
        double red[NA], acc[16][NA];
        for (int q = 0; q < nq; q++) for (int i = 0; i < NA; i++) acc[q][i] = 0.0;
        for (int k = w0; k < w1; k++) {                              // (blocks in order, whole vectors: accumulator (k - w0) % nq takes block k, as on the device)
            double* a = acc[(k - w0) % nq];
            const double* pk = part + (size_t) k * NA;
            for (int i = 0; i < NA; i++) a[i] += pk[i];
        }
        for (int i = 0; i < NA; i++) {
            double tot = 0.0;
            for (int q = 0; q < nq; q++) tot += acc[q][i];
            red[i] = tot;
        }
        const StatAcc<KT>* Sa = reinterpret_cast<const StatAcc<KT>*>(red);
        for (int t = 0; t < 16; t++) dst[24 * Kctx + t] = Sa->trans[t];
        if (te) { dst[(0 * 2 + 0) * Kctx] = Sa->te_num; dst[(0 * 2 + 1) * Kctx] = Sa->te_den; }
        for (int st = 0; st < 3; st++)
            if (!(st == 0 && te)) {
                double* dd = dst + (st * 3) * 2 * Kctx;
                dd[(0 * 2 + 0) * Kctx] = Sa->g_mnum[st]; dd[(0 * 2 + 1) * Kctx] = Sa->g_den[st];
                dd[(1 * 2 + 0) * Kctx] = Sa->g_vnum[st]; dd[(1 * 2 + 1) * Kctx] = Sa->g_den[st];
                dd[(2 * 2 + 0) * Kctx] = Sa->g_den[st];  dd[(2 * 2 + 1) * Kctx] = Sa->g_den[st];
            }
        for (int cc = 0; cc < KT && cc < ncol; cc++) {
            double* dd = dst + (3 * 3) * 2 * Kctx;
            dd[(0 * 2 + 0) * Kctx + cc] = Sa->c_mnum[cc]; dd[(0 * 2 + 1) * Kctx + cc] = Sa->c_den[cc];
            dd[(1 * 2 + 0) * Kctx + cc] = Sa->c_vnum[cc]; dd[(1 * 2 + 1) * Kctx + cc] = Sa->c_den[cc];
            dd[(2 * 2 + 0) * Kctx + cc] = Sa->c_den[cc];  dd[(2 * 2 + 1) * Kctx + cc] = Sa->c_wden;
        }
    }
    {   // the log-likelihood: rows_total_ll's order
        const double* ll = part + (size_t) nblk * NA;
        double a[64];
        for (int l = 0; l < 64; l++) { double acc = 0.0; for (int64_t c = l; c < C; c += 64) acc += ll[c]; a[l] = acc; }
        for (int o = 32; o > 0; o >>= 1) for (int l = 0; l < o; l++) a[l] += a[l + o];
        out[0] = a[0];
    }
    out[V] = part[(size_t) nblk * NA + (size_t) C];      // the flag word
}
}
static void host_rows_total(hf_ctx* ctx) {
    if (ctx->pass_kc <= 4) host_rows_total_t<4>(ctx); else if (ctx->pass_kc <= 8) host_rows_total_t<8>(ctx); else host_rows_total_t<16>(ctx);
}

static int wait_total(hf_ctx* ctx, hipStream_t st, bool polled, double* stats_host, bool host_total = false) {
    bool seen = false;
    if (polled) {
        volatile double* stamp = ctx->h_total + ctx->V + 1;
        const auto t0 = std::chrono::steady_clock::now();
        long spins = 0;
        for (;;) {
            if (*stamp == ctx->poll_seq) {
                std::atomic_thread_fence(std::memory_order_acquire);
                if (polled_block_consistent(ctx)) { seen = true; break; }
            }
            __builtin_ia32_pause();
            if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;   // let the stream report
        }
    }
    if (!seen && ctx->stream_stamp_ok) {
        // Completion through the stream itself: a 32-bit value that the command processor writes into the pinned block AFTER
        // everything enqueued before it has completed (hipStreamWriteValue32: a stream-ordered memory operation, so — unlike a
        // stamp written by the kernel, HF_POLL — its order against the kernel's own writes is defined), polled by the host.
        // Measured 4-5 us less per EM step than hipStreamSynchronize.  Bail-out: the stream after 2 s.
        volatile uint32_t* word = reinterpret_cast<volatile uint32_t*>(ctx->h_flags) + 1;
        const uint32_t want = ++ctx->stream_stamp;
        if (hipStreamWriteValue32(st, (void*) word, want, 0) == hipSuccess) {
            const auto t0 = std::chrono::steady_clock::now();
            long spins = 0;
            while (*word != want) {
                __builtin_ia32_pause();
                if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            seen = *word == want;
        } else { (void) hipGetLastError(); ctx->stream_stamp_ok = false; }
        if (seen) {
            accumulate_kernel_times(ctx);
            if (host_total) host_rows_total(ctx);
            std::memcpy(stats_host, ctx->h_total, (size_t) ctx->V * 8);
            const unsigned fl2 = (unsigned) ctx->h_total[ctx->V];
            if ((fl2 & HF_FLAG_SYNC) && ctx->seg_fused) {
                ctx->seg_fused = false;
                std::fprintf(stderr, "[hmm_flagger_hip] one-launch segment kernel: a hand-off timed out; this context falls back to k_seg_prod + k_seg_fb\n");
                return HF_E_RETRY;
            }
            return flags_to_code(fl2);
        }
    }
    if (!seen) {
        HIPCHK(hipStreamSynchronize(st));
        // the stream has drained and the stamp still is not there: hipStreamWriteValue32 returns success on this stack but the write
        // does not land where the host polls — stop paying the 2 s bail-out in every later pass
        if (ctx->stream_stamp_ok && ctx->stream_stamp != 0 &&
            *(reinterpret_cast<volatile uint32_t*>(ctx->h_flags) + 1) != ctx->stream_stamp) {
            ctx->stream_stamp_ok = false;
            std::fprintf(stderr, "[hmm_flagger_hip] the stream's completion stamp did not arrive; passes complete through hipStreamSynchronize from now on\n");
        }
    }
    if (seen && poll_mode() == 2) {   // HF_POLL=debug: did anything still arrive after the block was accepted?
        std::vector<double> snap(ctx->h_total, ctx->h_total + ctx->V + 1);
        HIPCHK(hipStreamSynchronize(st));
        for (int64_t v = 0; v <= ctx->V; v++)
            if (std::memcmp(&snap[(size_t) v], &ctx->h_total[v], 8) != 0)
                std::fprintf(stderr, "[poll debug] seq %.0f kind %d element %ld changed after acceptance: %.17g -> %.17g\n", ctx->poll_seq,
                             ctx->poll_kind, (long) v, snap[(size_t) v], ctx->h_total[v]);
    }
    accumulate_kernel_times(ctx);   // events of the kernels before the last one have completed
    if (host_total) host_rows_total(ctx);
    std::memcpy(stats_host, ctx->h_total, (size_t) ctx->V * 8);
    const unsigned fl = (unsigned) ctx->h_total[ctx->V];
    if ((fl & HF_FLAG_SYNC) && ctx->seg_fused) {   // the one-launch segment kernel gave up a wait: this context runs two launches from now on
        ctx->seg_fused = false;
        std::fprintf(stderr, "[hmm_flagger_hip] one-launch segment kernel: a hand-off timed out; this context falls back to k_seg_prod + k_seg_fb\n");
        return HF_E_RETRY;
    }
    return flags_to_code(fl);
}

int hf_finish(hf_ctx* ctx, double* stats_host, void* stream) {
    if (!ctx || !stats_host) return set_err(HF_E_ARG, "hf_finish: bad argument");
    hipStream_t st = (hipStream_t) stream;
    HIPCHK(hipSetDevice(ctx->device));
    // the reduction writes the V+1 doubles into pinned host memory over PCIe: no device-to-host copy afterwards
    double* out = ctx->d_total_host ? ctx->d_total_host : ctx->d_total;
    const bool own_total = ctx->pass_rows && !ctx->pass_nb;   // k_row_stats already wrote the total (and the stamp, when polled)
    const bool polled = own_total ? ctx->pass_polled : poll_ok(ctx, ctx->pass_rows ? HF_K_ROWS_TOTAL : HF_K_REDUCE);
    const double seq = own_total ? 0.0 : (polled ? next_stamp(ctx) : 0.0);
    ctx->poll_kind = ctx->pass_rows ? 1 : 2;
    int rc = own_total ? HF_OK
                       : (ctx->pass_rows ? launch_nb_total(ctx, st, out, true, seq)
                                         : reduce_chunks_seq(ctx, ctx->d_chunk_stats, nullptr, ctx->C, out, stream, seq));
    if (rc) return rc;
    if (pass_events(ctx)) HIPCHK(hipEventRecord(ctx->ev1, st));
    if (own_total && ctx->pass_bound) {   // the pass wrote its total into a bound exchange slot (hf_bind_rank_total): fetch it from there
        HIPCHK(hipMemcpyAsync(ctx->h_total, ctx->d_rank_out, (size_t) ctx->V * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(ctx->h_total + ctx->V, ctx->d_rank_flag, 8, hipMemcpyDeviceToHost, st));
    } else if (!ctx->d_total_host)
        HIPCHK(hipMemcpyAsync(ctx->h_total, ctx->d_total, ((size_t) ctx->V + 1) * 8, hipMemcpyDeviceToHost, st));
    rc = wait_total(ctx, st, polled, stats_host, own_total && ctx->pass_host_total);
    if (rc == HF_E_RETRY) {   // the pass again, in two launches (the packed parameters are still in the pinned block)
        if (pass_events(ctx)) HIPCHK(hipEventRecord(ctx->ev0, st));
        rc = enqueue_pass(ctx, &ctx->last_p, ctx->last_mode, st, true);
        if (rc) return rc;
        return hf_finish(ctx, stats_host, stream);
    }
    return rc;
}

// Multi-GPU counterpart of hf_finish: the rows of ALL chunks are in `rows_dev` (all-gathered, row_index_dev maps list
// position -> row); reduce them in the fixed order straight into pinned host memory, wait, translate the flags.
// With an exchange geometry the flag rows of all ranks are OR-ed in, so every rank reports the same error and nobody is
// left waiting in the next collective.  One synchronisation per EM step.
static int finish_rows(hf_ctx* ctx, const double* rows_dev, const int32_t* row_index_dev, int64_t n_rows, ExchangeGeom xg,
                       double* stats_host, void* stream, const char* who) {
    if (!ctx || !rows_dev || !stats_host || n_rows < 0) return set_err(HF_E_ARG, std::string(who) + ": bad argument");
    hipStream_t st = (hipStream_t) stream;
    HIPCHK(hipSetDevice(ctx->device));
    double* out = ctx->d_total_host ? ctx->d_total_host : ctx->d_total;
    const bool polled = poll_ok(ctx, HF_K_REDUCE);
    const double seq = polled ? next_stamp(ctx) : 0.0;
    ctx->poll_kind = 2;
    int rc = reduce_chunks_seq(ctx, rows_dev, row_index_dev, n_rows, out, stream, seq, xg);
    if (rc) return rc;
    if (pass_events(ctx)) HIPCHK(hipEventRecord(ctx->ev1, st));
    if (!ctx->d_total_host)
        HIPCHK(hipMemcpyAsync(ctx->h_total, ctx->d_total, ((size_t) ctx->V + 1) * 8, hipMemcpyDeviceToHost, st));
    return wait_total(ctx, st, polled, stats_host);
}

int hf_finish_gathered(hf_ctx* ctx, const double* rows_dev, const int32_t* row_index_dev, int64_t n_chunks, double* stats_host,
                       void* stream) {
    return finish_rows(ctx, rows_dev, row_index_dev, n_chunks, ExchangeGeom(), stats_host, stream, "hf_finish_gathered");
}

int hf_finish_exchange(hf_ctx* ctx, const double* rows_dev, const int32_t* row_index_dev, int64_t n_rows, int n_ranks,
                       int rows_per_rank, int flag_row, double* stats_host, void* stream) {
    if (n_ranks < 1 || rows_per_rank < 2 || flag_row < 0 || flag_row >= rows_per_rank)
        return set_err(HF_E_ARG, "hf_finish_exchange: bad exchange geometry");
    ExchangeGeom xg;
    xg.n_ranks = n_ranks; xg.rows_per_rank = rows_per_rank; xg.flag_row = flag_row;
    return finish_rows(ctx, rows_dev, row_index_dev, n_rows, xg, stats_host, stream, "hf_finish_exchange");
}

int hf_bind_chunk_stats(hf_ctx* ctx, double* rows_dev) {
    if (!ctx) return set_err(HF_E_ARG, "hf_bind_chunk_stats: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (rows_dev) {
        if (ctx->own_chunk_stats) ctx_free(ctx, ctx->d_chunk_stats);
        ctx->d_chunk_stats = rows_dev; ctx->own_chunk_stats = false;
    } else if (!ctx->own_chunk_stats) {
        ctx->d_chunk_stats = nullptr;
        HIPCHK(hipMalloc((void**) &ctx->d_chunk_stats, ((size_t) ctx->C * (size_t) ctx->V + 1) * 8));
        ctx->own_chunk_stats = true;
    }
    return HF_OK;
}

int hf_bind_rank_total(hf_ctx* ctx, double* total_dev, double* flag_row_dev) {
    if (!ctx || ((total_dev == nullptr) != (flag_row_dev == nullptr))) return set_err(HF_E_ARG, "hf_bind_rank_total: bad argument");
    ctx->d_rank_out = total_dev; ctx->d_rank_flag = flag_row_dev;
    return HF_OK;
}

int hf_write_flag_row(hf_ctx* ctx, double* row_dev, void* stream) {
    if (!ctx || !row_dev) return set_err(HF_E_ARG, "hf_write_flag_row: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->pass_rows && !ctx->pass_nb && ctx->pass_bound && row_dev == ctx->d_rank_flag) return HF_OK;   // written by the pass itself
    hipLaunchKernelGGL(k_flag_row, dim3(1), dim3(64), 0, (hipStream_t) stream, ctx->d_flags, row_dev);
    HIPCHK(hipGetLastError());
    return HF_OK;
}

// One EM step in one call: E-step with the model's current parameters, reduced statistics back on the host,
// then (do_mstep) HMM_estimateParameters.  What runHMMFlagger repeats (hmm_flagger.c:337-445) without going back
// to the caller between the two halves.
int hf_em_iterate(hf_ctx* ctx, hfm_model* model, int mode, int do_mstep, double tol, double* stats_host, int* converged,
                  void* stream) {
    if (!ctx || !model || !stats_host || (mode != HF_MODE_FULL && mode != HF_MODE_FORWARD_ONLY))
        return set_err(HF_E_ARG, "hf_em_iterate: bad argument");
    using clk = std::chrono::steady_clock;
    const auto tp = clk::now();
    hf_params p;
    hfm_params(model, &p);
    int rc = HF_OK;
    const auto t0 = clk::now();
    auto t1 = t0, t2 = t0;
    rc = hf_estep(ctx, &p, mode, stream);
    t1 = clk::now();
    if (rc == HF_OK) rc = hf_finish(ctx, stats_host, stream);
    t2 = clk::now();
    if (rc != HF_OK) return rc;
    hfm_set_loglikelihood(model, stats_host[0]);
    if (do_mstep && mode == HF_MODE_FULL) {
        const int cv = hfm_estimate(model, stats_host, tol);
        if (converged) *converged = cv;
    }
    if (ctx->host_trace) {
        const auto t3 = clk::now();
        float gpu_ms = 0.f;
        (void) hipEventElapsedTime(&gpu_ms, ctx->ev0, ctx->ev1);
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        ctx->ht[0] += us(t0, t1); ctx->ht[1] += us(t1, t2); ctx->ht[2] += us(t2, t3); ctx->ht[3] += gpu_ms * 1e3; ctx->ht[4] += us(tp, t0); ctx->ht_n++;
    }
    return HF_OK;
}

int hf_get_labels(hf_ctx* ctx, int8_t* labels_host) {
    if (!ctx || !labels_host) return set_err(HF_E_ARG, "hf_get_labels: bad argument");
    if (!ctx->have_full) return set_err(HF_E_ARG, "hf_get_labels: the last pass was not HF_MODE_FULL (a forward-only pass decodes nothing)");
    HIPCHK(hipSetDevice(ctx->device));
    // through the context's pinned block (a pageable destination costs a staging copy inside the runtime: the first 1.5 MB download of a
    // process was measured at several milliseconds, in the middle of the command line's EM loop)
    // A KERNEL writes them into the context's pinned block (as the passes write their totals), the host copies them on: a process's first
    // device-to-host COPY into a fresh pinned block was measured at 8.9 ms (the second at 0.15 ms: profiles/r04e_cli_wall.txt) — in the
    // middle of the command line's EM loop, for the "initial" summary tables; the kernel takes ~0.1 ms the first time too.
    if (!ctx->h_label && ctx->N > 0 && ctx->d_total_host) {
        char* p = pin_cache().acquire((size_t) ctx->N + 16);
        void* dp = nullptr;
        if (p && hipHostGetDevicePointer(&dp, p, 0) == hipSuccess) { ctx->h_label = reinterpret_cast<int8_t*>(p); ctx->d_label_host = reinterpret_cast<int8_t*>(dp); }
        else { (void) hipGetLastError(); pin_cache().release(p); }
    }
    if (ctx->d_label_host && ctx->N > 0) {
        const int64_t n16 = (ctx->N + 15) / 16;      // (d_label and the pinned block are padded to 16 bytes)
        hipLaunchKernelGGL(k_copy16, dim3((unsigned) ((n16 + 255) / 256)), dim3(256), 0, ctx->last_stream, reinterpret_cast<const uint4*>(ctx->d_label),
                           reinterpret_cast<uint4*>(ctx->d_label_host), n16);   // (behind the pass, on its stream)
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->last_stream));
        std::memcpy(labels_host, ctx->h_label, (size_t) ctx->N);
    } else { HIPCHK(hipStreamSynchronize(ctx->last_stream)); HIPCHK(hipMemcpy(labels_host, ctx->d_label, (size_t) ctx->N, hipMemcpyDeviceToHost)); }
    return HF_OK;
}

int hf_get_forward_backward(hf_ctx* ctx, int64_t first, int64_t n, double* f_host, double* b_host, double* scales_host) {
    if (!ctx || first < 0 || n < 0 || first + n > ctx->N) return set_err(HF_E_ARG, "hf_get_forward_backward: bad range");
    if (!ctx->have_full)   // a forward-only pass of the segment kernels keeps everything in registers: nothing of THIS pass to return
        return set_err(HF_E_ARG, "hf_get_forward_backward: the last pass was not HF_MODE_FULL (forward, backward and scale values would be stale)");
    HIPCHK(hipSetDevice(ctx->device));
    if (n == 0) return HF_OK;
    HIPCHK(hipStreamSynchronize(ctx->last_stream));   // (the copies below run on the null stream: not ordered against a non-blocking stream by itself)
    if (scales_host && !ctx->fb_recs) HIPCHK(hipMemcpy(scales_host, ctx->d_scale + first, (size_t) n * 8, hipMemcpyDeviceToHost));
    if (!f_host && !b_host && !ctx->fb_recs) return HF_OK;
    if (ctx->fb_recs) {   // pair records (hf_seg.h): b_t is the second half of the record at pos[t], f_t the first half of the one at pos_f[t]
        if (!ctx->recs_all || !ctx->scales_all) {
            // the last full pass wrote no scales (statistics by emission row), or ran in sub-passes through the pass buffer (only its last sub-pass's records are left).  The segment kernel once
            // more over all segments, into the all-windows buffer (the tables of the pass are still in place: rows of A, parameters; labels,
            // scales and log-likelihood partials are rewritten with the same values)
            const int rc_ = all_records_buffer(ctx);
            if (rc_) return rc_;
            if (!ctx->d_scale_s) HIPCHK(hipMalloc((void**) &ctx->d_scale_s, (size_t) ctx->n_slots * 8));   // the scales: on first use (an EM pass writes none)
            // On the stream of the pass itself (the null stream is not ordered against a non-blocking user stream), and its flag
            // word is read like a pass's: a hand-off that timed out in THIS launch would otherwise leave silently wrong vectors — the context
            // falls back to two launches and the re-run is repeated; any other flag is the pass's own error, reported as hf_check would.
            hipStream_t st = ctx->last_stream;
            for (int attempt = 0;; attempt++) {
                if (!ctx->seg_fused) {
                    if (!ctx->d_segQ) HIPCHK(hipMalloc((void**) &ctx->d_segQ, (size_t) ctx->nseg * 64 * 16 * 8));
                    hipLaunchKernelGGL(k_seg_prod, dim3((unsigned) ctx->nseg), dim3(64), seg_lds_bytes(), st, ctx->d_seg, ctx->d_arow, ctx->d_lutA, ctx->d_segQ, ctx->d_Pseg);
                }
                const unsigned epoch = ++ctx->seg_epoch;
                launch_seg_fb(ctx, st, true, ctx->d_recs_all, -1, epoch, epoch, false);
                HIPCHK(hipGetLastError());
                HIPCHK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                const unsigned fl = *ctx->h_flags;
                if ((fl & HF_FLAG_SYNC) && ctx->seg_fused && attempt == 0) {
                    ctx->seg_fused = false;
                    std::fprintf(stderr, "[hmm_flagger_hip] one-launch segment kernel: a hand-off timed out; this context falls back to k_seg_prod + k_seg_fb\n");
                    unsigned keep = fl & ~(unsigned) HF_FLAG_SYNC;        // (the pass's own flags stay for a later hf_check)
                    HIPCHK(hipMemcpyAsync(ctx->d_flags, &keep, 4, hipMemcpyHostToDevice, st));
                    HIPCHK(hipStreamSynchronize(st));
                    continue;
                }
                const int code = flags_to_code(fl);
                if (code != HF_OK) return code;
                break;
            }
            ctx->recs_all = true; ctx->scales_all = true;
        }
        // the positions of a range are scattered over the plan: gathered on the device, one copy back (maps uploaded on first use)
        if (!ctx->d_slot_of) {
            HIPCHK(hipMalloc((void**) &ctx->d_slot_of, (size_t) ctx->N * 4));
            std::vector<int32_t> so((size_t) ctx->N, 0);     // window -> slot (the scales are kept in slot order)
            for (const SegDesc& d : ctx->h_segs)
                for (int64_t x = 0; x < d.n; x++) so[(size_t) (d.t0 + x)] = seg_slot(d, x);
            HIPCHK(hipMemcpy(ctx->d_slot_of, so.data(), (size_t) ctx->N * 4, hipMemcpyHostToDevice));
        }
        double* d_out = nullptr;
        HIPCHK(hipMalloc((void**) &d_out, (size_t) n * 9 * 8));
        hipLaunchKernelGGL(k_gather_fb, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->last_stream, first, n, ctx->d_pos, ctx->d_pos_f, ctx->d_slot_of,
                           ctx->d_recs_all, ctx->d_scale_s, d_out, d_out + n * 4, d_out + n * 8);
        HIPCHK(hipStreamSynchronize(ctx->last_stream));
        hipError_t e1 = hipSuccess, e2 = hipSuccess, e3 = hipSuccess;
        if (f_host) e1 = hipMemcpy(f_host, d_out, (size_t) n * 32, hipMemcpyDeviceToHost);
        if (b_host) e2 = hipMemcpy(b_host, d_out + n * 4, (size_t) n * 32, hipMemcpyDeviceToHost);
        if (scales_host) e3 = hipMemcpy(scales_host, d_out + n * 8, (size_t) n * 8, hipMemcpyDeviceToHost);
        hipFree(d_out);
        HIPCHK(e1); HIPCHK(e2); HIPCHK(e3);
        return HF_OK;
    }
    // f and b live tile-major / lane-minor on the device (hf_scan.h fb_slot): fetch the tiles that cover the range and
    // put every window's four values back in window order
    constexpr int64_t L = HF_SCAN_L, TW = 64 * L;
    auto locate = [&](int64_t t, int64_t* tile, int* lane, int* j) {
        size_t c = (size_t) (std::upper_bound(ctx->h_off.begin(), ctx->h_off.end(), t) - ctx->h_off.begin()) - 1;
        const int64_t w = t - ctx->h_off[c];
        *tile = ctx->h_tile0[c] + w / TW;
        const int rem = (int) (w % TW);
        *lane = rem / (int) L; *j = rem % (int) L;
    };
    int64_t tile_lo, tile_hi; int lane, j;
    locate(first, &tile_lo, &lane, &j);
    locate(first + n - 1, &tile_hi, &lane, &j);
    const size_t tile_doubles = (size_t) TW * 4;
    std::vector<double> buf((size_t) (tile_hi - tile_lo + 1) * tile_doubles);
    for (int which = 0; which < 2; which++) {
        double* dst = which == 0 ? f_host : b_host;
        if (!dst) continue;
        const double* src = which == 0 ? ctx->d_f : ctx->d_b;
        HIPCHK(hipMemcpy(buf.data(), src + (size_t) tile_lo * tile_doubles, buf.size() * 8, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; i++) {
            int64_t tile;
            locate(first + i, &tile, &lane, &j);
            for (int h = 0; h < 2; h++) {
                const size_t slot = (size_t) ((((tile - tile_lo) * L + j) * 2 + h) * 64 + lane);
                dst[i * 4 + 2 * h] = buf[slot * 2]; dst[i * 4 + 2 * h + 1] = buf[slot * 2 + 1];
            }
        }
    }
    return HF_OK;
}

int hf_get_posterior(hf_ctx* ctx, int64_t first, int64_t n, double* post_host) {
    if (!ctx || !post_host) return set_err(HF_E_ARG, "hf_get_posterior: bad argument");
    std::vector<double> f((size_t) n * 4), b((size_t) n * 4), sc((size_t) n);
    int rc = hf_get_forward_backward(ctx, first, n, f.data(), b.data(), sc.data());
    if (rc) return rc;
    for (int64_t i = 0; i < n; i++) { // hmm.c:671-685
        double total = 0.0;
        for (int s = 0; s < 4; s++) { post_host[i * 4 + s] = f[i * 4 + s] * b[i * 4 + s] * sc[i]; total += post_host[i * 4 + s]; }
        for (int s = 0; s < 4; s++) post_host[i * 4 + s] /= total;
    }
    return HF_OK;
}

int hf_set_profiling(hf_ctx* ctx, unsigned kernel_mask) {
    if (!ctx) return set_err(HF_E_ARG, "hf_set_profiling: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    // timing-only events: no system-scope fence when they fire (the default event writes back and invalidates the caches — k_seg_fb's 130 MB
    // of records would leave the L2 before k_pair_sums reads them: a sampled step was measured 15 us slower than an unsampled one)
    if (kernel_mask && !ctx->kev[0])
        for (int i = 0; i < 2 * HF_NKERNELS; i++)
            if (hipEventCreateWithFlags(&ctx->kev[i], hipEventDisableSystemFence) != hipSuccess) { (void) hipGetLastError(); HIPCHK(hipEventCreate(&ctx->kev[i])); }
    ctx->prof_mask = kernel_mask & (((1u << HF_NKERNELS) - 1u) | HF_PROF_PASS);
    for (int i = 0; i < HF_NKERNELS; i++) { ctx->kran[i] = false; ctx->ksum[i] = 0.0; ctx->kcount[i] = 0; }
    return HF_OK;
}

int hf_set_profiling_stride(hf_ctx* ctx, int every_nth_pass) {
    if (!ctx || every_nth_pass < 1) return set_err(HF_E_ARG, "hf_set_profiling_stride: bad argument");
    ctx->prof_stride = every_nth_pass; ctx->prof_pass = 0;
    return HF_OK;
}

int hf_kernel_time_sums(hf_ctx* ctx, double sum_ms[HF_NKERNELS], int64_t launches[HF_NKERNELS]) {
    if (!ctx || !sum_ms || !launches) return set_err(HF_E_ARG, "hf_kernel_time_sums: bad argument");
    for (int i = 0; i < HF_NKERNELS; i++) { sum_ms[i] = ctx->ksum[i]; launches[i] = ctx->kcount[i]; }
    return HF_OK;
}

int hf_kernel_times(hf_ctx* ctx, float ms[HF_NKERNELS]) {
    if (!ctx || !ms || !ctx->prof_mask) return set_err(HF_E_ARG, "hf_kernel_times: profiling is off");
    for (int i = 0; i < HF_NKERNELS; i++) {
        ms[i] = 0.f;
        if (ctx->kran[i]) {
            if (ctx->klast_ok[i]) ms[i] = ctx->klast[i];       // hf_finish has read this pair already
            else HIPCHK(hipEventElapsedTime(&ms[i], ctx->kev[2 * i], ctx->kev[2 * i + 1]));
        }
    }
    return HF_OK;
}

const char* hf_kernel_name(int k) {
    // slots 1-3 belonged to round 1's tile kernels (never launched since round 2); k_nb_total is the only kernel left in the
    // HF_K_ROWS_TOTAL slot (the Gaussian models' total is part of k_row_stats since round 3)
    static const char* names[HF_NKERNELS] = {"k_tables", "(retired)", "(retired)", "(retired)", "k_stats_tile", "k_chunk_stats",
                                             "k_reduce", "k_emit_rows", "k_fwd_seq", "k_bwd_seq", "k_pair_sums", "k_row_stats",
                                             "k_nb_total", "k_seg_prod", "k_seg_fb", "k_arows"};
    return k >= 0 && k < HF_NKERNELS ? names[k] : "?";
}

// self-test hook (tests/test_estep_gpu.py): quotient through prediv/divp next to the plain a / d, element-wise
__global__ void k_selftest_div(int64_t n, const double* __restrict__ a, const double* __restrict__ d,
                               double* __restrict__ fast, double* __restrict__ exact, int32_t* __restrict__ safe) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = divp(a[i], prediv(d[i]));
    exact[i] = a[i] / d[i];
    safe[i] = (div_operand_safe(a[i]) && d[i] != 0.0 && div_operand_safe(d[i])) ? 1 : 0;
}

int hf_selftest_division(int device, int64_t n, const double* a, const double* d, double* fast, double* exact, int32_t* safe) {
    if (n < 0 || !a || !d || !fast || !exact || !safe) return set_err(HF_E_ARG, "hf_selftest_division: bad argument");
    if (hf_device_count() <= 0) return set_err(HF_E_NOGPU, "hf_selftest_division: no HIP device");
    HIPCHK(hipSetDevice(device));
    double *da = nullptr, *dd = nullptr, *df = nullptr, *de = nullptr; int32_t* ds = nullptr;
    const size_t b = (size_t) (n ? n : 1) * 8;
    HIPCHK(hipMalloc((void**) &da, b)); HIPCHK(hipMalloc((void**) &dd, b)); HIPCHK(hipMalloc((void**) &df, b));
    HIPCHK(hipMalloc((void**) &de, b)); HIPCHK(hipMalloc((void**) &ds, b));
    HIPCHK(hipMemcpy(da, a, (size_t) n * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dd, d, (size_t) n * 8, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(k_selftest_div, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, n, da, dd, df, de, ds);
    HIPCHK(hipMemcpy(fast, df, (size_t) n * 8, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(exact, de, (size_t) n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(safe, ds, (size_t) n * 4, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(dd); hipFree(df); hipFree(de); hipFree(ds);
    return HF_OK;
}

// self-test hook (tests/test_estep_gpu.py): the emission densities' exp on the device (hf_exp.h) next to the same function and libm's exp on the host
__global__ void k_selftest_exp(int64_t n, const double* __restrict__ x, double* __restrict__ y) {
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = hf_exp(x[i]);      // (the restatement itself, whichever exp the emission kernels were built with)
}
int hf_selftest_exp(int device, int64_t n, const double* x, double* dev_out, double* host_out, double* libm_out) {
    if (n < 0 || !x || !dev_out || !host_out || !libm_out) return set_err(HF_E_ARG, "hf_selftest_exp: bad argument");
    if (hf_device_count() <= 0) return set_err(HF_E_NOGPU, "hf_selftest_exp: no HIP device");
    HIPCHK(hipSetDevice(device));
    double *dx = nullptr, *dy = nullptr;
    const size_t b = (size_t) (n ? n : 1) * 8;
    HIPCHK(hipMalloc((void**) &dx, b)); HIPCHK(hipMalloc((void**) &dy, b));
    HIPCHK(hipMemcpy(dx, x, (size_t) n * 8, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(k_selftest_exp, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, n, dx, dy);
    HIPCHK(hipMemcpy(dev_out, dy, (size_t) n * 8, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dy);
    for (int64_t i = 0; i < n; i++) { host_out[i] = hf_exp(x[i]); libm_out[i] = std::exp(x[i]); }
    return HF_OK;
}

int hf_last_kernel_ms(hf_ctx* ctx, float* ms) {
    if (!ctx || !ms || !ctx->ev_valid) return set_err(HF_E_ARG, "hf_last_kernel_ms: nothing timed yet");
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return HF_OK;
}

} // extern "C"
