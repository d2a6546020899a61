// hf_multi.hip — multi-GPU E-step of HMM-Flagger inside one process (include/hmm_flagger_multi.h): the chunk list is cut
// into contiguous shards, one per GPU; every GPU has its own host thread, HIP stream, E-step context (hf_estep.hip) and
// RCCL rank; a pass is E-step on the shard -> ONE all-gather over xGMI -> ordered reduction, identical on every rank.
// Replaces the pthread pool + in-order merge of EM_runOneIterationForList (programs/submodules/hmm/hmm.c:739-763).
// Host code only (the kernels are hf_estep.hip's); compiled with hipcc for the HIP and RCCL headers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "../../include/hmm_flagger_multi.h"
#include "../../include/hmm_flagger_model.h"
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <chrono>
#include <vector>
#include <immintrin.h>

static thread_local std::string g_merr;
static int merr(int code, const std::string& msg) { g_merr = msg; return code; }

// ------------------------------------------------------------------------------------------
// communicator
// ------------------------------------------------------------------------------------------
namespace {
// host barrier shared by the ranks of a loopback group
struct LoopGroup {
    int n = 0, device = 0;
    std::mutex m; std::condition_variable cv; int arrived = 0; unsigned long gen = 0;
    std::vector<const double*> src;
    std::vector<hipEvent_t> ready, done;
    std::atomic<int> refs{0};
    void barrier() {
        std::unique_lock<std::mutex> g(m);
        const unsigned long my = gen;
        if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(g, [&] { return gen != my; });
    }
};
}  // namespace

struct hf_comm {
    int rank = 0, size = 1, device = 0, transport = HF_TRANSPORT_RCCL;
    ncclComm_t nccl = nullptr;
    LoopGroup* loop = nullptr;
};

#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) \
    return merr(HF_E_HIP, std::string(#x) + ": " + ncclGetErrorString(r_)); } while (0)
#define HIPCHKM(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return merr(HF_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" {

const char* hf_comm_last_error(void) { return g_merr.c_str(); }
const char* hf_multi_last_error(void) { return g_merr.c_str(); }

int hf_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) <= HF_COMM_ID_BYTES, "ncclUniqueId does not fit HF_COMM_ID_BYTES");
    if (!id_out) return merr(HF_E_ARG, "hf_comm_unique_id: bad argument");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    std::memset(id_out, 0, HF_COMM_ID_BYTES);
    std::memcpy(id_out, &id, sizeof id);
    return HF_OK;
}

int hf_comm_init_rank(int n_ranks, int rank, int device, const void* id, hf_comm** out) {
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !id || !out) return merr(HF_E_ARG, "hf_comm_init_rank: bad argument");
    if (hf_device_count() <= device) return merr(HF_E_NOGPU, "hf_comm_init_rank: device index beyond the visible devices");
    HIPCHKM(hipSetDevice(device));
    ncclUniqueId nid;
    std::memcpy(&nid, id, sizeof nid);
    hf_comm* c = new hf_comm();
    c->rank = rank; c->size = n_ranks; c->device = device;
    ncclResult_t r = ncclCommInitRank(&c->nccl, n_ranks, nid, rank);
    if (r != ncclSuccess) { delete c; return merr(HF_E_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); }
    *out = c;
    return HF_OK;
}

int hf_comm_init_all(int n, const int* devices, hf_comm** out) {
    if (n < 1 || !devices || !out) return merr(HF_E_ARG, "hf_comm_init_all: bad argument");
    const int visible = hf_device_count();
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= visible)
            return merr(HF_E_NOGPU, "hf_comm_init_all: " + std::to_string(n) + " GPUs requested, device " + std::to_string(devices[i]) +
                                    " is not among the " + std::to_string(visible) + " visible");
        for (int j = 0; j < i; j++)
            if (devices[j] == devices[i]) return merr(HF_E_ARG, "hf_comm_init_all: one RCCL rank per GPU (a device is listed twice)");
    }
    std::vector<ncclComm_t> comms((size_t) n);
    NCCLCHK(ncclCommInitAll(comms.data(), n, devices));
    for (int i = 0; i < n; i++) {
        hf_comm* c = new hf_comm();
        c->rank = i; c->size = n; c->device = devices[i]; c->nccl = comms[(size_t) i];
        out[i] = c;
    }
    return HF_OK;
}

int hf_comm_init_loopback(int n, int device, hf_comm** out) {
    if (n < 1 || !out) return merr(HF_E_ARG, "hf_comm_init_loopback: bad argument");
    if (hf_device_count() <= device || device < 0) return merr(HF_E_NOGPU, "hf_comm_init_loopback: no such device");
    HIPCHKM(hipSetDevice(device));
    LoopGroup* g = new LoopGroup();
    g->n = n; g->device = device; g->src.assign((size_t) n, nullptr);
    g->ready.resize((size_t) n); g->done.resize((size_t) n);
    for (int i = 0; i < n; i++) {
        HIPCHKM(hipEventCreateWithFlags(&g->ready[(size_t) i], hipEventDisableTiming));
        HIPCHKM(hipEventCreateWithFlags(&g->done[(size_t) i], hipEventDisableTiming));
    }
    g->refs = n;
    for (int i = 0; i < n; i++) {
        hf_comm* c = new hf_comm();
        c->rank = i; c->size = n; c->device = device; c->transport = HF_TRANSPORT_LOOPBACK; c->loop = g;
        out[i] = c;
    }
    return HF_OK;
}

void hf_comm_destroy(hf_comm* c) {
    if (!c) return;
    if (c->nccl) { hipSetDevice(c->device); ncclCommDestroy(c->nccl); }
    if (c->loop && --c->loop->refs == 0) {
        hipSetDevice(c->loop->device);
        for (auto e : c->loop->ready) hipEventDestroy(e);
        for (auto e : c->loop->done) hipEventDestroy(e);
        delete c->loop;
    }
    delete c;
}

int hf_comm_rank(const hf_comm* c) { return c ? c->rank : -1; }
int hf_comm_size(const hf_comm* c) {   // what the communicator itself says (RCCL), not what it was asked for
    if (!c) return 0;
    if (c->nccl) { int n = 0; if (ncclCommCount(c->nccl, &n) == ncclSuccess) return n; return -1; }
    return c->size;
}

int hf_comm_allgather(hf_comm* c, const double* send_dev, double* recv_dev, int64_t count, void* stream) {
    if (!c || !send_dev || !recv_dev || count < 0) return merr(HF_E_ARG, "hf_comm_allgather: bad argument");
    hipStream_t st = (hipStream_t) stream;
    if (c->transport == HF_TRANSPORT_RCCL) {
        NCCLCHK(ncclAllGather(send_dev, recv_dev, (size_t) count, ncclDouble, c->nccl, st));
        return HF_OK;
    }
    // loopback: every rank lives on the same device; publish the send slot, wait for everybody's, copy, and keep the
    // slot alive until every peer has copied it
    LoopGroup* g = c->loop;
    HIPCHKM(hipSetDevice(c->device));
    g->src[(size_t) c->rank] = send_dev;
    HIPCHKM(hipEventRecord(g->ready[(size_t) c->rank], st));
    g->barrier();
    for (int k = 0; k < c->size; k++) {
        double* dst = recv_dev + (size_t) k * (size_t) count;
        if (k == c->rank && dst == send_dev) continue;          // in place
        HIPCHKM(hipStreamWaitEvent(st, g->ready[(size_t) k], 0));
        HIPCHKM(hipMemcpyAsync(dst, g->src[(size_t) k], (size_t) count * 8, hipMemcpyDeviceToDevice, st));
    }
    HIPCHKM(hipEventRecord(g->done[(size_t) c->rank], st));
    g->barrier();
    for (int k = 0; k < c->size; k++)
        if (k != c->rank) HIPCHKM(hipStreamWaitEvent(st, g->done[(size_t) k], 0));
    return HF_OK;
}

int hf_shard_bounds(const int64_t* chunk_off, int32_t n_chunks, int world, int32_t* bounds) {
    if (!chunk_off || n_chunks < 0 || world < 1 || !bounds) return merr(HF_E_ARG, "hf_shard_bounds: bad argument");
    // the same rule as flagger_amd/dist.py shard_bounds: the boundary closest to r/world of the windows, never backwards
    const int64_t base = chunk_off[0], total = chunk_off[n_chunks] - base;
    bounds[0] = 0;
    for (int r = 1; r < world; r++) {
        const double target = (double) total * r / world;
        int32_t lo = 0, hi = n_chunks + 1;                         // first k with csum[k] >= target
        while (lo < hi) { const int32_t mid = (lo + hi) / 2; if ((double) (chunk_off[mid] - base) < target) lo = mid + 1; else hi = mid; }
        int32_t k = lo;
        const int32_t kc = k < n_chunks ? k : n_chunks;
        if (k > 0) {
            const double below = (double) (chunk_off[k - 1] - base) - target, above = (double) (chunk_off[kc] - base) - target;
            if ((below < 0 ? -below : below) <= (above < 0 ? -above : above)) k -= 1;
        }
        if (k < bounds[r - 1]) k = bounds[r - 1];
        if (k > n_chunks) k = n_chunks;
        bounds[r] = k;
    }
    bounds[world] = n_chunks;
    return HF_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// hf_multi: one worker thread per device
// ------------------------------------------------------------------------------------------
namespace {
enum Cmd { CMD_NONE = 0, CMD_ESTEP, CMD_LABELS, CMD_POSTERIOR, CMD_EXIT };

struct RankState {
    int r = 0, device = 0;
    hf_ctx* ctx = nullptr;
    hf_comm* comm = nullptr;
    hipStream_t st = nullptr;
    double* xbuf = nullptr;            // [world][rows_per_rank][V]: the all-gather buffer; this rank's slot is written in place
    int32_t* d_row_index = nullptr;    // list position -> row of xbuf
    int64_t w0 = 0, nw = 0;            // window range of the shard
    int32_t c0 = 0, nc = 0;            // chunk range
    std::vector<int64_t> off;          // shard's chunk_off (starting at 0)
    std::vector<double> stats;
    int rc = HF_OK; std::string err;
    std::thread th;
    hipEvent_t xe0 = nullptr, xe1 = nullptr; double x_us = 0.0; long x_n = 0;   // HF_HOST_TRACE=1: the all-gather, bracketed by events
    double h_us[4] = {0, 0, 0, 0};   // HF_HOST_TRACE=2: host time in hf_estep / flag row + all-gather enqueue / hf_finish_exchange / the whole pass
};
}  // namespace

struct hf_multi {
    int world = 1, exchange = HF_EXCHANGE_CHUNKS, transport = HF_TRANSPORT_RCCL, algo = HF_ALGO_SCAN;
    int n_regions = 1, max_comps = 2;
    int64_t V = 0, N = 0; int32_t C = 0;
    int rows_per_rank = 2, flag_row = 1, maxc = 1;
    int local_rank = -1;               // >= 0: hf_multi_create_rank — this process holds rank `local_rank` only (ranks[0]), no workers
    std::vector<int32_t> bounds;
    std::vector<int64_t> shard_nw;     // windows of every rank's shard
    std::vector<RankState> ranks;
    const hf_windows* w = nullptr;     // valid during hf_multi_create only
    // command channel: main bumps `gen`, workers run `cmd` and decrement `pending`
    std::mutex m; std::condition_variable cv_cmd, cv_done;
    std::atomic<unsigned long> gen{0}; std::atomic<int> pending{0};
    int cmd = CMD_NONE;
    const hf_params* p = nullptr; int mode = HF_MODE_FULL;
    int8_t* labels_out = nullptr;
    int64_t post_first = 0, post_n = 0; double* post_out = nullptr;
};

namespace {

int rank_create(hf_multi* M, RankState& R) {
    const hf_windows* w = M->w;
    hf_windows sub = *w;
    sub.n_windows = R.nw; sub.n_chunks = R.nc;
    R.off.resize((size_t) R.nc + 1);
    for (int32_t c = 0; c <= R.nc; c++) R.off[(size_t) c] = w->chunk_off[R.c0 + c] - w->chunk_off[R.c0];
    sub.chunk_off = R.off.data();
    sub.cov = w->cov + R.w0; sub.mapq = w->mapq + R.w0; sub.clip = w->clip + R.w0; sub.annot = w->annot + R.w0;
    sub.chunk_s = w->chunk_s + R.c0; sub.chunk_e = w->chunk_e + R.c0; sub.chunk_ctg_len = w->chunk_ctg_len + R.c0;
    int rc = hf_create(&sub, M->n_regions, M->max_comps, R.device, M->algo, &R.ctx);
    if (rc != HF_OK) { R.err = hf_last_error(); return rc; }
    if (hipSetDevice(R.device) != hipSuccess || hipStreamCreateWithFlags(&R.st, hipStreamNonBlocking) != hipSuccess) {
        R.err = "hipStreamCreate failed"; return HF_E_HIP;
    }
    const size_t slot = (size_t) M->rows_per_rank * (size_t) M->V;
    if (hipMalloc((void**) &R.xbuf, (size_t) M->world * slot * 8) != hipSuccess) { R.err = "hipMalloc of the exchange buffer failed"; return HF_E_HIP; }
    hipMemset(R.xbuf, 0, (size_t) M->world * slot * 8);
    std::vector<int32_t> rows;
    if (M->exchange == HF_EXCHANGE_CHUNKS) {
        for (int k = 0; k < M->world; k++)
            for (int32_t c = M->bounds[(size_t) k]; c < M->bounds[(size_t) k + 1]; c++)
                rows.push_back(k * M->rows_per_rank + (c - M->bounds[(size_t) k]));
        rc = hf_set_stats_mode(R.ctx, HF_STATS_CHUNKS);
        if (rc == HF_OK) rc = hf_bind_chunk_stats(R.ctx, R.xbuf + (size_t) R.r * slot);
    } else {
        for (int k = 0; k < M->world; k++) rows.push_back(k * M->rows_per_rank);
        rc = hf_set_stats_mode(R.ctx, HF_STATS_ROWS);
        // a full pass then writes its total and its flag word straight into this rank's slot: no copy kernels before the all-gather
        if (rc == HF_OK) rc = hf_bind_rank_total(R.ctx, R.xbuf + (size_t) R.r * slot, R.xbuf + (size_t) R.r * slot + (size_t) M->flag_row * (size_t) M->V);
    }
    if (rc != HF_OK) { R.err = hf_last_error(); return rc; }
    if (rows.empty()) rows.push_back(0);
    if (hipMalloc((void**) &R.d_row_index, rows.size() * 4) != hipSuccess ||
        hipMemcpy(R.d_row_index, rows.data(), rows.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        R.err = "row index upload failed"; return HF_E_HIP;
    }
    R.stats.assign((size_t) M->V, 0.0);
    {
        const char* e = std::getenv("HF_HOST_TRACE");
        if (e && e[0] == '1') {
            hipEventCreate(&R.xe0); hipEventCreate(&R.xe1);
            std::fprintf(stderr, "[hf_multi] rank %d of %d on GPU %d: chunks %d..%d (%d), windows %lld..%lld (%lld), exchange %s, %d RCCL ranks\n", R.r, M->world,
                         R.device, R.c0, R.c0 + R.nc - 1, R.nc, (long long) R.w0, (long long) (R.w0 + R.nw - 1), (long long) R.nw,
                         M->exchange == HF_EXCHANGE_CHUNKS ? "chunks" : "ranks", hf_comm_size(R.comm));
        }
    }
    return HF_OK;
}

// everything a rank owns (hf_multi_destroy and every failure path of the two constructors)
void rank_release(RankState& R) {
    if (R.th.joinable()) R.th.join();
    hipSetDevice(R.device);
    if (R.xe0) {
        if (R.x_n) std::fprintf(stderr, "[hf_multi] rank %d: %ld passes, all-gather %.1f us per pass (events on the pass's stream)\n", R.r, R.x_n, R.x_us / R.x_n);
        if (R.h_us[3] > 0.0) std::fprintf(stderr, "[hf_multi] rank %d host time (sums, ms): hf_estep %.2f, collective enqueue %.2f, hf_finish_exchange %.2f, pass %.2f\n", R.r, R.h_us[0] / 1e3, R.h_us[1] / 1e3, R.h_us[2] / 1e3, R.h_us[3] / 1e3);
        hipEventDestroy(R.xe0); hipEventDestroy(R.xe1); R.xe0 = R.xe1 = nullptr;
    }
    if (R.ctx) { hf_bind_chunk_stats(R.ctx, nullptr); hf_destroy(R.ctx); R.ctx = nullptr; }
    if (R.xbuf) { hipFree(R.xbuf); R.xbuf = nullptr; }
    if (R.d_row_index) { hipFree(R.d_row_index); R.d_row_index = nullptr; }
    if (R.st) { hipStreamDestroy(R.st); R.st = nullptr; }
    if (R.comm) { hf_comm_destroy(R.comm); R.comm = nullptr; }
}

int rank_estep_once(hf_multi* M, RankState& R);
int rank_estep(hf_multi* M, RankState& R) {
    int rc = rank_estep_once(M, R);
    if (rc == HF_E_RETRY) rc = rank_estep_once(M, R);   // every rank got it together (the flag words are OR-ed): all run the pass again
    return rc;
}
int rank_estep_once(hf_multi* M, RankState& R) {
    const size_t slot = (size_t) M->rows_per_rank * (size_t) M->V;
    double* mine = R.xbuf + (size_t) R.r * slot;
    static const bool htrace = [] { const char* e = std::getenv("HF_HOST_TRACE"); return e && e[0] == '2'; }();
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto lap = [&](int k, clk::time_point a) { if (htrace) R.h_us[k] += std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
    int rc = hf_estep(R.ctx, M->p, M->mode, R.st);
    lap(0, t0);
    const auto t1 = clk::now();
    // a rank whose launch failed must still take part in the collective, or the others hang: it sends what it has
    // and reports its own error afterwards
    std::string first_err;
    if (rc != HF_OK) first_err = hf_last_error();
    if (rc == HF_OK && M->exchange == HF_EXCHANGE_RANKS) {
        rc = hf_rank_total(R.ctx, mine, R.st);
        if (rc != HF_OK) first_err = hf_last_error();
    }
    int rc2 = hf_write_flag_row(R.ctx, mine + (size_t) M->flag_row * (size_t) M->V, R.st);
    if (rc2 != HF_OK && first_err.empty()) first_err = hf_last_error();
    if (R.xe0) hipEventRecord(R.xe0, R.st);
    const int rc3 = hf_comm_allgather(R.comm, mine, R.xbuf, (int64_t) slot, R.st);   // whatever happened above: the peers are waiting in it
    if (R.xe0) hipEventRecord(R.xe1, R.st);
    lap(1, t1);
    const auto t2 = clk::now();
    if (rc3 != HF_OK && first_err.empty()) first_err = hf_comm_last_error();
    if (rc == HF_OK) rc = rc2 != HF_OK ? rc2 : rc3;
    if (rc != HF_OK) { R.err = first_err; hipStreamSynchronize(R.st); return rc; }
    const int64_t n_rows = M->exchange == HF_EXCHANGE_CHUNKS ? (int64_t) M->C : (int64_t) M->world;
    rc = hf_finish_exchange(R.ctx, R.xbuf, R.d_row_index, n_rows, M->world, M->rows_per_rank, M->flag_row, R.stats.data(), R.st);
    if (rc != HF_OK) R.err = hf_last_error();
    lap(2, t2); lap(3, t0);
    if (R.xe0 && rc == HF_OK) { float ms = 0.f; if (hipEventElapsedTime(&ms, R.xe0, R.xe1) == hipSuccess) { R.x_us += ms * 1e3; R.x_n++; } }
    return rc;
}

void worker(hf_multi* M, int r) {
    RankState& R = M->ranks[(size_t) r];
    hipSetDevice(R.device);
    unsigned long seen = 0;
    for (;;) {
        // wait for the next command: spin briefly (an EM step is ~100 us), then sleep
        int spins = 0;
        while (M->gen.load(std::memory_order_acquire) == seen) {
            if (++spins < 20000) _mm_pause();
            else {
                std::unique_lock<std::mutex> g(M->m);
                M->cv_cmd.wait(g, [&] { return M->gen.load(std::memory_order_acquire) != seen; });
            }
        }
        seen = M->gen.load(std::memory_order_acquire);
        const int cmd = M->cmd;
        R.rc = HF_OK; R.err.clear();
        if (cmd == CMD_ESTEP) R.rc = rank_estep(M, R);
        else if (cmd == CMD_LABELS) {
            if (R.nw > 0) { R.rc = hf_get_labels(R.ctx, M->labels_out + R.w0); if (R.rc) R.err = hf_last_error(); }
        } else if (cmd == CMD_POSTERIOR) {
            const int64_t a = M->post_first > R.w0 ? M->post_first : R.w0;
            const int64_t b = M->post_first + M->post_n < R.w0 + R.nw ? M->post_first + M->post_n : R.w0 + R.nw;
            if (a < b) {
                R.rc = hf_get_posterior(R.ctx, a - R.w0, b - a, M->post_out + (a - M->post_first) * 4);
                if (R.rc) R.err = hf_last_error();
            }
        }
        if (M->pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> g(M->m);
            M->cv_done.notify_all();
        }
        if (cmd == CMD_EXIT) return;
    }
}

// run `cmd` on every rank; returns the first non-zero code in rank order (its text in g_merr)
int run_all(hf_multi* M, int cmd) {
    {
        std::lock_guard<std::mutex> g(M->m);
        M->cmd = cmd;
        M->pending.store(M->world, std::memory_order_release);
        M->gen.fetch_add(1, std::memory_order_acq_rel);
    }
    M->cv_cmd.notify_all();
    int spins = 0;
    while (M->pending.load(std::memory_order_acquire) != 0) {
        if (++spins < 200000) _mm_pause();
        else {
            std::unique_lock<std::mutex> g(M->m);
            M->cv_done.wait(g, [&] { return M->pending.load(std::memory_order_acquire) == 0; });
        }
    }
    for (auto& R : M->ranks)
        if (R.rc != HF_OK) return merr(R.rc, "rank " + std::to_string(R.r) + " (GPU " + std::to_string(R.device) + "): " + R.err);
    return HF_OK;
}

}  // namespace

extern "C" {

int hf_multi_create(const hf_windows* w, int n_regions, int max_comps, int n_devices, const int* devices, int algo, int exchange,
                    int transport, hf_multi** out) {
    if (!w || !out || n_devices < 1 || n_devices > 64 || (exchange != HF_EXCHANGE_CHUNKS && exchange != HF_EXCHANGE_RANKS) ||
        (transport != HF_TRANSPORT_RCCL && transport != HF_TRANSPORT_LOOPBACK))
        return merr(HF_E_ARG, "hf_multi_create: bad argument");
    const int visible = hf_device_count();
    if (visible <= 0) return merr(HF_E_NOGPU, "hf_multi_create: no HIP device (there is no CPU fallback)");
    std::vector<int> dev((size_t) n_devices);
    for (int i = 0; i < n_devices; i++) dev[(size_t) i] = devices ? devices[i] : (transport == HF_TRANSPORT_LOOPBACK ? 0 : i);
    if (transport == HF_TRANSPORT_RCCL && n_devices > visible)
        return merr(HF_E_NOGPU, std::to_string(n_devices) + " GPUs requested but only " + std::to_string(visible) +
                                " visible: one RCCL rank per GPU, ranks cannot share a device");
    hf_multi* M = new hf_multi();
    M->world = n_devices; M->exchange = exchange; M->transport = transport; M->algo = algo;
    M->n_regions = n_regions; M->max_comps = max_comps;
    M->V = hf_stats_len(n_regions, max_comps); M->N = w->n_windows; M->C = w->n_chunks;
    M->bounds.assign((size_t) n_devices + 1, 0);
    int rc = hf_shard_bounds(w->chunk_off, w->n_chunks, n_devices, M->bounds.data());
    if (rc != HF_OK) { delete M; return rc; }
    M->maxc = 1;
    for (int r = 0; r < n_devices; r++) { const int n = M->bounds[(size_t) r + 1] - M->bounds[(size_t) r]; if (n > M->maxc) M->maxc = n; }
    for (int r = 0; r < n_devices; r++) M->shard_nw.push_back(w->chunk_off[M->bounds[(size_t) r + 1]] - w->chunk_off[M->bounds[(size_t) r]]);
    if (exchange == HF_EXCHANGE_CHUNKS) { M->rows_per_rank = M->maxc + 1; M->flag_row = M->maxc; }
    else { M->rows_per_rank = 2; M->flag_row = 1; }
    std::vector<hf_comm*> comms((size_t) n_devices, nullptr);
    rc = transport == HF_TRANSPORT_RCCL ? hf_comm_init_all(n_devices, dev.data(), comms.data())
                                        : hf_comm_init_loopback(n_devices, dev[0], comms.data());
    if (rc != HF_OK) { delete M; return rc; }
    M->ranks.resize((size_t) n_devices);
    M->w = w;
    for (int r = 0; r < n_devices; r++) {
        RankState& R = M->ranks[(size_t) r];
        R.r = r; R.device = transport == HF_TRANSPORT_LOOPBACK ? dev[0] : dev[(size_t) r]; R.comm = comms[(size_t) r];
        R.c0 = M->bounds[(size_t) r]; R.nc = M->bounds[(size_t) r + 1] - R.c0;
        R.w0 = w->chunk_off[R.c0] - w->chunk_off[0]; R.nw = w->chunk_off[R.c0 + R.nc] - w->chunk_off[R.c0];
    }
    {   // upload the shards in parallel, one thread per device
        std::vector<std::thread> ths;
        for (int r = 0; r < n_devices; r++) ths.emplace_back([M, r] { RankState& R = M->ranks[(size_t) r]; R.rc = rank_create(M, R); });
        for (auto& t : ths) t.join();
    }
    M->w = nullptr;
    for (auto& R : M->ranks)
        if (R.rc != HF_OK) {
            const int code = R.rc;
            merr(code, "rank " + std::to_string(R.r) + " (GPU " + std::to_string(R.device) + "): " + R.err);
            const std::string keep = g_merr;
            for (auto& Q : M->ranks) rank_release(Q);
            delete M;
            g_merr = keep;
            return code;
        }
    for (int r = 0; r < n_devices; r++) M->ranks[(size_t) r].th = std::thread(worker, M, r);
    *out = M;
    return HF_OK;
}

// One process per GPU (the layout torch.distributed.run / mpirun start): this process is rank `rank` of `world`, owns
// `device` and the rank's shard of `w` (every process passes the same whole window set, or at least the same chunk list and
// its own shard's windows at their global positions); `unique_id` = hf_comm_unique_id() of rank 0, carried to the others by
// the launcher.  Collective: every rank must call it.  hf_multi_estep then runs on the caller's thread.
int hf_multi_create_rank(const hf_windows* w, int n_regions, int max_comps, int world, int rank, int device, int algo, int exchange,
                         const void* unique_id, hf_multi** out) {
    if (!w || !out || !unique_id || world < 1 || world > 4096 || rank < 0 || rank >= world ||
        (exchange != HF_EXCHANGE_CHUNKS && exchange != HF_EXCHANGE_RANKS))
        return merr(HF_E_ARG, "hf_multi_create_rank: bad argument");
    // A rank that returns before ncclCommInitRank leaves its peers blocked inside it: only the checks every rank evaluates
    // identically (the arguments above) and the one without which no communicator can exist (a device) come first — the
    // LAUNCHER must agree on those (bench.py / hmm.RankEMList all-reduce a go / no-go before this call).  Everything else is
    // checked after the communicator exists, and the ranks then agree on the outcome through it (below).
    const int visible = hf_device_count();
    if (visible <= 0) return merr(HF_E_NOGPU, "hf_multi_create_rank: no HIP device (there is no CPU fallback)");
    if (device < 0 || device >= visible) return merr(HF_E_NOGPU, "hf_multi_create_rank: device " + std::to_string(device) + " is not visible");
    hf_multi* M = new hf_multi();
    M->world = world; M->exchange = exchange; M->transport = HF_TRANSPORT_RCCL; M->algo = algo; M->local_rank = rank;
    M->n_regions = n_regions; M->max_comps = max_comps;
    M->V = hf_stats_len(n_regions, max_comps); M->N = w->n_windows; M->C = w->n_chunks;
    M->ranks.resize(1);
    RankState& R = M->ranks[0];
    R.r = rank; R.device = device;
    int rc = hf_comm_init_rank(world, rank, device, unique_id, &R.comm);      // collective: first
    if (rc != HF_OK) { const std::string keep = g_merr; rank_release(R); delete M; g_merr = keep; return rc; }
    M->bounds.assign((size_t) world + 1, 0);
    rc = hf_shard_bounds(w->chunk_off, w->n_chunks, world, M->bounds.data());
    if (rc == HF_OK) {
        M->maxc = 1;
        for (int r = 0; r < world; r++) { const int n = M->bounds[(size_t) r + 1] - M->bounds[(size_t) r]; if (n > M->maxc) M->maxc = n; }
        for (int r = 0; r < world; r++) M->shard_nw.push_back(w->chunk_off[M->bounds[(size_t) r + 1]] - w->chunk_off[M->bounds[(size_t) r]]);
        if (exchange == HF_EXCHANGE_CHUNKS) { M->rows_per_rank = M->maxc + 1; M->flag_row = M->maxc; }
        else { M->rows_per_rank = 2; M->flag_row = 1; }
        R.c0 = M->bounds[(size_t) rank]; R.nc = M->bounds[(size_t) rank + 1] - R.c0;
        R.w0 = w->chunk_off[R.c0] - w->chunk_off[0]; R.nw = w->chunk_off[R.c0 + R.nc] - w->chunk_off[R.c0];
        M->w = w;
        rc = rank_create(M, R);
        M->w = nullptr;
        if (rc != HF_OK) merr(rc, "rank " + std::to_string(rank) + " (GPU " + std::to_string(device) + "): " + R.err);
    }
    // every rank learns whether every rank's set-up worked: one status word each, all-gathered over the new communicator
    // (a rank that failed above still takes part, so nobody waits for it in the first EM pass)
    {
        double* d_status = nullptr;
        std::vector<double> h_status((size_t) world, 0.0);
        bool agreed = hipSetDevice(device) == hipSuccess && hipMalloc((void**) &d_status, (size_t) world * 8) == hipSuccess;
        if (agreed) {
            const double mine = (double) rc;
            agreed = hipMemcpy(d_status + rank, &mine, 8, hipMemcpyHostToDevice) == hipSuccess &&
                     hf_comm_allgather(R.comm, d_status + rank, d_status, 1, nullptr) == HF_OK &&
                     hipDeviceSynchronize() == hipSuccess &&
                     hipMemcpy(h_status.data(), d_status, (size_t) world * 8, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (d_status) hipFree(d_status);
        if (!agreed && rc == HF_OK) rc = merr(HF_E_HIP, "hf_multi_create_rank: the ranks could not exchange their set-up status");
        if (rc == HF_OK)
            for (int r = 0; r < world; r++)
                if (h_status[(size_t) r] != 0.0) { rc = merr((int) h_status[(size_t) r], "hf_multi_create_rank: the set-up of rank " + std::to_string(r) + " failed"); break; }
    }
    if (rc != HF_OK) {
        const std::string keep = g_merr;
        rank_release(R);
        delete M;
        g_merr = keep;
        return rc;
    }
    *out = M;
    return HF_OK;
}

void hf_multi_destroy(hf_multi* M) {
    if (!M) return;
    if (M->local_rank < 0) (void) run_all(M, CMD_EXIT);
    for (auto& R : M->ranks) rank_release(R);
    delete M;
}

int hf_multi_estep(hf_multi* M, const hf_params* p, int mode, double* stats_host) {
    if (!M || !p || !stats_host || (mode != HF_MODE_FULL && mode != HF_MODE_FORWARD_ONLY)) return merr(HF_E_ARG, "hf_multi_estep: bad argument");
    M->p = p; M->mode = mode;
    if (M->local_rank >= 0) {           // one process per GPU: this thread is the rank
        RankState& R = M->ranks[0];
        hipSetDevice(R.device);
        const int rc1 = rank_estep(M, R);
        if (rc1 != HF_OK) return merr(rc1, "rank " + std::to_string(R.r) + " (GPU " + std::to_string(R.device) + "): " + R.err);
        std::memcpy(stats_host, R.stats.data(), (size_t) M->V * 8);
        return HF_OK;
    }
    const int rc = run_all(M, CMD_ESTEP);
    if (rc != HF_OK) return rc;
    std::memcpy(stats_host, M->ranks[0].stats.data(), (size_t) M->V * 8);
    return HF_OK;
}

// One EM step in one call, the multi-GPU counterpart of hf_em_iterate: sharded E-step + exchange, then the (replicated) M-step.
int hf_multi_em_iterate(hf_multi* M, hfm_model* model, int mode, int do_mstep, double tol, double* stats_host, int* converged) {
    if (!M || !model || !stats_host) return merr(HF_E_ARG, "hf_multi_em_iterate: bad argument");
    hf_params p;
    hfm_params(model, &p);
    const int rc = hf_multi_estep(M, &p, mode, stats_host);
    if (rc != HF_OK) return rc;
    hfm_set_loglikelihood(model, stats_host[0]);
    if (do_mstep && mode == HF_MODE_FULL) {
        const int cv = hfm_estimate(model, stats_host, tol);
        if (converged) *converged = cv;
    }
    return HF_OK;
}

int hf_multi_get_labels(hf_multi* M, int8_t* labels_host) {
    if (!M || !labels_host) return merr(HF_E_ARG, "hf_multi_get_labels: bad argument");
    M->labels_out = labels_host;
    if (M->local_rank >= 0) {           // this rank's windows only, at their global positions
        RankState& R = M->ranks[0];
        if (R.nw <= 0) return HF_OK;
        const int rc = hf_get_labels(R.ctx, labels_host + R.w0);
        return rc == HF_OK ? HF_OK : merr(rc, hf_last_error());
    }
    return run_all(M, CMD_LABELS);
}

int hf_multi_get_posterior(hf_multi* M, int64_t first, int64_t n, double* post_host) {
    if (!M || !post_host || first < 0 || n < 0 || first + n > M->N) return merr(HF_E_ARG, "hf_multi_get_posterior: bad range");
    M->post_first = first; M->post_n = n; M->post_out = post_host;
    if (M->local_rank >= 0) {
        RankState& R = M->ranks[0];
        const int64_t a = first > R.w0 ? first : R.w0, b = first + n < R.w0 + R.nw ? first + n : R.w0 + R.nw;
        if (a >= b) return HF_OK;
        const int rc = hf_get_posterior(R.ctx, a - R.w0, b - a, post_host + (a - first) * 4);
        return rc == HF_OK ? HF_OK : merr(rc, hf_last_error());
    }
    return run_all(M, CMD_POSTERIOR);
}

int hf_multi_world(const hf_multi* M) { return M ? M->world : 0; }
int hf_multi_comm_ranks(const hf_multi* M) { return (M && !M->ranks.empty()) ? hf_comm_size(M->ranks[0].comm) : 0; }
int64_t hf_multi_stats_len(const hf_multi* M) { return M ? M->V : 0; }
static const RankState* rank_of(const hf_multi* M, int r) {
    if (!M || r < 0 || r >= M->world) return nullptr;
    if (M->local_rank >= 0) return r == M->local_rank ? &M->ranks[0] : nullptr;
    return &M->ranks[(size_t) r];
}
int64_t hf_multi_shard_windows(const hf_multi* M, int r) { return (M && r >= 0 && r < M->world) ? M->shard_nw[(size_t) r] : 0; }
int32_t hf_multi_shard_chunks(const hf_multi* M, int r) { return (M && r >= 0 && r < M->world) ? M->bounds[(size_t) r + 1] - M->bounds[(size_t) r] : 0; }
hf_ctx* hf_multi_local_ctx(hf_multi* M) { return (M && M->local_rank >= 0) ? M->ranks[0].ctx : nullptr; }
int64_t hf_multi_local_first_window(const hf_multi* M) { return (M && M->local_rank >= 0) ? M->ranks[0].w0 : 0; }
int64_t hf_multi_local_windows(const hf_multi* M) { return (M && M->local_rank >= 0) ? M->ranks[0].nw : (M ? M->N : 0); }

int hf_multi_rank_stats(hf_multi* M, int r, double* stats_host) {
    const RankState* R = rank_of(M, r);
    if (!R || !stats_host) return merr(HF_E_ARG, "hf_multi_rank_stats: bad argument");
    std::memcpy(stats_host, R->stats.data(), (size_t) M->V * 8);
    return HF_OK;
}

}  // extern "C"
