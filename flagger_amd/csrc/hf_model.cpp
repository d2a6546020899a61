// hf_model.cpp — host-side model of the MI355X build: initial parameters, the parameter view the
// E-step consumes and the M-step (include/hmm_flagger_model.h).  The estimators of the reference
// (ParameterEstimator / TransitionCountData, hmm_utils.h:93-107, 826-834) are not objects here:
// the M-step reads the flat statistics vector the device produced.
#include "../../include/hmm_flagger_model.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <algorithm>
#include <chrono>

namespace {
constexpr int S = HF_NSTATES;
constexpr int KM = HF_MAXCOMP;
constexpr double kMinCountForUpdate = 10;   // hmm_utils.h:11
constexpr double kTruncFraction = 0.25;     // hmm_utils.h:12
constexpr double kErrBinding = 0.1;         // hmm_utils.h:14
constexpr double kTermination = 1e-4;       // hmm_utils.c:2112
constexpr double kDiagProb = 0.99;          // hmm.c:15
constexpr double kPseudoCount = 0.001;      // hmm.c:16
constexpr int kMaxCov = HF_NB_MAX_COVERAGE; // hmm_utils.h:15 MAX_COVERAGE_VALUE

// digamma/digamma.c:27-117 — digamma in long double: reflection below 0, recurrence below 1, exact values at 1, 2, 3,
// duplication formula above 3, and on (1,3) the Chebyshev expansion of R. J. Mathar, arXiv:math.CA/0403344 app. E
// (J. Wimp, Math. Comp. 15 (1961) 174, table 1); the coefficients are the published table.
long double digammal_(long double x) {
    static const long double euler = 0.5772156649015328606065120900824024L;
    static const long double ln2 = 0.6931471805599453094172321214581766L;
    static const long double pi = 3.1415926535897932384626433832795029L;
    static const long double K[] = {
        .30459198558715155634315638246624251L, .72037977439182833573548891941219706L,
        -.12454959243861367729528855995001087L, .27769457331927827002810119567456810e-1L,
        -.67762371439822456447373550186163070e-2L, .17238755142247705209823876688592170e-2L,
        -.44817699064252933515310345718960928e-3L, .11793660000155572716272710617753373e-3L,
        -.31253894280980134452125172274246963e-4L, .83173997012173283398932708991137488e-5L,
        -.22191427643780045431149221890172210e-5L, .59302266729329346291029599913617915e-6L,
        -.15863051191470655433559920279603632e-6L, .42459203983193603241777510648681429e-7L,
        -.11369129616951114238848106591780146e-7L, .304502217295931698401459168423403510e-8L,
        -.81568455080753152802915013641723686e-9L, .21852324749975455125936715817306383e-9L,
        -.58546491441689515680751900276454407e-10L, .15686348450871204869813586459513648e-10L,
        -.42029496273143231373796179302482033e-11L, .11261435719264907097227520956710754e-11L,
        -.30174353636860279765375177200637590e-12L, .80850955256389526647406571868193768e-13L,
        -.21663779809421233144009565199997351e-13L, .58047634271339391495076374966835526e-14L,
        -.15553767189204733561108869588173845e-14L, .41676108598040807753707828039353330e-15L,
        -.11167065064221317094734023242188463e-15L };
    if (x < 0.0L) return digammal_(1.0L - x) + pi / tanl(pi * (1.0L - x));
    if (x < 1.0L) return digammal_(1.0L + x) - 1.0L / x;
    if (x == 1.0L) return -euler;
    if (x == 2.0L) return 1.0L - euler;
    if (x == 3.0L) return 1.5L - euler;
    if (x > 3.0L) return 0.5L * (digammal_(x / 2.0L) + digammal_((x + 1.0L) / 2.0L)) + ln2;
    long double t0 = 1.0L, t1 = x - 2.0L;
    long double res = K[0] + K[1] * t1;
    x -= 2.0L;
    for (size_t n = 2; n < sizeof(K) / sizeof(K[0]); n++) {
        const long double t2 = 2.0L * x * t1 - t0;
        res += K[n] * t2;
        t0 = t1;
        t1 = t2;
    }
    return res;
}
}

struct hfm_model {
    int model_type = 0, R = 1, K = 2;
    int ncomp[S] = {1, 1, 1, 2};
    double alpha[4][4] = {};
    double max_mapq = 0.25, min_mapq = 0.75, min_clip = 1.0;
    double loglik = 0.0;
    std::vector<double> trans;     // [R][5][5]
    std::vector<double> lambda, trunc;  // [R]
    std::vector<double> mean, var, weight; // [R][4][KM]; negative_binomial model: mean = theta, var = lambda (hmm_utils.h NegativeBinomial)
    // negative_binomial: per-iteration tables handed to the E-step (rebuilt by hfm_params)
    mutable std::vector<double> nb_E, nb_P, nb_dig, nb_r, nb_beta;
    int nb_max_x = kMaxCov;          // tables filled for x = 0..nb_max_x (hfm_set_max_coverage)
    bool nb() const { return model_type == HF_MODEL_NEGATIVE_BINOMIAL; }
    double& M(int r, int s, int c) { return mean[((size_t) r * S + s) * KM + c]; }
    double& Vr(int r, int s, int c) { return var[((size_t) r * S + s) * KM + c]; }
    double& W(int r, int s, int c) { return weight[((size_t) r * S + s) * KM + c]; }
    double& T(int r, int i, int j) { return trans[(size_t) r * 25 + i * 5 + j]; }
    bool gaussian_state(int s) const { return !(s == 0 && model_type == HF_MODEL_TRUNC_EXP_GAUSSIAN); }
};

// a few host threads for the negative-binomial tables of hfm_params (started on first use; the caller works too)
namespace {
struct NbPool {
    std::mutex m, run_m;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    const std::function<void(size_t)>* job = nullptr;
    size_t n = 0, gen = 0;
    std::atomic<size_t> next{0}, done{0}, gen_hint{0};
    int busy = 0;
    bool stop = false;
    void worker() {
        size_t seen = 0;
        for (;;) {
            const std::function<void(size_t)>* f; size_t N;
            {
                // (an EM loop asks again within ~0.2 ms: poll for that long before sleeping — a condition variable's wake-up costs 10-30 us)
                const auto t_spin = std::chrono::steady_clock::now();
                while (gen_hint.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() - t_spin < std::chrono::microseconds(400)) __builtin_ia32_pause();
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen; f = job; N = n;
                if (!f) continue;
                busy++;
            }
            for (size_t k; (k = next.fetch_add(1)) < N;) { (*f)(k); done.fetch_add(1); }
            { std::lock_guard<std::mutex> g(m); busy--; }
            cv_done.notify_all();
        }
    }
    void run(size_t N, const std::function<void(size_t)>& f) {
        std::unique_lock<std::mutex> only(run_m, std::try_to_lock);      // (two models filling their tables at once: the second one by itself)
        if (N < 4 || !only.owns_lock()) { for (size_t k = 0; k < N; k++) f(k); return; }
        {
            std::lock_guard<std::mutex> g(m);
            if (th.empty()) {
                const unsigned hw = std::thread::hardware_concurrency();
                const size_t T = std::min<size_t>(7, hw > 1 ? hw - 1 : 0);
                for (size_t t = 0; t < T; t++) th.emplace_back([this] { worker(); });
            }
            next.store(0); done.store(0); job = &f; n = N; gen++; gen_hint.store(gen, std::memory_order_release);
        }
        cv.notify_all();
        for (size_t k; (k = next.fetch_add(1)) < N;) { f(k); done.fetch_add(1); }
        while (done.load(std::memory_order_acquire) < N) __builtin_ia32_pause();      // (microseconds: polled, not slept on)
        for (;;) {                                                                   // ... and no worker still holds the job
            std::lock_guard<std::mutex> g(m);
            if (busy == 0) { job = nullptr; n = 0; break; }
        }
    }
    ~NbPool() { { std::lock_guard<std::mutex> g(m); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
};
NbPool& nb_pool() { static NbPool p; return p; }
}

extern "C" {

hfm_model* hfm_create(int model_type, int n_collapsed, const int32_t* region_coverages, int n_regions,
                      int start_only, int avg_alignment_len, int window_len, const double* alpha16,
                      double max_high_mapq_ratio, double min_high_mapq_ratio) {
    if ((model_type != HF_MODEL_TRUNC_EXP_GAUSSIAN && model_type != HF_MODEL_GAUSSIAN &&
         model_type != HF_MODEL_NEGATIVE_BINOMIAL) || n_collapsed < 1 ||
        n_collapsed > KM || n_regions < 1 || n_regions > HF_MAXREGIONS || !region_coverages)
        return nullptr;
    hfm_model* m = new hfm_model();
    m->model_type = model_type; m->R = n_regions; m->K = n_collapsed;
    m->ncomp[3] = n_collapsed;                                   // hmm_flagger.c:179-183
    if (alpha16) std::memcpy(m->alpha, alpha16, sizeof(m->alpha));
    m->max_mapq = max_high_mapq_ratio; m->min_mapq = min_high_mapq_ratio;
    m->min_clip = 1.0;                                           // hmm_flagger.c:222
    const size_t R = (size_t) n_regions;
    m->trans.assign(R * 25, 0.0); m->lambda.assign(R, 1.0); m->trunc.assign(R, 0.0);
    m->mean.assign(R * S * KM, 0.0); m->var.assign(R * S * KM, 0.0); m->weight.assign(R * S * KM, 0.0);
    double median = region_coverages[0];                         // hmm_flagger.c:189-195
    if (start_only) median *= (double) window_len / avg_alignment_len;
    double base[S][KM] = {};
    base[0][0] = median * kErrBinding * 1.0;                     // hmm_flagger.c:213-220, initialRandomDev = 0
    base[1][0] = median * 0.5 * 1.0;
    base[2][0] = median * 1.0 * 1.0;
    for (int i = 0; i < n_collapsed; i++) base[3][i] = base[2][0] * (i + 2) * 1.0;
    for (int r = 0; r < n_regions; r++) {
        const double scale = (double) region_coverages[r] / median;  // hmm_flagger.c:199
        m->lambda[r] = 1.0;                                      // hmm_utils.c:1620
        m->trunc[r] = base[2][0] * scale * kTruncFraction;
        for (int s = 0; s < S; s++)
            for (int c = 0; c < m->ncomp[s]; c++) {              // hmm_utils.c:658-673, 733-741
                const double mu = base[s][c] * scale;            // hmm.c:43-47
                m->M(r, s, c) = mu;
                m->Vr(r, s, c) = mu * 1.0;
                m->W(r, s, c) = 1.0 / m->ncomp[s];
                if (model_type == HF_MODEL_NEGATIVE_BINOMIAL) {  // NegativeBinomial_constructByMean(mean, 1.5), hmm_utils.c:428-457, 1635-1639
                    const double var = mu * 1.5;
                    const double theta = mu / var;
                    const double rr = std::pow(mu, 2) / (var - mu);
                    m->M(r, s, c) = theta;
                    m->Vr(r, s, c) = -1 * rr * std::log(mu / var);
                }
            }
        for (int i = 0; i < 5; i++)                              // hmm_utils.c:2109-2128
            for (int j = 0; j < 5; j++)
                m->T(r, i, j) = i == j ? kDiagProb * (1.0 - kTermination)
                                       : (1.0 - kDiagProb) / (S - 1) * (1.0 - kTermination);
        for (int s = 0; s < S; s++) { m->T(r, S, s) = 1.0 / S; m->T(r, s, S) = kTermination; }
        m->T(r, S, S) = 0.0;
    }
    return m;
}

hfm_model* hfm_copy(const hfm_model* m) { return m ? new hfm_model(*m) : nullptr; }
void hfm_destroy(hfm_model* m) { delete m; }
int hfm_n_regions(const hfm_model* m) { return m->R; }
int hfm_max_comps(const hfm_model* m) { return m->K; }
int hfm_model_type(const hfm_model* m) { return m->model_type; }
double hfm_max_high_mapq_ratio(const hfm_model* m) { return m->max_mapq; }
double hfm_min_high_mapq_ratio(const hfm_model* m) { return m->min_mapq; }
double hfm_min_highly_clipped_ratio(const hfm_model* m) { return m->min_clip; }
double hfm_loglikelihood(const hfm_model* m) { return m->loglik; }

void hfm_params(const hfm_model* m, hf_params* out) {
    out->model_type = m->model_type; out->n_regions = m->R;
    for (int s = 0; s < S; s++) out->ncomp[s] = m->ncomp[s];
    std::memcpy(out->alpha, m->alpha, sizeof(out->alpha));
    out->trans = m->trans.data(); out->lambda = m->lambda.data(); out->trunc_point = m->trunc.data();
    out->mean = m->mean.data(); out->var = m->var.data(); out->weight = m->weight.data();
    out->nb_E = out->nb_P = out->nb_dig = out->nb_r = out->nb_beta = nullptr;
    out->nb_max_x = 0;
    if (!m->nb()) return;
    // negative_binomial: everything that depends on x only is tabulated here with the host libm, exactly as the
    // reference evaluates it (lgamma/exp/log of glibc, digamma in long double): emission value E[r][s][x]
    // (hmm_utils.c:480-520), component probabilities P[r][s][c][x], digamma table (:394-408), r and beta (:545-547)
    const int NX = kMaxCov + 1, K = m->K;
    const int NXF = m->nb_max_x + 1;                       // entries actually filled (the rest stay 0 and are never read)
    hfm_model* mm = const_cast<hfm_model*>(m);
    static double lgx1[kMaxCov + 1];                       // lgamma(x + 1), x = 0..250
    static bool lgx1_ready = false;
    if (!lgx1_ready) { for (int x = 0; x < NX; x++) lgx1[x] = lgamma(x + 1); lgx1_ready = true; }
    m->nb_E.assign((size_t) m->R * S * NX, 0.0);
    m->nb_P.assign((size_t) m->R * S * K * NX, 0.0);
    m->nb_dig.assign((size_t) m->R * S * K * NX, 0.0);
    m->nb_r.assign((size_t) m->R * S * K, 0.0);
    m->nb_beta.assign((size_t) m->R * S * K, 0.0);
    // one job per (region, state, component): ~250 lgamma + exp each (25 us) — on a small pool of host threads (the tables cost 0.1 ms
    // per EM iteration on one thread, a quarter of the negative-binomial step; each job writes its own rows: same values in any order)
    // (a component's x range in NB_SPLIT pieces: 9 jobs of 25 us on 8 threads would leave most of them idle half the time)
    constexpr int NB_SPLIT = 4;
    struct Job { int r, s, c, part; };
    std::vector<Job> jobs;
    for (int r = 0; r < m->R; r++)
        for (int s = 0; s < S; s++)
            for (int c = 0; c < m->ncomp[s]; c++)
                for (int part = 0; part < NB_SPLIT; part++) jobs.push_back({r, s, c, part});
    nb_pool().run(jobs.size(), [&](size_t j) {
                const int r = jobs[j].r, s = jobs[j].s, c = jobs[j].c, part = jobs[j].part;
                const int x_lo = (int) ((int64_t) NXF * part / NB_SPLIT), x_hi = (int) ((int64_t) NXF * (part + 1) / NB_SPLIT);
                const double theta = mm->M(r, s, c), lambda = mm->Vr(r, s, c), w = mm->W(r, s, c);
                const double rr = -1 * lambda / std::log(theta);
                const size_t pc = ((size_t) r * S + s) * K + c;
                if (part == 0) {
                    m->nb_r[pc] = rr;
                    m->nb_beta[pc] = -1 * theta / (1 - theta) - 1 / std::log(theta);
                }
                double* P = &m->nb_P[pc * NX];
                double* D = &m->nb_dig[pc * NX];
                // the x-independent terms once per component (pure functions of the same arguments: same doubles)
                int sg = 0;                                       // (lgamma_r: lgamma's value without the global signgam)
                const double lg_r = lgamma_r(rr, &sg), r_log_theta = rr * std::log(theta), log_1m_theta = std::log(1 - theta);
                for (int x = x_lo; x < x_hi; x++) {
                    double p = w * std::exp(lgamma_r(rr + x, &sg) - lg_r - lgx1[x] + r_log_theta + (double) x * log_1m_theta);
                    if (!(p != p) && p < 1e-40) p = 1e-40;       // NaN is kept: the E-step reports it if the value is used
                    P[x] = p;
                }
                if (part == 0) {                                  // the recurrence of the digamma table is sequential (and cheap): one piece does it
                    D[0] = (double) digammal_(rr);
                    for (int x = 1; x < NXF; x++) D[x] = D[x - 1] + 1.0 / (rr + x - 1);
                }
            });
    for (int r = 0; r < m->R; r++)
        for (int s = 0; s < S; s++)
            for (int x = 0; x < NXF; x++) {
                double tot = 0.0;                                 // Double_sum1DArray, component order
                for (int c = 0; c < m->ncomp[s]; c++) tot += m->nb_P[((((size_t) r * S + s) * K + c)) * NX + x];
                m->nb_E[((size_t) r * S + s) * NX + x] = tot;
            }
    out->nb_E = m->nb_E.data(); out->nb_P = m->nb_P.data(); out->nb_dig = m->nb_dig.data();
    out->nb_r = m->nb_r.data(); out->nb_beta = m->nb_beta.data();
    out->nb_max_x = m->nb_max_x;
}

void hfm_set_max_coverage(hfm_model* m, int max_x) {
    if (m) m->nb_max_x = max_x < 1 ? 1 : (max_x > kMaxCov ? kMaxCov : max_x);   // hf_params.nb_max_x <= 0 means "all of them"
}

// hmm_utils.c:949-956
static double trunc_exp_loglik(double lam, double b, double num, double den) {
    return den * std::log(lam) - den * std::log(1.0 - std::exp(-lam * b)) - num * lam;
}

// hmm_utils.c:969-1011 TruncExponential_estimateLambda — golden-section search on [0, truncPoint]
static double estimate_lambda(double trunc_point, double num, double den, double tol) {
    double a = 0.0, b = trunc_point;
    const double invphi = (std::sqrt(5.0) - 1.0) / 2.0, invphi2 = (3.0 - std::sqrt(5.0)) / 2.0;
    double h = b - a;
    if (h <= tol) return (b + a) / 2.0;
    const int n = (int) std::ceil(std::log(tol / h) / std::log(invphi));
    double c = a + invphi2 * h, d = a + invphi * h;
    double yc = trunc_exp_loglik(c, trunc_point, num, den), yd = trunc_exp_loglik(d, trunc_point, num, den);
    for (int k = 0; k < n - 1; k++) {
        if (yc > yd) {
            b = d; d = c; yd = yc; h = invphi * h; c = a + invphi2 * h;
            yc = trunc_exp_loglik(c, trunc_point, num, den);
        } else {
            a = c; c = d; yc = yd; h = invphi * h; d = a + invphi * h;
            yd = trunc_exp_loglik(d, trunc_point, num, den);
        }
    }
    return yc > yd ? (a + d) / 2.0 : (c + b) / 2.0;
}

// binding factor of (state, parameter, component): hmm_utils.c:191-238, 290-304, 143-157
static double binding(int s, int p, int c, bool nb = false) {
    if (p == 2) return 0.0;
    if (nb && p == 0) return 1.0;            // theta is bound across every state and component (hmm_utils.c:240-288)
    if (s == 0) return kErrBinding;
    if (s == 1) return 0.5;
    if (s == 2) return 1.0;
    return 2.0 + (double) c * 1.0;
}

int hfm_estimate(hfm_model* m, const double* stats, double tol) {
    bool converged = true;
    m->loglik = stats[0];
    const int K = m->K;
    const int64_t stride = hf_region_stride(K);
    for (int r = 0; r < m->R; r++) {
        const double* st = stats + 1 + r * stride;
        auto num = [&](int s, int p, int c) { return st[((s * 3 + p) * 2 + 0) * K + c]; };
        auto den = [&](int s, int p, int c) { return st[((s * 3 + p) * 2 + 1) * K + c]; };
        // --- emissions: hmm_utils.c:1860-1903, one parameter type at a time (1817-1858) ---
        for (int p = 0; p < 3; p++) {
            double bnum = 0.0, bden = 0.0;                        // bound estimator, :1791-1815
            for (int s = 0; s < S; s++) {
                if (!m->gaussian_state(s)) continue;
                for (int c = 0; c < m->ncomp[s]; c++) {
                    const double f = binding(s, p, c, m->nb());
                    if (0.0 < f) { bnum += num(s, p, c) / f; bden += den(s, p, c); }
                }
            }
            const double bound = bden == 0 ? 0.0 : bnum / bden;   // :76-92
            for (int s = 0; s < S; s++) {
                if (!m->gaussian_state(s)) continue;
                for (int c = 0; c < m->ncomp[s]; c++) {
                    const double f = binding(s, p, c, m->nb());
                    double est, count;
                    if (0.0 < f) { est = bound * f; count = bden; }
                    else { count = den(s, p, c); est = count == 0 ? 0.0 : num(s, p, c) / den(s, p, c); }
                    if (kMinCountForUpdate < count) {             // :1846, 842-859
                        double& slot = p == 0 ? m->M(r, s, c) : p == 1 ? m->Vr(r, s, c) : m->W(r, s, c);
                        const double old = slot;
                        slot = est;
                        const double diff = 1.0e-4 < old ? std::fabs(est / old - 1.0) : 0.0;
                        converged &= diff < tol;
                    }
                }
            }
        }
        if (m->model_type == HF_MODEL_TRUNC_EXP_GAUSSIAN) {       // :1872-1884
            const double n0 = num(0, 0, 0), d0 = den(0, 0, 0);
            const double est = d0 == 0 ? 0.0 : estimate_lambda(m->trunc[r], n0, d0, 1e-6);
            if (kMinCountForUpdate < d0) {                        // :1036-1054
                const double old = m->lambda[r];
                m->lambda[r] = est;
                const double diff = 1.0e-4 < old ? std::fabs(est / old - 1.0) : 0.0;
                converged &= diff < tol;
            }
            m->trunc[r] = m->M(r, 2, 0) * kTruncFraction;
        }
        // --- transitions: hmm_utils.c:2185-2219 ---
        const double* cnt = st + 24 * K;
        for (int i = 0; i < S; i++) {
            double row = 0.0;
            for (int j = 0; j < S; j++) row += cnt[i * 4 + j] + kPseudoCount;
            for (int j = 0; j < S; j++) {
                const double old = m->T(r, i, j);
                const double nv = (cnt[i * 4 + j] + kPseudoCount) / row * (1.0 - kTermination);
                m->T(r, i, j) = nv;
                const double diff = 1.0e-6 < old ? std::fabs(nv / old - 1.0) : 0.0;
                converged &= diff < tol;
            }
        }
        for (int i = 0; i < S; i++) m->T(r, i, S) = kTermination;
        for (int j = 0; j < S; j++) m->T(r, S, j) = 1.0 / S;
        m->T(r, S, S) = 0.0;
    }
    return converged ? 1 : 0;
}

// ---- TSV writers: hmm.c:137-239 ----
static const char* kStateNames[5] = {"Err", "Dup", "Hap", "Col", "Msj"};

int hfm_write_transition_tsv(const hfm_model* mc, const char* path) {
    hfm_model* m = const_cast<hfm_model*>(mc);
    FILE* f = std::fopen(path, "w");
    if (!f) return -1;
    std::fprintf(f, "#Region\tState\tErr\tDup\tHap\tCol\tEnd\n");
    for (int r = 0; r < m->R; r++)
        for (int i = 0; i < S + 1; i++) {
            std::fprintf(f, "%d\t%s", r, i < S ? kStateNames[i] : "Start");
            for (int j = 0; j < S + 1; j++) std::fprintf(f, "\t%.5e", m->T(r, i, j));
            std::fprintf(f, "\n");
        }
    std::fclose(f);
    return 0;
}

int hfm_write_emission_tsv(const hfm_model* mc, const char* path) {
    hfm_model* m = const_cast<hfm_model*>(mc);
    FILE* f = std::fopen(path, "w");
    if (!f) return -1;
    std::fprintf(f, "#State\tDistribution\tComponents\tParameter");
    for (int r = 0; r < m->R; r++) std::fprintf(f, "\tValues_Region_%d", r);
    std::fprintf(f, "\n");
    for (int s = 0; s < S; s++) {
        const bool te = !m->gaussian_state(s);
        const int np = te ? 2 : 3;                                // hmm_utils.c:1539-1575
        for (int p = 0; p < np; p++) {
            const char* pname = te ? (p == 0 ? "Mean" : "Trunc_Point") : (p == 0 ? "Mean" : p == 1 ? "Var" : "Weight");
            std::fprintf(f, "%s\t%s\t%d\t%s", kStateNames[s], te ? "Truncated Exponential" : m->nb() ? "Negative Binomial" : "Gaussian",
                         te ? 1 : m->ncomp[s], pname);
            for (int r = 0; r < m->R; r++) {
                std::fprintf(f, "\t");
                if (te) std::fprintf(f, "%.5e", p == 0 ? 1.0 / m->lambda[r] : m->trunc[r]);  // :1056-1069
                else if (m->nb() && p < 2)                       // logged as mean and variance, hmm_utils.c:463-473, 1560-1570
                    for (int c = 0; c < m->ncomp[s]; c++) {
                        const double theta = m->M(r, s, c), lambda = m->Vr(r, s, c);
                        const double rr = -1 * lambda / std::log(theta);
                        std::fprintf(f, c ? ",%.5e" : "%.5e", p == 0 ? rr * (1 - theta) / theta : rr * (1 - theta) / std::pow(theta, 2));
                    }
                else
                    for (int c = 0; c < m->ncomp[s]; c++)
                        std::fprintf(f, c ? ",%.5e" : "%.5e", p == 0 ? m->M(r, s, c) : p == 1 ? m->Vr(r, s, c) : m->W(r, s, c));
            }
            std::fprintf(f, "\n");
        }
    }
    std::fclose(f);
    return 0;
}

int64_t hfm_param_len(const hfm_model* m) { return (int64_t) m->R * (25 + 2 + 3 * S * KM); }

void hfm_get_param_vector(const hfm_model* m, double* out) {
    for (int r = 0; r < m->R; r++) {
        std::memcpy(out, &m->trans[(size_t) r * 25], 25 * 8); out += 25;
        *out++ = m->lambda[r]; *out++ = m->trunc[r];
        std::memcpy(out, &m->mean[(size_t) r * S * KM], S * KM * 8); out += S * KM;
        std::memcpy(out, &m->var[(size_t) r * S * KM], S * KM * 8); out += S * KM;
        std::memcpy(out, &m->weight[(size_t) r * S * KM], S * KM * 8); out += S * KM;
    }
}

void hfm_set_param_vector(hfm_model* m, const double* in) {
    for (int r = 0; r < m->R; r++) {
        std::memcpy(&m->trans[(size_t) r * 25], in, 25 * 8); in += 25;
        m->lambda[r] = *in++; m->trunc[r] = *in++;
        std::memcpy(&m->mean[(size_t) r * S * KM], in, S * KM * 8); in += S * KM;
        std::memcpy(&m->var[(size_t) r * S * KM], in, S * KM * 8); in += S * KM;
        std::memcpy(&m->weight[(size_t) r * S * KM], in, S * KM * 8); in += S * KM;
    }
}

void hfm_set_loglikelihood(hfm_model* m, double ll) { m->loglik = ll; }

void hfm_scale_initial_means(hfm_model* m, double f) {
    for (int r = 0; r < m->R; r++) {
        for (int s = 0; s < S; s++)
            for (int c = 0; c < m->ncomp[s]; c++) {
                const double g = s == 3 ? f * f : f;
                if (m->nb()) {                                       // re-derive theta, lambda from the scaled mean
                    const double theta = m->M(r, s, c), rr0 = -1 * m->Vr(r, s, c) / std::log(theta);
                    const double mu = rr0 * (1 - theta) / theta * g, var = mu * 1.5;
                    m->M(r, s, c) = mu / var;
                    m->Vr(r, s, c) = -1 * (std::pow(mu, 2) / (var - mu)) * std::log(mu / var);
                    continue;
                }
                m->M(r, s, c) *= g;
                m->Vr(r, s, c) = m->M(r, s, c) * 1.0;
            }
        m->trunc[r] = m->M(r, 2, 0) * kTruncFraction;
    }
}

// hmm.c:80-87 HMM_isFeasible (hmm_utils.c:685-694, 920-925, 2130-2139)
int hfm_is_feasible(const hfm_model* mc) {
    hfm_model* m = const_cast<hfm_model*>(mc);
    bool ok = true;
    for (int r = 0; r < m->R; r++) {
        for (int s = 0; s < S; s++) {
            if (!m->gaussian_state(s)) { ok &= 0 < m->lambda[r]; ok &= 0 < m->trunc[r]; continue; }
            for (int c = 0; c < m->ncomp[s]; c++) {
                ok &= 0 < m->M(r, s, c);
                if (m->nb()) ok &= m->M(r, s, c) < 1;                 // 0 < theta < 1, hmm_utils.c:367-376
                ok &= 0 < m->Vr(r, s, c);
                ok &= (0 <= m->W(r, s, c)) && (m->W(r, s, c) <= 1);
            }
        }
        for (int i = 0; i < S; i++)
            for (int j = 0; j < S; j++)
                if (m->T(r, i, j) < 0 || 1 < m->T(r, i, j)) return 0;
    }
    return ok ? 1 : 0;
}

// hf_warmup + one miniature EM per kernel family: every kernel of the default pass (the emission tables with the parameter block in
// the kernel arguments and by copy, the one-launch segment kernel in both modes, pair sums, the row statistics of every component-count
// template) is launched once on a synthetic track of 2 x 1 100 windows, so that the caller's first real pass does not pay the
// runtime's first-launch work (the command line's first E-step took 0.3-0.4 ms against 0.114 ms for the third:
// profiles/r04a_cli_wall.txt).  A few milliseconds, meant for the thread that brings the runtime up while the input is read.
int hfm_warmup_pipeline(int device) {
    int rc = hf_warmup(device);
    if (rc != HF_OK) return rc;
    const int C = 2, T = 1100, W = 4000;
    std::vector<int64_t> off = {0, T, 2 * T};
    std::vector<uint16_t> cov((size_t) C * T), mapq((size_t) C * T), clip((size_t) C * T, 0);
    std::vector<uint64_t> ann((size_t) C * T, 0);
    std::vector<int32_t> cs = {0, T * W}, ce = {T * W - 1, 2 * T * W - 1}, cl = {2 * T * W, 2 * T * W};
    struct Cfg { int K, R; };
    const Cfg cfgs[] = {{3, 1}, {6, 1}, {10, 1}, {6, 2}};
    for (const Cfg& cf : cfgs) {
        for (size_t t = 0; t < cov.size(); t++) {
            cov[t] = (uint16_t) (14 + (t * 7) % 13 + ((t / 97) % 5 == 0 ? 20 : 0));
            mapq[t] = cov[t];
            ann[t] = cf.R > 1 ? (uint64_t) ((t / 300) % (size_t) cf.R) << 58 : 0;
        }
        hf_windows w;
        std::memset(&w, 0, sizeof w);
        w.n_windows = (int64_t) cov.size(); w.n_chunks = C; w.chunk_off = off.data();
        w.cov = cov.data(); w.mapq = mapq.data(); w.clip = clip.data(); w.annot = ann.data();
        w.chunk_s = cs.data(); w.chunk_e = ce.data(); w.chunk_ctg_len = cl.data();
        w.window_len = W; w.mean_read_len = 15000; w.adjust_contig_ends = 1; w.min_read_frac = 0.95;
        w.max_high_mapq_ratio = 0.25; w.min_high_mapq_ratio = 0.75;
        const int32_t regcov[2] = {20, 24};
        hfm_model* m = hfm_create(HF_MODEL_TRUNC_EXP_GAUSSIAN, cf.K, regcov, cf.R, 0, 15000, W, nullptr, 0.25, 0.75);
        if (!m) return HF_E_ARG;
        w.min_highly_clipped_ratio = hfm_min_highly_clipped_ratio(m);
        hf_ctx* ctx = nullptr;
        rc = hf_create(&w, cf.R, hfm_max_comps(m), device, HF_ALGO_SCAN, &ctx);
        if (rc == HF_OK) {
            std::vector<double> st((size_t) hf_chunk_stats_len(ctx));
            int cv = 0;
            rc = hf_em_iterate(ctx, m, HF_MODE_FULL, 1, 1e-3, st.data(), &cv, nullptr);
            if (rc == HF_OK) rc = hf_em_iterate(ctx, m, HF_MODE_FORWARD_ONLY, 0, 1e-3, st.data(), &cv, nullptr);
            if (rc == HF_OK) rc = hf_em_iterate(ctx, m, HF_MODE_FULL, 0, 1e-3, st.data(), &cv, nullptr);
            std::vector<int8_t> lab(cov.size());
            if (rc == HF_OK) rc = hf_get_labels(ctx, lab.data());
            hf_destroy(ctx);
        }
        hfm_destroy(m);
        if (rc != HF_OK) return rc;
    }
    return HF_OK;
}

// hmm_flagger.c:105-111 + 1012-1013
int hfm_best_collapsed_comps(const uint16_t* cov, int64_t n, const int32_t* region_coverages, int n_regions) {
    int maxc = 0;
    for (int64_t i = 0; i < n; i++) if (maxc < cov[i]) maxc = cov[i];
    int minr = region_coverages[0];
    for (int r = 1; r < n_regions; r++) if (region_coverages[r] < minr) minr = region_coverages[r];
    if (minr == 0) return -1;
    int k = maxc / minr + 1;
    return k < 2 ? 2 : (k > 10 ? 10 : k);
}

// hmm_flagger.c:491-515 + data_types.c:490-518 (tab-separated, no header, last character of each
// line dropped before splitting)
int hfm_read_alpha_tsv(const char* path, double* alpha16) {
    FILE* f = std::fopen(path, "r");
    if (!f) return -1;
    for (int i = 0; i < 16; i++) alpha16[i] = 0.0;
    char* line = nullptr; size_t cap = 0; int i = 0;
    while (i < 4 && getline(&line, &cap, f) != -1) {
        size_t L = std::strlen(line);
        if (L > 0) line[L - 1] = '\0';
        int j = 0; char* save = nullptr;
        for (char* tok = strtok_r(line, "\t", &save); tok && j < 4; tok = strtok_r(nullptr, "\t", &save))
            alpha16[i * 4 + j++] = std::atof(tok);
        i++;
    }
    std::free(line);
    std::fclose(f);
    for (int k = 0; k < 16; k++) if (1.0 < alpha16[k] || alpha16[k] < 0.0) return -2;
    return 0;
}


} // extern "C"

// ---- SQUAREM (hmm.c:820-1098).  Parameters are visited in the reference's iterator order: per region, per
// state, per component, per parameter (mean, var, weight | lambda), then the 4x4 transition block. ----
struct hfm_squarem {
    hfm_model m0, prime, rates_r, rates_v;
    double alpha = 0.0;
    hfm_squarem(const hfm_model& a) : m0(a), prime(a), rates_r(a), rates_v(a) {}
    template <class F> static void for_each_param(hfm_model& m, F f) {   // f(double& slot, int region)
        for (int r = 0; r < m.R; r++) {
            for (int s = 0; s < S; s++) {
                if (!m.gaussian_state(s)) { f(m.lambda[r], 0); continue; }   // trunc point is not iterated (:1132-1134)
                for (int c = 0; c < m.ncomp[s]; c++) { f(m.M(r, s, c), 0); f(m.Vr(r, s, c), 0); f(m.W(r, s, c), 0); }
            }
            for (int i = 0; i < S; i++) for (int j = 0; j < S; j++) f(m.T(r, i, j), 1);
        }
    }
    static std::vector<double*> slots(hfm_model& m) {
        std::vector<double*> v;
        for_each_param(m, [&](double& x, int) { v.push_back(&x); });
        return v;
    }
    void compute_values() {                                              // hmm.c:921-997
        auto p0 = slots(m0), pp = slots(prime), pr = slots(rates_r), pv = slots(rates_v);
        for (size_t i = 0; i < p0.size(); i++)
            *pp[i] = *p0[i] - 2 * *pr[i] * alpha + *pv[i] * std::pow(alpha, 2);
        for (int r = 0; r < prime.R; r++) {                              // HMM_normalizeWeightsAndTransitionRows, hmm.c:89-94
            for (int s = 0; s < S; s++) {
                if (!prime.gaussian_state(s)) continue;
                double sum = 0.0;
                for (int c = 0; c < prime.ncomp[s]; c++) sum += prime.W(r, s, c);
                if (0.0 < sum) { const double k = 1.0 / sum; for (int c = 0; c < prime.ncomp[s]; c++) prime.W(r, s, c) *= k; }
            }
            for (int i = 0; i < S; i++) {                                // hmm_utils.c:2165-2183
                double row = 0.0;
                for (int j = 0; j < S; j++) row += prime.T(r, i, j);
                for (int j = 0; j < S; j++) prime.T(r, i, j) = prime.T(r, i, j) / row * (1.0 - kTermination);
            }
            for (int i = 0; i < S; i++) prime.T(r, i, S) = kTermination;
            prime.T(r, S, S) = 0.0;
        }
    }
    hfm_model* shrink_once(double margin) {                              // hmm.c:871-884
        alpha = (alpha - 1) / 2;
        if (alpha > (-1 - margin)) { alpha = -1.0; prime = m0; return &prime; }
        compute_values();
        return &prime;
    }
};

extern "C" {

hfm_squarem* hfm_squarem_create(const hfm_model* m0, const hfm_model* m1c, const hfm_model* m2c) {
    hfm_squarem* a = new hfm_squarem(*m0);
    hfm_model m1(*m1c), m2(*m2c);
    auto p0 = hfm_squarem::slots(a->m0), p1 = hfm_squarem::slots(m1), p2 = hfm_squarem::slots(m2);
    auto pr = hfm_squarem::slots(a->rates_r), pv = hfm_squarem::slots(a->rates_v);
    double num = 0.0, den = 0.0;                                         // hmm.c:999-1098
    for (size_t i = 0; i < p0.size(); i++) {
        const double r = *p1[i] - *p0[i];
        const double v = *p2[i] - *p1[i] - r;
        num += std::pow(r, 2);
        den += std::pow(v, 2);
        *pr[i] = r; *pv[i] = v;
    }
    a->alpha = -1 * std::sqrt(num / den);
    if (a->alpha > -1) a->alpha = -1;
    return a;
}
void hfm_squarem_destroy(hfm_squarem* a) { delete a; }
double hfm_squarem_alpha(const hfm_squarem* a) { return a->alpha; }

hfm_model* hfm_squarem_model_prime(hfm_squarem* a) {
    a->compute_values();
    hfm_model* p = &a->prime;
    while (!hfm_is_feasible(p)) p = a->shrink_once(1e-2);
    return p;
}

hfm_model* hfm_squarem_shrink(hfm_squarem* a) {
    hfm_model* p = a->shrink_once(1e-2);
    while (!hfm_is_feasible(p)) p = a->shrink_once(1e-2);
    return p;
}

} // extern "C"
