// hf_squarem.h — one accelerated iteration of the EM loop (hmm_flagger.c:382-416) on top of the C ABI.
#pragma once
#include "../../include/hmm_flagger_hip.h"
#include "../../include/hmm_flagger_model.h"
#include <cstdio>
#include <vector>

// On entry `stats` holds the statistics of the E-step at *model (= model 0).  On exit *model is the accelerated
// model (model prime) and `stats` the statistics of ITS E-step, ready for the caller's M-step.
// estep(model, mode, stats_out) runs one pass; returns HF_OK or an HF_E_* code.
template <class EStep>
int squarem_iteration(hfm_model** model, std::vector<double>& stats, double tol, EStep estep, int* passes) {
    hfm_model* m = *model;
    hfm_set_loglikelihood(m, stats[0]);
    hfm_model* m0 = hfm_copy(m);                                   // SquareAccelerator_setModel0, :388
    hfm_estimate(m, stats.data(), tol);                            // :390
    hfm_model* m1 = hfm_copy(m);                                   // :391
    int rc = estep(m, HF_MODE_FULL, stats.data());                 // :393-394
    if (rc != HF_OK) { hfm_destroy(m0); hfm_destroy(m1); return rc; }
    ++*passes;
    hfm_estimate(m, stats.data(), tol);                            // :395
    hfm_squarem* acc = hfm_squarem_create(m0, m1, m);              // :396-398, hmm.c:886-918
    const double ll0 = hfm_loglikelihood(m0);
    hfm_model* prime = hfm_squarem_model_prime(acc);
    rc = estep(prime, HF_MODE_FORWARD_ONLY, stats.data());
    // hmm.c:904-914: shrink alpha until the likelihood is not below model 0's.  A shrink that comes within 1e-2 of -1 sets
    // alpha = -1 and makes prime a COPY OF MODEL 0 (hmm.c:871-884) — the loop's fixed point: its likelihood is ll0 by
    // definition, and the reference leaves the loop there because both sides come from the same code path.  Here ll0 comes
    // from a FULL pass and stats[0] from a FORWARD_ONLY pass, whose summation orders are matched by construction but not by
    // contract: never shrink again once the fixed point is reached.  (An alpha that STARTS at -1, hmm.c:1095-1097, is not
    // the fixed point: prime is then the extrapolation with alpha = -1, i.e. model 2, and must still be tested and shrunk.)
    bool fixed_point = false;
    while (rc == HF_OK && stats[0] < ll0 && !fixed_point) {
        prime = hfm_squarem_shrink(acc);
        fixed_point = hfm_squarem_alpha(acc) == -1.0;
        rc = estep(prime, HF_MODE_FORWARD_ONLY, stats.data());
    }
    if (rc == HF_OK) {
        std::fprintf(stderr, "Computed alpha rate for accelerating EM = %.4f\n", hfm_squarem_alpha(acc));
        rc = estep(prime, HF_MODE_FULL, stats.data());             // :400-401
        ++*passes;
    }
    if (rc == HF_OK) {
        hfm_model* next = hfm_copy(prime);                         // :403-405
        hfm_destroy(m);
        *model = next;
    }
    hfm_squarem_destroy(acc);
    hfm_destroy(m0); hfm_destroy(m1);
    return rc;
}
