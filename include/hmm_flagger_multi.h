/*
 * hmm_flagger_multi.h — C ABI of the multi-GPU E-step: chunks statically sharded over the GPUs of one node, one
 * RCCL all-gather of sufficient statistics per EM iteration over xGMI (BASELINE.json configs[3]).
 *
 * What it replaces in mobinasri/flagger: the thread pool over chunks and the in-order merge of
 * EM_runOneIterationForList / EM_runForwardForList (programs/submodules/hmm/hmm.c:739-763, 790-816): chunks are
 * independent inside a pass, the only coupling is `HMM_mergeEstimators` over the chunk list.
 *
 * Two levels:
 *   hf_comm   a communicator rank (RCCL; one per GPU) + the all-gather — for hosts that run one PROCESS per GPU
 *             (hf_comm_unique_id on rank 0, ship the 128 bytes to the others, hf_comm_init_rank everywhere) and
 *             drive hf_estep / hf_finish_exchange themselves;
 *   hf_multi  the whole thing inside ONE process: one host thread, one stream and one RCCL rank per device; what
 *             `hmm_flagger --gpus N` uses.  Every rank holds the same statistics bits after the exchange, runs on
 *             its own shard only, and the labels stay sharded until they are asked for.
 *
 * Exchanges (both ONE collective per pass, latency-bound at these sizes):
 *   HF_EXCHANGE_CHUNKS  per-chunk vectors all-gathered (in place: the pass writes them into this rank's slot), then
 *                       every rank sums ALL chunks in chunk-list order — the reference's own merge order, so an
 *                       N-GPU run is bit-identical to a 1-GPU run of the per-chunk statistics (HF_STATS_CHUNKS):
 *                       same EM trajectory, same final_flagger_prediction.bed.  Default.
 *   HF_EXCHANGE_RANKS   every rank reduces its own shard first (statistics by emission row), one vector per rank is
 *                       all-gathered and summed in rank order: fewer bytes and a faster statistics path; equal to the
 *                       1-GPU result up to the rounding of a different summation order (1e-12 relative).
 * Each rank's device error flags travel in the same buffer, so HF_E_SCALE / HF_E_NAN are reported by all ranks together.
 */
#ifndef HMM_FLAGGER_MULTI_H
#define HMM_FLAGGER_MULTI_H

#include "hmm_flagger_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { HF_EXCHANGE_CHUNKS = 0, HF_EXCHANGE_RANKS = 1 };
enum { HF_TRANSPORT_RCCL = 0,      /* one rank per GPU over xGMI: the product path */
       HF_TRANSPORT_LOOPBACK = 1   /* TEST ONLY: all ranks share ONE device and gather with device-to-device copies
                                      behind a host barrier (RCCL refuses two ranks on one GPU): exercises sharding,
                                      the exchange layout and the ordered reduction on a 1-GPU box */ };

#define HF_COMM_ID_BYTES 128
typedef struct hf_comm hf_comm;

int hf_comm_unique_id(void *id_out /* HF_COMM_ID_BYTES */);
/* one rank of a multi-process job; blocks until all n_ranks have called it (ncclCommInitRank) */
int hf_comm_init_rank(int n_ranks, int rank, int device, const void *id, hf_comm **out);
/* all ranks of a single-process job at once (ncclCommInitAll); devices must be distinct */
int hf_comm_init_all(int n, const int *devices, hf_comm **out /* [n] */);
/* n loopback ranks on one device (HF_TRANSPORT_LOOPBACK) */
int hf_comm_init_loopback(int n, int device, hf_comm **out /* [n] */);
void hf_comm_destroy(hf_comm *c);
int hf_comm_rank(const hf_comm *c);
int hf_comm_size(const hf_comm *c);
/* every rank contributes `count` doubles from send_dev; recv_dev gets size*count doubles in rank order; in place when
 * send_dev == recv_dev + rank*count.  Device pointers, asynchronous on `stream`; collective: every rank must call it. */
int hf_comm_allgather(hf_comm *c, const double *send_dev, double *recv_dev, int64_t count, void *stream);
const char *hf_comm_last_error(void);

/* Contiguous runs of the chunk list balanced by window count: bounds[r]..bounds[r+1] is rank r's run (world+1 ints). */
int hf_shard_bounds(const int64_t *chunk_off, int32_t n_chunks, int world, int32_t *bounds);

typedef struct hf_multi hf_multi;
/* Shard `w` over the devices (hf_shard_bounds), upload every shard on its own thread, set up the communicator and the
 * exchange buffers.  Fails with HF_E_NOGPU when fewer than n_devices (distinct, HF_TRANSPORT_RCCL) devices are visible. */
int hf_multi_create(const hf_windows *w, int n_regions, int max_comps, int n_devices, const int *devices, int algo,
                    int exchange, int transport, hf_multi **out);
/* The same object for hosts that run one PROCESS per GPU (torch.distributed.run, mpirun: what the reference's
 * EM_runOneIterationForList would be called from under such a launcher): this process is rank `rank` of `world` on `device`
 * and holds that rank's shard only; unique_id = rank 0's hf_comm_unique_id(), carried to the other ranks by the launcher.
 * Collective (every rank calls it, with the same chunk list).  hf_multi_estep runs on the calling thread and every rank
 * gets the same reduced vector; hf_multi_get_labels / _get_posterior fill this rank's windows only (at their global
 * positions: hf_multi_local_first_window, hf_multi_local_windows); hf_multi_rank_stats answers for the own rank. */
int hf_multi_create_rank(const hf_windows *w, int n_regions, int max_comps, int world, int rank, int device, int algo,
                         int exchange, const void *unique_id, hf_multi **out);
hf_ctx *hf_multi_local_ctx(hf_multi *m);   /* borrowed: the rank's E-step context (profiling switches, getters); NULL for hf_multi_create objects */
int64_t hf_multi_local_first_window(const hf_multi *m);
int64_t hf_multi_local_windows(const hf_multi *m);
void hf_multi_destroy(hf_multi *m);
/* One pass on every shard + the exchange + the ordered reduction: EM_runOneIterationForList (HF_MODE_FULL) /
 * EM_runForwardForList (HF_MODE_FORWARD_ONLY) for the whole chunk list.  stats_host gets the reduced vector
 * (hf_stats_len doubles; every rank computed the same bits).  Returns HF_OK or the HF_E_* every rank agreed on. */
int hf_multi_estep(hf_multi *m, const hf_params *p, int mode, double *stats_host);
/* hf_multi_estep with the model's current parameters + HMM_estimateParameters on the returned vector (every rank: the same
 * vector, the same M-step): what hf_em_iterate is for one context.  struct hfm_model: include/hmm_flagger_model.h */
struct hfm_model;
int hf_multi_em_iterate(hf_multi *m, struct hfm_model *model, int mode, int do_mstep, double tol, double *stats_host, int *converged);
int hf_multi_get_labels(hf_multi *m, int8_t *labels_host);                              /* [n_windows], list order */
int hf_multi_get_posterior(hf_multi *m, int64_t first, int64_t n, double *post_host);   /* [n][4] */
int hf_multi_world(const hf_multi *m);
int hf_multi_comm_ranks(const hf_multi *m);   /* ranks of the RCCL communicator, as ncclCommCount reports them (loopback: the group size) */
int64_t hf_multi_stats_len(const hf_multi *m);
/* windows / chunks of rank r's shard (reporting) */
int64_t hf_multi_shard_windows(const hf_multi *m, int r);
int32_t hf_multi_shard_chunks(const hf_multi *m, int r);
/* debug: the statistics vector rank r computed in the last pass (must equal rank 0's bit for bit) */
int hf_multi_rank_stats(hf_multi *m, int r, double *stats_host);
const char *hf_multi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
