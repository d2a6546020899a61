#!/usr/bin/env python
"""bench.py — windows/s through EM + decode on BASELINE.json's workload, with the roofline of the
dominant kernel and a CPU baseline beside it.

One "step" = one EM pass over the whole synthetic diploid track: E-step on the GPU (emission,
forward, backward, posterior decode, sufficient statistics, ordered reduction), the statistics
vector to the host, the M-step, and the next parameter block back up — i.e. exactly what
runHMMFlagger repeats (hmm_flagger.c:337-445).  The windows are resident in HBM before the timed
region.  With --gpus N the chunk list is sharded across N ranks (one process per GPU over RCCL: the driver's
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, or — when WORLD_SIZE is not set —
bench.py re-executes itself under that launcher) and the statistics are all-gathered every step
(flagger_amd/dist.py).  It refuses to run when fewer than N GPUs are visible: it never reports a run of fewer
ranks as N.  --scaling strong (default) shards the fixed BASELINE workload (configs[3]); --scaling weak gives every
rank a whole configs[2] genome (N genomes in one EM: the statistics exchange is the same, the per-GPU work fixed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_WINDOW = 17.0   # 16-byte packed window record (chunk.c:669-697) + 1 label byte, SURVEY §8d
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector (VALU) peak: half of the 157.3 TF f32 vector peak of /opt/skills/guides/MI355X_MICROARCH.md
                                # (256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz)


def effective_cores() -> int:
    """Logical CPUs this process may actually use: min(os.cpu_count(), cgroup v2 cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def cpu_baseline(store_full, K, alpha, cores):
    """Oracle (CPU port of the reference path) on a bounded sample of the same workload."""
    import numpy as np
    from oracle_py import Oracle
    from flagger_amd import synth
    sample = store_full                    # the full workload: needs >= `cores` chunks to keep every core busy
    orc = Oracle(sample, 0, K, alpha, threads=cores)
    passes = 3
    orc.run_iteration()                    # warm-up (allocates f/b)
    t0 = time.perf_counter()
    for _ in range(passes):
        assert orc.run_iteration() == 0
        orc.estimate_parameters(1e-3)
    dt = time.perf_counter() - t0
    n = sample.n_windows
    orc.close()
    return {"value": n * passes / dt, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{passes} EM passes (E-step + M-step) over the same workload ({n} windows, "
                      f"{sample.n_chunks} chunks), oracle/ C port with {cores} threads over chunks "
                      f"(~{n * passes / 2.3e5:.0f} s of single-core work)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of a quarter of a second behind a warm-up long enough for the GPU's clocks and the HIP runtime's
    # one-off work (a ~30 ms stall some 80 ms into a process's first launches, profiles/tools/r03_step_drift.py): 20 steps behind
    # 3 warm-up steps measure 0.127 ms per step, behind 1000 warm-up steps 0.121, 2000 steps behind 1000: 0.118 (same box)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink contig lengths (testing only; 1.0 = BASELINE workload)")
    ap.add_argument("--algo", choices=["scan", "seq"], default="scan")
    ap.add_argument("--config", type=int, choices=[2, 4, 5, 6], default=2,
                    help="BASELINE.json configs[n]: 2 = the headline workload (default); 4 = ONT-R10 preset, 7 bias regions, 8 kb windows; "
                         "5 = not a BASELINE config: configs[4] with coverage spread over 0..250 (worst case for the emission tables); "
                         "6 = not a BASELINE config: configs[2] with over-dispersed (negative-binomial, variance = 3 x mean) coverage")
    ap.add_argument("--overdispersion", type=float, default=3.0, help="config 6 only: variance / mean of the negative-binomial coverage")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-em-run", action="store_true", help="skip the `em_run` leg (a real EM to convergence in a fresh context, and the cold "
                                                             "command line) that follows the timed region")
    ap.add_argument("--creates", type=int, default=9, help="fresh contexts of the em_run leg's hf_create timing (median / min / max reported)")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="the dominant kernel is bracketed by a pair of HIP events in every n-th timed step; default 0 = max(1, min(32, steps // 5)): "
                         "at least 5 samples however short the timed region is (the driver's --steps 20: every 4th step; a pair costs the pass "
                         "it is in 10 us of turn-around, profiles/r06_kstamp.txt — at every 2nd step that was 5 % of the timed region); the line "
                         "reports the MEDIAN of the samples and their number")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="no HIP events inside the timed region; the dominant kernel's duration then comes from the untimed "
                         "passes that follow — for comparing launch paths, not the default")
    ap.add_argument("--exchange", choices=["ranks", "chunks"], default="ranks",
                    help="multi-GPU exchange: one statistics vector per rank, all-gathered and summed in rank order (default: BASELINE "
                         "north_star's single collective over the EM sufficient statistics; statistics by emission row on every rank, "
                         "equal to a one-GPU run up to the rounding of the order), or the per-chunk vectors summed in chunk-list order "
                         "(statistics, EM trajectory and BED labels identical for every N bit for bit by construction; slower per-chunk "
                         "statistics kernels: what `hmm_flagger --gpus N` defaults to since round 5 — its files must not depend on N under --accelerate)")
    ap.add_argument("--collective", choices=["native", "torch"], default="native",
                    help="multi-GPU path: native = pass + RCCL all-gather + ordered reduction in ONE library call per EM pass on the "
                         "pass's own stream (hf_multi_create_rank; torch.distributed only carries the RCCL id at start-up); "
                         "torch = the exchange through torch.distributed.all_gather_into_tensor (flagger_amd/dist.py)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: the BASELINE workload sharded over the GPUs (configs[3]); weak: one whole configs[2] genome per GPU")
    ap.add_argument("--no-weak-leg", action="store_true",
                    help="with --gpus N > 1 and strong scaling the line also carries a `weak_scaling` object (one whole genome per "
                         "GPU, timed after the main region); this switch skips it")
    ap.add_argument("--no-second-exchange", action="store_true",
                    help="on the multi-GPU path the line also carries `other_exchange` (the exchange that was not timed as the headline) "
                         "and `n_invariance` (a 5 % copy of the workload through --exchange chunks on all ranks against one context on "
                         "rank 0: log-likelihoods bit-identical, labels identical); this switch skips both")
    ap.add_argument("--force-weak-leg", action="store_true",
                    help="run the weak-scaling leg even with one rank (exercises that code path on a 1-GPU box; with --dist-path)")
    ap.add_argument("--dist-path", action="store_true",
                    help="take the multi-GPU code path (process group, all-gather, indexed reduction) even with one GPU")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if world_env is None and args.gpus > 1:
        # not under a launcher: become `torch.distributed.run --nproc-per-node N` ourselves (one process per GPU)
        import socket
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) visible: refusing to run fewer ranks "
                     f"(one RCCL rank per GPU)")
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import numpy as np
    import torch
    from flagger_amd import _native as N
    from flagger_amd import dist as fdist
    from flagger_amd import hmm, synth

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local_rank}, {torch.cuda.device_count()} visible): one RCCL rank per GPU")
    dist_path = world > 1 or args.dist_path
    if dist_path:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(local_rank)
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert tdist.get_world_size() == args.gpus

    # ---- workload: BASELINE.json configs[2] (= configs[3] when sharded over 8 GPUs) ----
    store = synth.config(args.config, scale=args.scale, overdispersion=args.overdispersion)
    base_store = store
    if args.scaling == "weak" and world > 1:      # one whole genome per rank: the chunk list of `world` genomes
        store = store.subset_chunks(list(range(store.n_chunks)) * world)
    K = hmm.getBestNumberOfCollapsedComps(store)
    alpha = synth.HIFI_ALPHA if args.config in (2, 6) else synth.ONT_R10_ALPHA
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, alpha)
    algo = N.HF_ALGO_SCAN if args.algo == "scan" else N.HF_ALGO_SEQ
    torch.cuda.set_device(local_rank)
    collective_used = {"name": args.collective if dist_path else "none"}

    def make_sharded(st, mdl, exchange=None):
        exchange = exchange or args.exchange
        if dist_path and collective_used["name"] == "native":
            uid = [hmm.comm_unique_id() if rank == 0 else None]
            tdist.broadcast_object_list(uid, src=0, device=torch.device("cuda", local_rank))
            sh, err = None, ""
            try:
                # (hf_multi_create_rank agrees on the outcome of every rank's set-up through the new communicator; what can fail
                # BEFORE a communicator exists — no visible device for this rank — was checked above, on every rank)
                sh = hmm.RankEMList(st, mdl, world, rank, local_rank, uid[0], True, 0.95, algo,
                                    exchange=N.HF_EXCHANGE_RANKS if exchange == "ranks" else N.HF_EXCHANGE_CHUNKS)
            except Exception as e:          # never seen; N > 1 cannot be tried on the 1-GPU development boxes
                err = repr(e)
            ok = torch.tensor([1 if sh is not None else 0], dtype=torch.int32, device="cuda")
            tdist.all_reduce(ok, op=tdist.ReduceOp.MIN)      # all ranks take the same path
            if int(ok.item()) == 1:
                return sh, sh.em
            if sh is not None:
                sh.close()
            collective_used["name"] = "torch (native set-up failed on some rank: %s)" % (err or "another rank")
        sh = fdist.make_sharded_hip(st, mdl, rank, world, local_rank, True, 0.95, algo, exchange=exchange)
        return sh, sh.local.em

    sharded, em = make_sharded(store, model)
    if not dist_path:                      # one GPU, no exchange: statistics by emission row (the library's default)
        em.set_stats_mode(N.HF_STATS_ROWS)
    em.set_profiling(True)                 # warm-up passes time every kernel to find the dominant one
    n_windows = store.n_windows

    sharded.force_collective = args.dist_path
    target = em if not dist_path else sharded   # single GPU: no exchange buffers in the path

    def step():
        if not dist_path:
            em.em_iterate(model, True, 1e-3)                      # E-step + decode + ordered reduce + M-step, one native call
        elif hasattr(target, "em_iterate"):
            target.em_iterate(model, True, 1e-3)                  # the same + the RCCL all-gather, one native call (hf_multi_em_iterate)
        else:
            hmm.EM_runOneIterationForList(target, model)          # E-step + decode + all-gather + ordered reduce
            hmm.HMM_estimateParameters(model, 1e-3)               # M-step on the host (replicated on every rank)
            hmm.HMM_resetEstimators(model)

    def barrier():
        if dist_path:
            tdist.barrier()
        torch.cuda.synchronize()

    dom = "k_seg_fb"                       # the dominant kernel unless the warm-up passes say otherwise
    for _ in range(args.warmup):
        step()
        kt = em.kernel_times()
        dom = max(kt, key=kt.get)
    stride = args.event_stride if args.event_stride > 0 else max(1, min(32, args.steps // 5))
    em.set_profiling([] if args.no_kernel_events else [dom])   # timed region: only the dominant kernel is bracketed by HIP events,
    em.set_profiling_stride(stride)                            # and only in every n-th step
    dom_ms, dom_samples = 0.0, []
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        if not args.no_kernel_events and i % stride == 0:      # this step carried the pair (hf_set_profiling_stride counts from 0): its duration
            dom_samples.append(em.kernel_times()[dom])
    barrier()
    dt = time.perf_counter() - t0
    ll = model.loglikelihood               # after the last TIMED step (the loops below go on iterating the same model)
    if os.environ.get("BENCH_DUMP_SAMPLES") and rank == 0:   # the sequence of samples on stderr (how the kernel settles after a cold start)
        print("[bench] %s samples (us): " % dom + " ".join("%.1f" % (v * 1e3) for v in dom_samples), file=sys.stderr)
    if dom_samples:                        # median: the first sample follows the barrier's idle gap and runs at a lower clock
        dom_ms = float(np.median(dom_samples))
    # what the event pairs of the timed region cost it: the same K steps once more, right behind it, with no event in the stream (the timed
    # region's sampled launches carry a completion signal of their own and the runtime serialises around it: ~12 us per sampled step)
    plain_ms = None
    if not args.no_kernel_events and not dist_path:
        em.set_profiling([])
        barrier()
        p0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        plain_ms = (time.perf_counter() - p0) / args.steps * 1e3
    # per-kernel breakdown from a few extra passes outside the timed region (every kernel bracketed)
    em.set_profiling_stride(1)
    em.set_profiling(True)
    ksum, extra = {}, min(max(args.steps, 1), 10)
    for _ in range(extra):
        step()
        for k, v in em.kernel_times().items():
            ksum[k] = ksum.get(k, 0.0) + v
    barrier()
    if dist_path:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- a REAL EM run (SURVEY 8d defines the metric as N x (I + 1) / t_EM of one): fresh model, fresh context, EM to convergence
    # (-n 100 -t 1e-3) + the final inference pass, wall-timed; no warm-up of its own.  Twice: in this (by now warm) process through the
    # same native calls as the timed region, and as a COLD process — the hmm_flagger command line on the same windows (.bin), which
    # times its own E-steps + M-steps (file reading, hf_create and output writing excluded, as the metric says) ----
    em_run = None
    if world == 1 and not dist_path and not args.no_em_run:
        try:
            rmodel = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, alpha)
            frac = 0.8 if args.config in (4, 5) else 0.95
            # hf_create (windows up, records, plan: what EM_construct + EM_renewParametersAndEstimatorsFromModel cost the reference per chunk and
            # iteration, hmm.c:253-298 — paid once here): `args.creates` fresh contexts one after another while the timed context is still alive
            # (one sample on a fresh box said 8.35 ms where the docs said 2.5-3.0); the line carries the median, min and max and the
            # library's own phases (hf_create_phases) of the median context; the last context runs the EM below
            creates, rem = [], None
            for ci in range(max(1, args.creates)):
                if rem is not None:
                    rem.close()
                c0 = time.perf_counter()
                rem = hmm.EMList(store, rmodel, True, frac, device=local_rank, algo=algo)
                creates.append(((time.perf_counter() - c0) * 1e3, rem.create_phases()))
            order = sorted(range(len(creates)), key=lambda i: creates[i][0])
            create_ms, create_phases = creates[order[len(order) // 2]]
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            passes, conv = 0, False
            while passes < 100 and not conv:
                conv = rem.em_iterate(rmodel, True, 1e-3)
                passes += 1
            rem.em_iterate(rmodel, False, 1e-3)                   # final inference, hmm_flagger.c:464
            passes += 1
            rdt = time.perf_counter() - r0
            rem.close()
            em_run = {"value": n_windows * passes / rdt, "unit": "windows/s", "passes": passes, "converged": bool(conv), "ms": rdt * 1e3,
                      "ms_per_pass": rdt / passes * 1e3, "vs_steady_state_step": (rdt / passes) / (dt / args.steps),
                      "hf_create_ms": create_ms, "hf_create_ms_min": creates[order[0]][0], "hf_create_ms_max": creates[order[-1]][0],
                      "hf_create_ms_all": [round(c[0], 3) for c in creates],
                      "hf_create_phases_ms": {k: round(v, 3) for k, v in create_phases.items()},
                      "hf_create_what": "%d fresh contexts, one after another, the timed context still alive; median / min / max of the wall time "
                                        "around EMList(...) and the library's phases of the median context" % len(creates),
                      "final_loglikelihood": rmodel.loglikelihood,
                      "what": "fresh context in this (warm) process: EM to convergence (100 iterations at most, tol 1e-3) + final pass, wall-timed, "
                              "hf_create not included (reported beside it)"}
            cli = os.path.join(ROOT, "flagger_amd", "csrc", "hmm_flagger")
            if os.path.exists(cli) and args.config in (2, 4, 6):
                import re
                import subprocess
                import tempfile
                with tempfile.TemporaryDirectory() as td:
                    store.write_bin(os.path.join(td, "w.bin"))
                    os.mkdir(os.path.join(td, "o"))
                    cmd = [cli, "-i", os.path.join(td, "w.bin"), "-n", "100", "-t", "1e-3", "-o", os.path.join(td, "o"), "--device", str(local_rank)]
                    cmd += (["-x", "ont-r10", "-A", os.path.join(ROOT, "tests", "golden", "alpha_ont_r10.tsv")] if args.config == 4 else
                            ["-W", str(store.window_len), "-A", os.path.join(ROOT, "tests", "golden", "alpha_hifi.tsv")])
                    best = None
                    for _ in range(3):                            # three cold processes: the median of their own EM+decode times
                        w0 = time.perf_counter()
                        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                        wall = time.perf_counter() - w0
                        m = re.search(r"EM\+decode: (\d+) passes over (\d+) windows in ([0-9.]+) s", r.stderr)
                        mw = re.search(r"output files took ([0-9.]+) s", r.stderr)      # the loop's wall time: log lines and per-iteration files included
                        if r.returncode == 0 and m:
                            best = (best or []) + [(float(m.group(3)), int(m.group(1)), wall, float(mw.group(1)) if mw else None)]
                    if best:
                        best.sort()
                        t_em, p_cli, wall, t_loop = best[len(best) // 2]
                        em_run["cli_cold_process"] = {"value": n_windows * p_cli / t_em, "unit": "windows/s", "passes": p_cli, "ms": t_em * 1e3,
                                                      "em_wall_ms": None if t_loop is None else t_loop * 1e3,   # (emWall next to emTime)
                                                      "ms_per_pass": t_em / p_cli * 1e3, "vs_steady_state_step": (t_em / p_cli) / (dt / args.steps),
                                                      "process_wall_ms": wall * 1e3, "runs": len(best),
                                                      "what": "`hmm_flagger -n 100 -t 1e-3` on the same windows (.bin), a fresh process each: its own "
                                                              "'EM+decode' line (E-steps + M-steps; loading, hf_create and output files excluded), median of the runs"}
        except Exception as e:                  # the headline number above stays valid
            em_run = {"error": repr(e)}

    # second leg at N > 1: the same exchange with one WHOLE genome per GPU (weak scaling), outside the timed region above
    weak = None
    if (world > 1 or (args.force_weak_leg and dist_path)) and args.scaling == "strong" and not args.no_weak_leg:
        try:
            wstore = base_store.subset_chunks(list(range(base_store.n_chunks)) * world)
            wmodel = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, wstore, alpha)
            wsh, _ = make_sharded(wstore, wmodel)

            def wstep():
                hmm.EM_runOneIterationForList(wsh, wmodel)
                hmm.HMM_estimateParameters(wmodel, 1e-3)
                hmm.HMM_resetEstimators(wmodel)
            for _ in range(max(args.warmup, 1)):
                wstep()
            barrier()
            w0 = time.perf_counter()
            for _ in range(args.steps):
                wstep()
            barrier()
            wt = torch.tensor([time.perf_counter() - w0], dtype=torch.float64, device="cuda")
            tdist.all_reduce(wt, op=tdist.ReduceOp.MAX)
            wdt = float(wt.item())
            weak = {"value": wstore.n_windows * args.steps / wdt, "unit": "windows/s", "ms_per_step": wdt / args.steps * 1e3,
                    "n_windows": wstore.n_windows, "windows_per_gpu": (wsh.n_local_windows if hasattr(wsh, "n_local_windows") else wsh.local_store.n_windows), "scaling": "weak",
                    "loglikelihood_after_last_step": wmodel.loglikelihood}
        except Exception as e:              # the headline number above stays valid
            weak = {"error": repr(e)}

    # how many ranks RCCL itself counts in the communicator of the timed region (ncclCommCount), not what was asked for
    rccl_ranks = None
    if dist_path and collective_used["name"] == "native":
        rccl_ranks = int(N.lib().hf_multi_comm_ranks(sharded._h))
        if rccl_ranks != world:
            sys.exit(f"bench.py: RCCL reports {rccl_ranks} ranks in the communicator, --gpus {world}: refusing to print a line")
    elif dist_path:
        rccl_ranks = tdist.get_world_size()

    # at N > 1 (or --dist-path): the OTHER exchange, timed the same way, and a proof that the sharded run is the one-GPU run:
    # a small copy of the workload through `--exchange chunks` on all ranks against a single context on rank 0
    other, invariance = None, None
    if dist_path and not args.no_second_exchange:
        try:
            oex = "chunks" if args.exchange == "ranks" else "ranks"
            omodel = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, alpha)
            osh, _ = make_sharded(store, omodel, oex)
            osh.force_collective = args.dist_path

            def ostep():
                hmm.EM_runOneIterationForList(osh, omodel)
                hmm.HMM_estimateParameters(omodel, 1e-3)
                hmm.HMM_resetEstimators(omodel)
            for _ in range(max(args.warmup, 1)):
                ostep()
            barrier()
            o0 = time.perf_counter()
            for _ in range(args.steps):
                ostep()
            barrier()
            ot = torch.tensor([time.perf_counter() - o0], dtype=torch.float64, device="cuda")
            tdist.all_reduce(ot, op=tdist.ReduceOp.MAX)
            odt = float(ot.item())
            other = {"exchange": oex, "value": n_windows * args.steps / odt, "unit": "windows/s", "ms_per_step": odt / args.steps * 1e3,
                     "loglikelihood_after_last_step": omodel.loglikelihood}
            getattr(osh, 'close', lambda: None)()
        except Exception as e:
            other = {"error": repr(e)}
        try:
            sscale = min(args.scale, 0.05)
            sstore = synth.config(args.config, scale=sscale, overdispersion=args.overdispersion)
            iters = 5

            def sharded_run(exchange):
                """`iters` EM iterations + the final inference pass (hmm_flagger.c:464) over all ranks: (log-likelihoods, labels of the
                whole input on every rank)."""
                import numpy as np
                smodel = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, sstore, alpha)
                ssh, _ = make_sharded(sstore, smodel, exchange)
                ssh.force_collective = args.dist_path
                lls = []
                for _ in range(iters):
                    hmm.EM_runOneIterationForList(ssh, smodel)
                    lls.append(smodel.loglikelihood)
                    hmm.HMM_estimateParameters(smodel, 1e-3)
                    hmm.HMM_resetEstimators(smodel)
                hmm.EM_runOneIterationForList(ssh, smodel)
                lls.append(smodel.loglikelihood)
                mine = ssh.local_labels() if hasattr(ssh, "local_labels") else ssh.local.labels()
                first = ssh.first_window if hasattr(ssh, "first_window") else int(sstore.chunk_off[ssh.bounds[rank]])
                parts = [None] * world
                tdist.all_gather_object(parts, (first, mine.tobytes()))
                getattr(ssh, 'close', lambda: None)()
                lab = np.full(sstore.n_windows, -1, dtype=np.int8)
                for f0, b in parts:
                    a = np.frombuffer(b, dtype=np.int8)
                    lab[f0:f0 + a.size] = a
                return lls, lab

            def one_context_run(stats_mode):
                rmodel = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, sstore, alpha)
                rem = hmm.EMList(sstore, rmodel, True, 0.95, device=local_rank, algo=algo)
                if stats_mode is not None:
                    rem.set_stats_mode(stats_mode)
                rlls = []
                for _ in range(iters):
                    hmm.EM_runOneIterationForList(rem, rmodel)
                    rlls.append(rmodel.loglikelihood)
                    hmm.HMM_estimateParameters(rmodel, 1e-3)
                    hmm.HMM_resetEstimators(rmodel)
                hmm.EM_runOneIterationForList(rem, rmodel)
                rlls.append(rmodel.loglikelihood)
                lab = rem.labels()
                rem.close()
                return rlls, lab

            lls, lab_n = sharded_run("chunks")
            lls_r, lab_r = sharded_run("ranks")
            if rank == 0:
                try:                                   # rank-0-only work: whatever happens here, every rank reaches the barrier below
                    rlls, lab_1 = one_context_run(N.HF_STATS_CHUNKS)
                    invariance = {"what": f"{iters} EM iterations + final pass, exchange chunks over {world} rank(s) vs one context (per-chunk statistics) on rank 0",
                                  "scale": sscale, "n_windows": sstore.n_windows, "loglikelihoods_bit_identical": lls == rlls,
                                  "labels_identical": bool((lab_n == lab_1).all()), "label_mismatches": int((lab_n != lab_1).sum()),
                                  "final_loglikelihood": lls[-1]}
                    # the headline exchange (`ranks`: every rank sums its shard by emission row first) is equal to the one-GPU run up
                    # to the rounding of a different order of additions: how far, and whether a single label moves
                    qlls, lab_q = one_context_run(None)
                    invariance["exchange_ranks_vs_one_context"] = {
                        "label_mismatches": int((lab_r != lab_q).sum()),
                        "loglikelihood_max_rel_diff": max(abs(a - b) / abs(b) for a, b in zip(lls_r, qlls))}
                except Exception as e:
                    invariance = {"error": repr(e)}
            barrier()
        except Exception as e:
            invariance = {"error": repr(e)}

    if rank == 0:
        kavg = {k: v / extra for k, v in ksum.items() if v > 0}
        if args.no_kernel_events:
            dom_ms = kavg.get(dom, 0.0)
        local_windows = sharded.n_local_windows if hasattr(sharded, "n_local_windows") else sharded.local_store.n_windows
        # a context past the Infinity Cache runs a pass in sub-passes (hf_sub_passes): the timed launch of the dominant kernel is the FIRST
        # sub-pass's, so the units of one launch are that sub-pass's windows (one sub-pass — every BASELINE configuration — all of them)
        sub_passes = getattr(em, "sub_passes", 1) or 1
        if dom == "k_seg_fb" and sub_passes > 1:
            local_windows = em.sub_pass_windows(0)
        achieved = ALGO_BYTES_PER_WINDOW * local_windows / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/pmc_traffic.json,
        # made by profiles/pmc_summary.py on this same command; FETCH_SIZE x2 on gfx950): full workload, 1 GPU only
        # NOT measured by this run: the PMC passes need rocprofv3 around the process, so the figure is read from the
        # committed summary of the same command (profiles/collect.sh -> profiles/pmc_traffic.json, FETCH_SIZE x2 on
        # gfx950) and labelled with where it came from; null for any other workload
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pmc_tag = None
        same_workload = world == 1 and args.scale == 1.0 and args.config == 2
        if os.path.exists(pmc) and same_workload:
            try:
                doc = json.load(open(pmc))
                pmc_tag = doc.get("_meta", {}).get("tag")
                for k, v in doc.items():
                    if not k.startswith("_") and k.split("<")[0] == dom:
                        traffic = v["hbm_bytes_per_launch"]
                import hashlib
                traffic_src = "profiles/pmc_traffic.json sha256:" + hashlib.sha256(open(pmc, "rb").read()).hexdigest()[:12] + \
                              (f" = profiles/{pmc_tag}_pmc_traffic.json" if pmc_tag else "") + \
                              " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not this run)"
            except Exception:
                traffic = None

        def committed_counters(suffix):
            """Per-launch counter averages of the dominant kernel from the committed SQ pass of the SAME snapshot as the traffic
            figure (profiles/<tag>_<suffix>.json, the tag is in pmc_traffic.json's _meta): (values, source) or (None, None)."""
            if not (pmc_tag and same_workload):
                return None, None
            f = os.path.join(ROOT, "profiles", f"{pmc_tag}_{suffix}.json")
            try:
                for k, v in json.load(open(f)).items():
                    if k.split("<")[0] == dom:
                        return v, f"profiles/{pmc_tag}_{suffix}.json (rocprofv3 --pmc SQ pass of this command, not this run)"
            except Exception:
                pass
            return None, None
        # what actually bounds the kernel: (1) the share of a SIMD's time in which its VALU issues, waves-per-SIMD x SQ_ACTIVE_INST_VALU /
        # SQ_WAVE_CYCLES; (2) the fp64 rate: flops from the SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 instruction counters (wavefront
        # instructions x 64 lanes, FMA = 2 flops) / this run's kernel time, against the fp64 vector peak
        valu_busy, valu_src = None, None
        sqb, src_b = committed_counters("pmc_sq_b")
        if sqb and sqb.get("SQ_WAVE_CYCLES"):
            valu_busy, valu_src = 3.0 * sqb["SQ_ACTIVE_INST_VALU"] / sqb["SQ_WAVE_CYCLES"], src_b
        fp64 = None
        f64c, src_f = committed_counters("pmc_fp64")
        if f64c and dom_ms > 0:
            fma, mul, add, trans = (f64c.get("SQ_INSTS_VALU_" + n + "_F64", 0.0) for n in ("FMA", "MUL", "ADD", "TRANS"))
            flops = 64.0 * (2.0 * fma + mul + add + trans)
            ach = flops / (dom_ms * 1e-3) / 1e12
            fp64 = {"bound": "fp64-valu", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS,
                    "kernel": dom, "flops_per_launch": flops,
                    "wave_instructions_per_launch": {"fma_f64": fma, "mul_f64": mul, "add_f64": add, "trans_f64": trans},
                    "flops_source": src_f + ": (2 x SQ_INSTS_VALU_FMA_F64 + MUL_F64 + ADD_F64 + TRANS_F64) x 64 lanes per launch; "
                                    "divided by THIS run's median kernel time",
                    "note": "fp64 comparisons / max / ldexp / frexp / moves are VALU work too but no flops: the kernel's VALU issue share is "
                            "valu_busy_frac_of_simd_time in `roofline`"}
        out = {
            "metric": "coverage windows/sec through EM+decode; achieved HBM GB/s vs roofline",
            "value": n_windows * args.steps / dt, "unit": "windows/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "ms_per_step_without_kernel_events": plain_ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: synthetic 2x3.03 Gb diploid HiFi-like coverage, "
                                    "4 kb windows, 20 Mb chunks, trunc_exp_gaussian, HiFi v1.1.0 alpha, full EM step "
                                    "(E-step+decode on GPU, M-step on host)" if args.config == 2 else
                                    "BASELINE.json configs[4]: synthetic 2x3.03 Gb diploid, ONT-R10 preset (8 kb windows), 7 bias "
                                    "regions with their own emission parameters, ONT-R10 v1.1.0 alpha, full EM step" if args.config == 4 else
                                    "configs[2] with over-dispersed coverage (not a BASELINE config): negative binomial, variance = %g x mean" % args.overdispersion if args.config == 6 else
                                    "worst case for the emission tables (not a BASELINE config): configs[4] with coverage spread over "
                                    "0..250, ~1 window per (region, x, x_prev) key, K = 10")
                                   + ("" if args.scale == 1.0 else f" [scale {args.scale}]")
                                   + (f" [weak scaling: {world} such genomes, one per GPU]" if args.scaling == "weak" and world > 1 else ""),
                       "n_windows": n_windows, "n_chunks": store.n_chunks, "collapsed_comps": K,
                       "algo": args.algo,
                       "statistics": "per chunk, ordered reduction" if em.stats_mode == N.HF_STATS_CHUNKS else "by emission row",
                       "parallelism": (f"chunks sharded over {world} GPU(s), " + ("no exchange" if not dist_path else
                                       "all-gather of one statistics vector per rank" if args.exchange == "ranks" else
                                       "all-gather of per-chunk statistics")
                                       + ("" if not dist_path else ", one native call per pass (pass + RCCL all-gather + ordered reduction on one stream)"
                                          if collective_used["name"] == "native" else ", exchange through torch.distributed: " + collective_used["name"]))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": dom,
                         "kernel_ms_timed": dom_ms, "n_samples": len(dom_samples),
                         "kernel_ms_samples": {"median": dom_ms, "mean": float(np.mean(dom_samples)) if dom_samples else None,
                                               "min": min(dom_samples) if dom_samples else None, "max": max(dom_samples) if dom_samples else None},
                         "kernel_events": (0 if args.no_kernel_events else
                                           f"HIP event pair around {dom} in every {stride}. step of the timed region; kernel_ms_timed = median of n_samples"),
                         "limiter": "no single resource: fp64 VALU issue + dependent chains in the scans (VALU busy 60 % of SIMD time, roofline_fp64) and the CUs' "
                                    "L1 -> LDS path in the three walks over the rows of A (53 % of its 64 B/clk over the whole kernel, at its rate during the "
                                    "walks: profiles/r06_pmc_ta_*.json, DESIGN.md section 5), not HBM bytes",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_WINDOW * local_windows,
                         "kernel_ms_all": kavg, "algorithmic_bytes_per_window": ALGO_BYTES_PER_WINDOW,
                         "windows_per_launch": local_windows, "sub_passes": sub_passes, "cached_row_blocks": getattr(em, "seg_cached_steps", None),
                         "valu_busy_frac_of_simd_time": valu_busy, "valu_busy_source": valu_src,
                         "note": "`bound` names the ceiling this object is priced against (bytes: SURVEY 8d classifies the path as a streaming scan); "
                                 "what limits the kernel is in `limiter` and `roofline_fp64` (DESIGN.md section 5)"},
            "loglikelihood_after_last_step": ll,
        }
        if fp64 is not None:
            out["roofline_fp64"] = fp64
        if em_run is not None:
            out["em_run"] = em_run
        if weak is not None:
            out["weak_scaling"] = weak
        if other is not None:
            out["other_exchange"] = other
        if invariance is not None:
            out["n_invariance"] = invariance
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(store, K, alpha, effective_cores())
        print(json.dumps(out))
    if dist_path:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
