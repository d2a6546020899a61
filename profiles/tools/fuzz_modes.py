#!/usr/bin/env python
"""One-off fuzz: seeded random input shapes (as tests/test_estep_gpu.py::test_random_inputs_...) for every model type; for each,
one full pass in both statistics modes compared with each other (1e-11) and with the oracle (statistics 1e-9, labels exact).
  python profiles/tools/fuzz_modes.py <first seed> <n seeds>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from flagger_amd import hmm, synth, _native as N  # noqa: E402
from oracle_py import Oracle  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
PLANS = ["", "compact", "compact,bpw=3", "padded,bpw=2", "compact,bpw=8"]     # HF_STATS_PLAN of the seed (hf_create reads it)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    plan = PLANS[seed % len(PLANS)]
    if plan: os.environ["HF_STATS_PLAN"] = plan
    else: os.environ.pop("HF_STATS_PLAN", None)
    # round 5: forced sub-passes (hf_sub_passes) and cached row blocks (hf_seg_cached_steps) as further axes of the seed
    sp, ncs = [None, "1", "2", "3", "5"][(seed // 5) % 5], [None, "0", "2", "8"][(seed // 3) % 4]
    for k, v in (("HF_SUBPASSES", sp), ("HF_SEG_CACHED_STEPS", ncs)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    window_len = int(rng.choice([500, 1000, 4000]))
    chunk_len = int(rng.choice([20, 77, 300])) * window_len
    lengths = [int(rng.integers(2, 3000)) * window_len + int(rng.integers(0, window_len)) for _ in range(int(rng.integers(1, 6)))]
    R = int(rng.integers(1, 5))
    store = synth.synthesize(lengths, window_len, chunk_len, [int(rng.integers(8, 40)) for _ in range(R)], seed=seed,
                             avg_alignment_len=int(rng.choice([0, 300, 15_000, 3_000_000])), region_run_bases=(3 * window_len, 200 * window_len))
    clip = np.asarray(store.clip).copy(); hit = rng.random(clip.size) < 0.03
    clip[hit] = (np.asarray(store.cov)[hit] * 2 + 1).astype(clip.dtype); store.clip = clip
    K = int(rng.integers(2, 10))
    mt = [hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.MODEL_GAUSSIAN, hmm.MODEL_NEGATIVE_BINOMIAL][seed % 3]
    alpha = np.zeros((4, 4)) if mt == hmm.MODEL_NEGATIVE_BINOMIAL else [synth.HIFI_ALPHA, synth.ONT_R10_ALPHA, np.zeros((4, 4))][int(rng.integers(0, 3))]
    adjust = bool(rng.integers(0, 2))
    model = hmm.createModel(mt, K, store, alpha)
    em = hmm.EMList(store, model, adjust, 0.9)
    orc = Oracle(store, mt, K, alpha, 0.25, 0.75, adjust, 0.9, threads=8)
    try:
        em.launch(model); a = em.finish(); lab_a = em.labels(); mode_a = em.stats_mode
        em.set_stats_mode(N.HF_STATS_CHUNKS)
        em.launch(model); b = em.finish(); lab_b = em.labels()
        assert orc.run_iteration() == 0
        ref = orc.stats_vector(model.maxNumberOfComps); olab = orc.labels()
        sc = np.maximum(np.abs(b), 1e-9 * np.abs(b).max()); scr = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
        ok = (a[0] == b[0] and np.array_equal(lab_a, lab_b) and np.all(np.abs(a - b) <= 1e-11 * sc) and np.array_equal(lab_a, olab)
              and np.all(np.abs(a - ref) <= 1e-9 * scr) and np.all(np.abs(b - ref) <= 1e-9 * scr))
        if not ok:
            bad += 1
            print("seed", seed, "model", mt, "mode", mode_a, "windows", store.n_windows, "R", R, "K", K, "FAILED:",
                  "rows-chunks", float(np.max(np.abs(a - b) / sc)), "rows-oracle", float(np.max(np.abs(a - ref) / scr)),
                  "labels", int(np.count_nonzero(lab_a != olab)))
    finally:
        em.close(); orc.close()
print("seeds", first, "..", first + count - 1, "failures", bad)
