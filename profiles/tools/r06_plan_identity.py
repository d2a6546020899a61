#!/usr/bin/env python
"""hf_create split into steps must not change a plan.  For seeded random shapes (the generator of fuzz_modes.py: 1-5
contigs, 1-4 regions, three model types, forced plan layouts / sub-passes / cached row blocks) one full pass in both statistics modes
+ a forward-only pass; prints one digest per seed of everything a plan determines to the last bit: the statistics vectors of both modes,
the labels, the forward / backward / scale arrays of a window range, segment launches, sub-passes, cached steps.  Run once per library
(HF_LIBRARY_VARIANT=precreate: the round's library before the split) and diff the two outputs.
   python profiles/tools/r06_plan_identity.py <first seed> <n seeds>"""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from flagger_amd import hmm, synth, _native as N  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
PLANS = ["", "compact", "compact,bpw=3", "padded,bpw=2", "compact,bpw=8"]
for seed in range(first, first + count):
    rng = np.random.default_rng(5000 + seed)
    plan = PLANS[seed % len(PLANS)]
    if plan: os.environ["HF_STATS_PLAN"] = plan
    else: os.environ.pop("HF_STATS_PLAN", None)
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
    h = hashlib.sha256()
    try:
        em.launch(model); a = em.finish(); h.update(a.tobytes()); h.update(em.labels().tobytes())
        n = min(store.n_windows, 4000)
        for arr in em.forward_backward(max(0, store.n_windows - n), n): h.update(np.ascontiguousarray(arr).tobytes())
        em.launch(model, N.HF_MODE_FORWARD_ONLY); h.update(em.finish()[:1].tobytes())
        em.set_stats_mode(N.HF_STATS_CHUNKS)
        em.launch(model); b = em.finish(); h.update(b.tobytes()); h.update(em.labels().tobytes())
        print(seed, store.n_windows, store.n_chunks, R, K, mt, em.seg_launches, em.sub_passes, em.seg_cached_steps, h.hexdigest()[:24])
    finally:
        em.close()
