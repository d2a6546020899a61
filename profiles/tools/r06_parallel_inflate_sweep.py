#!/usr/bin/env python
"""Round 6: 60 random bam2cov-like tracks (1-6 contigs, runs of 5-600 bases, written Huffman-only like the reference's), each loaded by the one decoder
and by 2-8 decoders inside the stream with pieces of 8 KB .. 200 KB and probes of 1 B .. 100 KB: the windows must be identical.  (CPU only.)"""
import sys, os, hashlib, subprocess, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flagger_amd import synth
code = ("import sys, hashlib\nsys.path.insert(0, %r)\nfrom flagger_amd import io as fio\n"
        "st = fio.Table(sys.argv[1], 1000000, 4000).store()\nh = hashlib.sha256()\n"
        "for f in ('cov','mapq','clip','annot','chunk_off','chunk_s','chunk_e'):\n    h.update(getattr(st, f).tobytes())\nprint(st.n_windows, h.hexdigest())\n") % os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rnd = random.Random(1)
bad = 0; rounds_total = 0
for seed in range(60):
    L = [rnd.randint(20_000, 3_000_000) for _ in range(rnd.randint(1, 6))]
    mn = rnd.choice([5, 20, 50, 200]); mx = mn + rnd.choice([10, 100, 400])
    p = '/tmp/sweep.cov.gz'
    synth.write_cov_dense(p, L, seed=seed, min_run=mn, max_run=mx)
    outs = []
    for env in ({"HF_IO_PARALLEL": "0"}, {"HF_IO_PARALLEL": str(rnd.choice([2,3,4,8])), "HF_IO_PARALLEL_MIN": "0", "HF_IO_PIECE": str(rnd.choice([8192, 20000, 50000, 200000])), "HF_IO_PROBE": str(rnd.choice([1, 5000, 100000]))}):
        r = subprocess.run([sys.executable, "-c", code, p], capture_output=True, text=True, env=dict(os.environ, HF_IO_TRACE="1", **env))
        outs.append((r.stdout.strip(), [l for l in r.stderr.splitlines() if l.startswith('[hfio]')][-1].split(';')[-1] if r.stderr else ''))
    if outs[0][0] != outs[1][0] or not outs[0][0]:
        bad += 1; print("MISMATCH seed", seed, outs)
    rounds_total += int(outs[1][1].split()[0]) if outs[1][1] else 0
print("60 files: mismatches", bad, "rounds of parallel decoders in total", rounds_total)
