#!/bin/bash
# soak of the round-6 build (wide hand-off accesses: the 16-byte sc0 sc1 stores and loads are new): many consecutive EM steps — no hand-off wait may be given up (a "falls back" line on stderr) — at full size, with
# cached row blocks (scale 0.125), in sub-passes (scale 2, forced 3 at scale 1) and on the other configurations
set -u
cd "$(dirname "$0")/../.."
one() { python bench.py "$@" --warmup 10 --no-cpu-baseline --no-em-run --no-kernel-events 2> /tmp/soak.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*: ms_per_step', round(d['ms_per_step'],4), 'LL', d['loglikelihood_after_last_step'])"; echo "   hand-off time-outs: $(grep -c 'falls back' /tmp/soak.err)"; }
one --config 2 --steps 300000
one --config 2 --scale 0.125 --steps 100000
one --config 2 --scale 2 --steps 20000
HF_SUBPASSES=3 one --config 2 --steps 20000
one --config 4 --steps 20000
one --config 5 --steps 20000
one --config 6 --steps 20000
HF_SEG_XCD=1 one --config 2 --steps 20000
