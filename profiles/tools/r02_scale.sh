#!/bin/bash
# size sweep of the bench workload (VERDICT r01 item 5 i): configs[2] with every contig length x SCALE — per-kernel time, per
# window time and roofline fraction against the problem size.  bash profiles/tools/r02_scale.sh <tag>
set -u
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
for s in ${SCALES:-0.25 0.5 1 2 4 8}; do
  python bench.py --steps 30 --warmup 3 --scale $s --no-cpu-baseline --no-weak-leg > gpurun_out/$TAG/scale_$s.json 2> gpurun_out/$TAG/scale_$s.err || tail -3 gpurun_out/$TAG/scale_$s.err
done
python - <<PY
import json
out = []
for s in "${SCALES:-0.25 0.5 1 2 4 8}".split():
    try:
        d = json.load(open("gpurun_out/$TAG/scale_%s.json" % s))
    except Exception as e:
        print(s, "failed", e); continue
    k = d["roofline"]["kernel_ms_all"]
    row = {"scale": float(s), "windows": d["config"]["n_windows"], "chunks": d["config"].get("n_chunks"), "ms_per_step": d["ms_per_step"],
           "G_windows_per_s": d["value"] / 1e9, "ns_per_window": d["ms_per_step"] * 1e6 / d["config"]["n_windows"],
           "roofline_frac": d["roofline"]["frac"], "achieved_GBps": d["roofline"]["achieved"],
           "kernel_us": {n: round(v * 1e3, 1) for n, v in k.items()}}
    out.append(row)
    print(row)
json.dump(out, open("gpurun_out/$TAG/scale_sweep.json", "w"), indent=1)
PY
