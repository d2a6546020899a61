#!/bin/bash
# loader A/B on the GPU box's host: own inflater (default) against zlib's gzread (HF_IO_ZLIB=1), configs[2] as .cov.gz and as .cov
T=$(mktemp -d /tmp/ldab.XXXX)
python - <<PY
import sys
sys.path.insert(0, ".")
from flagger_amd import synth
st = synth.config(2)
st.write_cov("$T/c.cov.gz"); st.write_cov("$T/c.cov")
PY
cat > $T/ld.cpp <<'CPP'
#include "hmm_flagger_io.h"
#include <cstdio>
#include <chrono>
int main(int argc, char** argv) {
    for (int rep = 0; rep < 5; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        hfio_table* t = hfio_load(argv[1], 20000000, 4000);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf(" %.3f", dt); if (t) hfio_destroy(t); else printf("(failed)");
    }
    printf(" s\n");
}
CPP
g++ -O2 -I include -o $T/ld $T/ld.cpp -L flagger_amd/csrc -lhmmflagger_hip -Wl,-rpath,$PWD/flagger_amd/csrc
for f in c.cov.gz c.cov; do for z in 0 1; do echo -n "$f HF_IO_ZLIB=$z:"; HF_IO_ZLIB=$z HF_IO_TRACE=1 $T/ld $T/$f 2>&1 | tr '\n' ' '; echo; done; done
