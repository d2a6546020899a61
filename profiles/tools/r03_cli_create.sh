T=$(mktemp -d /tmp/cliwall.XXXX)
python - <<PY
import sys
sys.path.insert(0, ".")
from flagger_amd import synth
synth.config(2).write_bin("$T/cfg2.bin")
PY
mkdir -p $T/o
HF_HOST_TRACE=1 HF_CLI_TIMING=1 flagger_amd/csrc/hmm_flagger -i $T/cfg2.bin -n 100 -W 4000 -A tests/golden/alpha_hifi.tsv -o $T/o 2>&1 | grep -E "hf_create\]|phase\]"
