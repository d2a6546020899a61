python - <<'PY' 2>&1 | grep -E "hf_create|total"
import os, time, sys
os.environ["HF_HOST_TRACE"]="2"
sys.path.insert(0,".")
from flagger_amd import hmm, synth
store=synth.config(2); model=hmm.createModel(0, 6, store, synth.HIFI_ALPHA)
t=time.time(); em=hmm.EMList(store, model); print("total hf_create %.1f ms" % ((time.time()-t)*1e3))
t=time.time(); em2=hmm.EMList(store, model); print("total hf_create (second) %.1f ms" % ((time.time()-t)*1e3))
PY
