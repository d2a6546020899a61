import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flagger_amd import _native as N
k = sys.argv[1]
N.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"lib{k}.so")
from flagger_amd import hmm, synth
store = synth.config(2)
model = hmm.createModel(0, 6, store, synth.HIFI_ALPHA)
em = hmm.EMList(store, model)
nt = sum(-(-int(t) // 256) for t in (store.chunk_off[1:] - store.chunk_off[:-1]))
for it in range(3):
    hmm.EM_runForwardForList(em, model)
print("phase", k, "avg cycles/tile", model.loglikelihood / nt)
