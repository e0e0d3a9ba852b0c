import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench
from whisper_ipa_amd.whisper import Whisper
dims, W = bench.synthetic_weights_small(0, "small")
model = Whisper(dims, dtype=torch.bfloat16); model.load_weights(W); del W
audio = torch.from_numpy(bench.synthetic_audio(0, 64)).cuda()
setup = bench.decode_setup()
model.packed(); torch.cuda.synchronize()
ref = bench.one_pass(model, [audio], setup)
for trial in range(3):
    hs = [bench.pass_launch(model, [audio], setup, p) for p in range(4)]
    outs = [bench.pass_collect(h) for h in hs]
    for i, o in enumerate(outs):
        bad = np.argwhere(o != ref)
        print("trial", trial, "stream", i, "mismatches", len(bad), "first", bad[:3].tolist() if len(bad) else None)
one = bench.one_pass(model, [audio], setup)
print("single again equal:", (one == ref).all())
