import torch, os, sys
sys.path.insert(0, os.getcwd())
import render_in_between_amd as rib
from render_in_between_amd import synth
from oracle import generator_ref
cfg = rib.hsm_gen_config(); spec = rib.GenSpec.from_cfg(cfg); sd = synth.make_state_dict(spec, 1)
G = rib.Generator(cfg, use_tuning=False).eval(); G.load_state_dict(sd)
label, fake, prev = synth.make_inputs(spec, 1, 128, 128, 1)
img, mask = G(label, None, fake, prev); torch.cuda.synchronize()
taps = G.read_taps(1, 128, 128); ot = {}
generator_ref.RefGenerator(spec, sd)(label, None, fake, prev, taps=ot)
e = (taps["down_0.ys0"] - ot["down_0.ys0"]).abs()[0]
print("per-channel max err", [round(float(e[c].max()), 4) for c in range(e.shape[0])])
bad = (e.max(0).values > 1e-3)
print("bad pixel fraction", float(bad.float().mean()))
rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
print("bad rows", rows[:40]); print("bad cols", cols[:40])
x = taps["down_first"]; xo = ot["down_first"]
print("x err", float((x - xo).abs().max()))
