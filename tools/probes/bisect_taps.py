import torch, json, os, sys
sys.path.insert(0, os.getcwd())
import render_in_between_amd as rib
from render_in_between_amd import synth
from oracle import generator_ref
cfg = rib.hsm_gen_config(); spec = rib.GenSpec.from_cfg(cfg); sd = synth.make_state_dict(spec, 1)
G = rib.Generator(cfg, use_tuning=(os.environ.get("NOTUNE") is None)).eval(); G.load_state_dict(sd)
label, fake, prev = synth.make_inputs(spec, 1, 128, 128, 1)
img, mask = G(label, None, fake, prev); torch.cuda.synchronize()
taps = G.read_taps(1, 128, 128); ot = {}
oi, om = generator_ref.RefGenerator(spec, sd)(label, None, fake, prev, taps=ot)
bad = [(k, round(float((v - ot[k]).abs().max()) / max(1.0, float(ot[k].abs().max())), 5)) for k, v in taps.items()]
bad = [b for b in bad if b[1] > 1e-4]
print(os.environ.get("TAG"), "first bad:", bad[:3], "img", float((img.cpu()-oi).abs().max()))
