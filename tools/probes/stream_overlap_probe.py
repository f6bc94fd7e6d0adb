import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import render_in_between_amd as rib
from render_in_between_amd import synth
cfg = rib.hsm_gen_config(); spec = rib.GenSpec.from_cfg(cfg)
G0 = rib.Generator(cfg).eval(); G0.load_state_dict(synth.make_state_dict(spec, 0))
label, fake, prev = [t.cuda() for t in synth.make_inputs(spec, 1, 512, 512, 0)]
def run(lanes, steps=60):
    for i in range(6):
        g, st = lanes[i % len(lanes)]
        with torch.cuda.stream(st): g(label, None, fake, prev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps):
        g, st = lanes[i % len(lanes)]
        with torch.cuda.stream(st): g(label, None, fake, prev)
    torch.cuda.synchronize(); return steps / (time.perf_counter() - t)
gens = [G0] + [G0.clone() for _ in range(3)]
d = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(6)]
print("default only            %.1f fps" % run([(gens[0], d)]))
print("pool0 only              %.1f fps" % run([(gens[0], pool[0])]))
print("default + pool0         %.1f fps" % run([(gens[0], d), (gens[1], pool[0])]))
print("pool0 + pool1           %.1f fps" % run([(gens[0], pool[0]), (gens[1], pool[1])]))
print("pool0 + pool2           %.1f fps" % run([(gens[0], pool[0]), (gens[1], pool[2])]))
print("pool1 + pool3           %.1f fps" % run([(gens[0], pool[1]), (gens[1], pool[3])]))
print("default + pool1         %.1f fps" % run([(gens[0], d), (gens[1], pool[1])]))
print("pool0+1+2               %.1f fps" % run([(gens[0], pool[0]), (gens[1], pool[1]), (gens[2], pool[2])]))
print("pool0+1+2+3             %.1f fps" % run([(gens[i], pool[i]) for i in range(4)]))
