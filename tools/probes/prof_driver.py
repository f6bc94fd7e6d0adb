import cProfile, pstats, sys, os, tempfile, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import render_in_between_amd as rib
from render_in_between_amd import synth, evaluator as ev
import driver_bench as db
H = W = 512
cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=H, model_width=W, gauss_sigma=5, skeleton_thres=0.001, foot_thres=0.001)
spec = rib.GenSpec.from_cfg(cfg.gen)
G = rib.Generator(cfg.gen).eval(); G.load_state_dict(synth.make_state_dict(spec, 0, power_iters=3))
with tempfile.TemporaryDirectory() as root:
    n = db.write_clip(root, 5, 32, H, W)
    E = ev.Evaluator(cfg)          # round 4 defaults: batch by size, chunk 8, 2 lanes, worker processes
    E.evaluate_from_folder(G, *[os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")], os.path.join(root, "ow"))
    dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
    E.evaluate_from_folder(G, *dirs, os.path.join(root, "o0"))
    pr = cProfile.Profile(); pr.enable()
    E.evaluate_from_folder(G, *dirs, os.path.join(root, "o1"))
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(36)
