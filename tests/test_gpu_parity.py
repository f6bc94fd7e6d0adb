"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called
through the C ABI via the drop-in Generator class, against the oracle on the
same seeded inputs and against the committed reference fixtures.

Tolerance: the north star asks for <= 1e-3 max-abs in fp32; the HIP path and
the oracle are both fp32 with different summation orders, so the tests hold it
to 2e-4 (outputs are tanh/sigmoid bounded, O(1))."""
import json
import os

import numpy as np
import pytest
import torch

import render_in_between_amd as rib
from render_in_between_amd import synth

pytestmark = pytest.mark.gpu

TOL = 2e-4
NORTH_STAR_TOL = 1e-3
MID_CFG = dict(num_filters=16, max_num_filters=64,
               mask=dict(num_filters=32, max_num_filters=64),
               embed=dict(num_filters=32, max_num_filters=64))


def _cfg(name):
    return rib.hsm_gen_config(**MID_CFG) if name.startswith("mid") else rib.hsm_gen_config()


_cache = {}


def build(cfgname, seed):
    key = (cfgname, seed)
    if key not in _cache:
        _cache.clear()
        cfg = _cfg(cfgname)
        spec = rib.GenSpec.from_cfg(cfg)
        sd = synth.make_state_dict(spec, seed)
        G = rib.Generator(cfg).eval()
        G.load_state_dict(sd)
        _cache[key] = (spec, sd, G)
    return _cache[key]


def oracle(spec, sd):
    from oracle import generator_ref
    return generator_ref.RefGenerator(spec, sd)


def test_native_library_is_the_loaded_path():
    from render_in_between_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    with open("/proc/self/maps") as f:
        build("mid", 7)
        assert "librib.so" in f.read()


def test_layer_taps_match_oracle_mid64():
    """Every materialised intermediate of the HIP path vs the oracle: localises
    a wrong kernel variant to the first diverging layer."""
    spec, sd, G = build("mid", 7)
    label, fake, prev = synth.make_inputs(spec, 1, 64, 64, 7)
    img0, mask0 = [t.clone() for t in G(label, None, fake, prev)]     # production plan: buffers reused
    from render_in_between_amd import _native
    with pytest.raises(_native.RibError, match="rib_set_debug_taps"):
        G.read_taps(1, 64, 64)
    G.enable_taps()
    img, mask = G(label, None, fake, prev)
    torch.cuda.synchronize()
    assert torch.equal(img, img0) and torch.equal(mask, mask0)        # the workspace layout does not change a bit
    taps = G.read_taps(1, 64, 64)
    G.enable_taps(False)
    otaps = {}
    oimg, omask = oracle(spec, sd)(label, None, fake, prev, taps=otaps)
    report = {}
    for k, v in taps.items():
        assert k in otaps, k
        ref = otaps[k]
        report[k] = float((v - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    report["img"] = float((img.cpu() - oimg).abs().max())
    report["mask"] = float((mask.cpu() - omask).abs().max())
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/taps_mid64.json", "w") as f:
        json.dump(report, f, indent=1)
    bad = {k: v for k, v in report.items() if not v <= TOL}
    assert not bad, "first diverging taps: %s" % list(bad.items())[:6]


@pytest.mark.parametrize("name", ["mid_64", "full_64", "full_128", "full_b2_64", "full_noise_128",
                                  "full_256", "full_320x480", "full_512", "full_b3_96x160", "full_1024"])
def test_outputs_match_reference_fixtures(name, golden_dir, golden_report):
    """Against the outputs of the imported reference generator itself (round 6 added a batch-3 non-square case and the
    1024x1024 frame of BASELINE configs[4], which had only been compared with the oracle)."""
    rep = golden_report[name]
    spec, sd, G = build("mid" if name.startswith("mid") else "full", rep["seed"])
    label, fake, prev = synth.make_inputs(spec, rep["B"], rep["H"], rep["W"], rep["seed"],
                                          blobs=(name != "full_noise_128"))
    img, mask = G(label, None, fake, prev)
    assert img.is_cuda and img.shape == (rep["B"], 3, rep["H"], rep["W"]) and mask.shape == (rep["B"], 1, rep["H"], rep["W"])
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    s = rep["sub"]
    d_img = np.abs(img.cpu()[:, :, ::s, ::s].numpy() - g["img"]).max()
    d_mask = np.abs(mask.cpu()[:, :, ::s, ::s].numpy() - g["mask"]).max()
    assert d_img <= TOL and d_mask <= TOL, (d_img, d_mask)
    assert abs(float(img.double().mean()) - rep["img"]["mean"]) < 1e-5
    assert abs(float(mask.double().mean()) - rep["mask"]["mean"]) < 1e-5


def test_full_512_against_oracle_everywhere():
    """BASELINE config 2: 512x512, B=1, fp32, every pixel vs the CPU oracle."""
    spec, sd, G = build("full", 0)
    label, fake, prev = synth.make_inputs(spec, 1, 512, 512, 123)
    img, mask = G(label, None, fake, prev)
    oimg, omask = oracle(spec, sd)(label, None, fake, prev)
    d_img = float((img.cpu() - oimg).abs().max()); d_mask = float((mask.cpu() - omask).abs().max())
    with open("gpurun_out/parity_512.json", "w") as f:
        json.dump({"max_abs_img": d_img, "max_abs_mask": d_mask, "tolerance": NORTH_STAR_TOL}, f)
    assert d_img <= TOL and d_mask <= TOL, (d_img, d_mask)


def test_known_answer_properties_on_gpu():
    spec, sd, G = build("full", 0)
    label, fake, prev = synth.make_inputs(spec, 2, 64, 96, 5)
    a = G(label, None, fake, prev)
    b = G(label, torch.randn_like(label), fake, prev)            # label_prev is dead (F3)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])   # and the path is deterministic
    c = G(label[1:], None, fake[1:], prev[1:])                   # batch independence (F9)
    # a different batch size may pick different tiles / split-K factors, i.e. another fp32
    # summation order: equal to rounding, not bit for bit
    assert float((a[0][1:] - c[0]).abs().max()) <= 5e-5 and float((a[1][1:] - c[1]).abs().max()) <= 5e-5
    assert float(a[0].abs().max()) < 1.0 and 0.0 < float(a[1].min()) and float(a[1].max()) < 1.0


def test_chain_blend_quantise_match_reference_fixture(golden_dir, golden_report):
    rep = golden_report["chain3_128"]
    spec, sd, G = build("full", rep["seed"])
    H, W = rep["H"], rep["W"]
    key = synth.smooth_image(spec, 1, H, W, 1100)
    labels = torch.stack([synth.make_inputs(spec, 1, H, W, 1100 + t)[0] for t in range(3)])
    dains = torch.stack([synth.smooth_image(spec, 1, H, W, 1200 + t) for t in range(3)])
    imgs, masks, fuses = G.chain(key, labels, dains)
    g = np.load(os.path.join(golden_dir, "chain3_128.npz"))
    assert np.abs(fuses[0].cpu().numpy() - g["fuse0"]).max() <= TOL
    assert np.abs(fuses[-1].cpu().numpy() - g["fuse_last"]).max() <= 3 * TOL
    # chain == step-by-step calls of the drop-in object + blend (the reference loop)
    prev = key
    for t in range(3):
        img, mask = G(labels[t], None, dains[t], prev)
        prev = G.blend(img, mask, dains[t])
        assert torch.equal(prev, fuses[t]) and torch.equal(img, imgs[t]) and torch.equal(mask, masks[t])
    q = G.quantise(fuses[-1]).cpu().numpy()[0]
    diff = np.abs(q.astype(int) - g["quant_last"].astype(int))
    assert q.dtype == np.uint8 and diff.max() <= 1 and (diff > 0).mean() < 1e-3
    # quantise is bit-exact given the same float input
    from oracle import generator_ref
    assert np.array_equal(q, generator_ref.quantise_uint8(fuses[-1].cpu()))
    # blend vs oracle on the same inputs
    ob = generator_ref.blend(imgs[0].cpu(), masks[0].cpu(), dains[0])
    assert float((G.blend(imgs[0], masks[0], dains[0]).cpu() - ob).abs().max()) <= 1e-6


def test_edge_shapes_and_errors():
    spec, sd, G = build("full", 0)
    label, fake, prev = synth.make_inputs(spec, 1, 16, 16, 9)    # smallest legal frame
    img, mask = G(label, None, fake, prev)
    oimg, omask = oracle(spec, sd)(label, None, fake, prev)
    # a 1x1 deepest map makes InstanceNorm degenerate (var = 0 -> rstd = 1/sqrt(eps)); compare loosely
    assert float((img.cpu() - oimg).abs().max()) < 5e-3 and float((mask.cpu() - omask).abs().max()) < 5e-3
    label, fake, prev = synth.make_inputs(spec, 1, 32, 80, 9)    # ragged aspect ratio
    img, mask = G(label, None, fake, prev)
    oimg, omask = oracle(spec, sd)(label, None, fake, prev)
    assert float((img.cpu() - oimg).abs().max()) <= 5 * TOL and float((mask.cpu() - omask).abs().max()) <= 5 * TOL
    with pytest.raises(Exception, match="multiples of 16"):
        G(label[:, :, :24, :24], None, fake[:, :, :24, :24], prev[:, :, :24, :24])
    with pytest.raises(ValueError):
        G(label[:, :5], None, fake, prev)
    G2 = rib.Generator(rib.hsm_gen_config())
    with pytest.raises(Exception, match="weights not loaded"):
        G2(label, None, fake, prev)
    with pytest.raises(RuntimeError, match="Missing key"):
        G2.load_state_dict({k: v for k, v in sd.items() if "res_0" not in k})


def test_warp_extension_matches_grid_sample():
    """Extension op, off the reference path: pinned to torch's grid_sample."""
    spec, sd, G = build("full", 0)
    B, H, W = 2, 48, 64
    img = synth.smooth_image(spec, B, H, W, 77)
    flow = (torch.rand(B, 2, H, W) - 0.5) * 12
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    grid = torch.stack([xs, ys], -1)[None].repeat(B, 1, 1, 1)
    grid = grid + torch.stack([flow[:, 0] * 2 / (W - 1), flow[:, 1] * 2 / (H - 1)], -1)
    ref = torch.nn.functional.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)
    out = G.warp(img, flow).cpu()
    assert float((out - ref).abs().max()) <= 2e-5
    # k_warp stages a window of 8 pixels of flow reach around each 16x16 output tile in LDS; flows that reach further
    # take the global-load path.  Sizes that do not tile evenly, flows far beyond the window and beyond the frame.
    for (B, H, W, amp, seed) in ((1, 50, 70, 7.9, 1), (2, 50, 70, 120.0, 2), (1, 17, 33, 40.0, 3), (1, 128, 128, 16.0, 4), (2, 96, 160, 5.0, 5)):
        g = torch.Generator().manual_seed(seed)
        img = synth.smooth_image(spec, B, H, W, 70 + seed)
        flow = (torch.rand(B, 2, H, W, generator=g) - 0.5) * 2 * amp
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
        grid = torch.stack([xs, ys], -1)[None].repeat(B, 1, 1, 1)
        grid = grid + torch.stack([flow[:, 0] * 2 / (W - 1), flow[:, 1] * 2 / (H - 1)], -1)
        ref = torch.nn.functional.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)
        out = G.warp(img, flow).cpu()
        assert float((out - ref).abs().max()) <= 5e-5, (B, H, W, amp)
    # full-size frames (the round-4 kernel was 2.3e-4 off at 1024x1024 - its base grid 2x/(W-1) - 1 differs from
    # torch.linspace by an ulp of 1, i.e. 1e-4 pixels at that width): white-noise image = the steepest gradients a [-1, 1] frame has,
    # smooth +-3 px flow (every tap in the staged window), +-40 px flow (every pixel on the global-load path), a width that is
    # not a multiple of 4 (element-wise staging / stores) and one that does not tile
    for (B, H, W, amp, seed) in ((2, 1024, 1024, 3.0, 11), (1, 1024, 1024, 40.0, 12), (2, 510, 1022, 6.0, 13), (1, 250, 333, 9.0, 14), (4, 250, 333, 50.0, 15)):
        g = torch.Generator().manual_seed(seed)
        img = torch.rand(B, 3, H, W, generator=g) * 2 - 1
        low = torch.randn(B, 2, max(1, H // 32), max(1, W // 32), generator=g)
        flow = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False).clamp(-1, 1) * amp
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
        grid = torch.stack([xs, ys], -1)[None].repeat(B, 1, 1, 1)
        grid = grid + torch.stack([flow[:, 0] * 2 / (W - 1), flow[:, 1] * 2 / (H - 1)], -1)
        ref = torch.nn.functional.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)
        out = G.warp(img, flow).cpu()
        assert float((out - ref).abs().max()) <= 5e-5, (B, H, W, amp)
    # zero flow is the identity (up to the fp32 un-normalisation of the sampling grid)
    assert float((G.warp(img, torch.zeros_like(flow)).cpu() - img).abs().max()) <= 1e-4
    # eight channels (the most rib_warp takes: 8 x 16 KB = 126 KB of staged window in dynamic LDS)
    img8 = torch.rand(1, 8, 96, 160, generator=torch.Generator().manual_seed(9)) * 2 - 1
    flow8 = (torch.rand(1, 2, 96, 160, generator=torch.Generator().manual_seed(10)) - 0.5) * 10
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, 96), torch.linspace(-1, 1, 160), indexing="ij")
    grid = torch.stack([xs, ys], -1)[None] + torch.stack([flow8[:, 0] * 2 / 159, flow8[:, 1] * 2 / 95], -1)
    ref8 = torch.nn.functional.grid_sample(img8, grid, mode="bilinear", padding_mode="border", align_corners=True)
    assert float((G.warp(img8, flow8).cpu() - ref8).abs().max()) <= 5e-5


def test_weight_export_import_roundtrip_is_bit_exact():
    """The single-broadcast multi-GPU hand-off: a second handle fed only the
    exported blob reproduces the first one's frames bit for bit."""
    spec, sd, G = build("full", 0)
    blob = G.export_weights()
    G2 = rib.Generator(rib.hsm_gen_config()).import_weights(blob.clone())
    label, fake, prev = synth.make_inputs(spec, 1, 64, 64, 4)
    a = G(label, None, fake, prev); b = G2(label, None, fake, prev)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_driver_on_gpu_matches_oracle_loop(tmp_path):
    """The folder driver through the real generator (device-side chain + GPU quantise) against the
    oracle's frame-by-frame loop on the same files."""
    import numpy as np
    from PIL import Image
    from render_in_between_amd import evaluator as ev
    from oracle import generator_ref
    from tests.test_driver import _write_example, oracle_labels
    root = str(tmp_path)
    n = _write_example(root, n_key=2, rate=4, H=32, W=48)
    spec, sd, G = build("full", 0)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    E = ev.Evaluator(cfg, lanes=1)
    out = os.path.join(root, "o", "Generated_frames")
    written = E.evaluate_from_folder(G, os.path.join(root, "inputs"), os.path.join(root, "DAIN"),
                                     os.path.join(root, "Predict_motion"), out)
    assert len(written) == n == 5
    R = oracle(spec, sd)
    prev, osz = E.load_image(os.path.join(root, "inputs", "clipA", "0000.png"))
    prev = prev.unsqueeze(0)
    for i in range(1, 4):
        d, _ = E.load_image(os.path.join(root, "DAIN", "clipA", "f%03d.png" % i))
        lab = oracle_labels([E.load_pose(os.path.join(root, "Predict_motion", "clipA", "f%03d_keypoints.json" % i), osz)], 32, 48)[0]
        img, mask = R(lab.unsqueeze(0), None, d.unsqueeze(0), prev)
        prev = generator_ref.blend(img, mask, d.unsqueeze(0))
        want = generator_ref.quantise_uint8(prev).astype(int)
        got = np.asarray(Image.open(written[i])).astype(int)
        assert np.abs(got - want).max() <= 1 and (got != want).mean() < 2e-3, i


def test_driver_lanes_are_bit_identical(tmp_path):
    """Several segments in flight on separate streams/handles give exactly the single-lane frames."""
    import numpy as np
    from PIL import Image
    from render_in_between_amd import evaluator as ev
    from tests.test_driver import _write_example
    root = str(tmp_path)
    n = _write_example(root, n_key=4, rate=2, H=32, W=48)        # 3 independent segments
    spec, sd, G = build("full", 0)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
    a = ev.Evaluator(cfg, lanes=1, batch=1).evaluate_from_folder(G, *dirs, os.path.join(root, "a"))
    b = ev.Evaluator(cfg, lanes=3, batch=1).evaluate_from_folder(G, *dirs, os.path.join(root, "b"))
    assert len(a) == len(b) == n == 7
    for fa, fb in zip(a, b):
        assert np.array_equal(np.asarray(Image.open(fa)), np.asarray(Image.open(fb))), fa


def test_driver_batched_and_chunked_segments(tmp_path):
    """equal-length segments run as ONE chain of batch B, cut into time chunks.  Chunking alone
    (batch 1: the same batch-1 plans, prev handed from chunk to chunk on the device) must not change a byte; batching
    may pick other tile variants for the batch-B plan, so frames agree to the last uint8 step at most (the fp32 frames
    agree to ~1e-5, far below 1/255); ragged groups (5 segments at batch 4 -> 4 + 1) and two lanes included."""
    import numpy as np
    from PIL import Image
    from render_in_between_amd import evaluator as ev
    from tests.test_driver import _write_example
    root = str(tmp_path)
    n = _write_example(root, n_key=6, rate=4, H=32, W=48)        # 5 independent segments of 3 frames
    spec, sd, G = build("full", 0)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
    assert ev.Evaluator.group_segments([(0, [1, 2, 3])] * 5, 4) == [[0, 1, 2, 3], [4]]
    base = ev.Evaluator(cfg, lanes=1, batch=1, chunk=0).evaluate_from_folder(G, *dirs, os.path.join(root, "a"))
    chunked = ev.Evaluator(cfg, lanes=1, batch=1, chunk=2).evaluate_from_folder(G, *dirs, os.path.join(root, "b"))
    batched = ev.Evaluator(cfg, lanes=2, batch=4, chunk=2).evaluate_from_folder(G, *dirs, os.path.join(root, "c"))
    default = ev.Evaluator(cfg).evaluate_from_folder(G, *dirs, os.path.join(root, "d"))      # batch 8 at this size, chunk 4, worker processes
    assert len(base) == len(chunked) == len(batched) == len(default) == n == 21
    for fa, fb, fc, fd in zip(base, chunked, batched, default):
        a = np.asarray(Image.open(fa)).astype(int)
        assert np.array_equal(a, np.asarray(Image.open(fb))), fb
        for f in (fc, fd):
            c = np.asarray(Image.open(f)).astype(int)
            assert np.abs(a - c).max() <= 1 and (a != c).mean() < 2e-3, f


def test_plan_batch_makes_a_samples_frames_independent_of_the_grouping():
    """VERDICT r04 item 3 / SURVEY 4 ("rank r's frames == single-GPU frames bit-for-bit"): with rib_set_plan_batch(n) every plan
    follows batch n's kernel choices, so sample b of a batch-4 chain, of a ragged batch-3 chain and of a batch-1 chain are the
    same bits - at a size where Winograd layers, split-K and the level-wise SPADE GEMMs all take part - while the default
    policy (every batch its own table) only promises ~1e-5.  The setting is per handle, survives clone() and can be reset."""
    spec, sd, G = build("full", 0)
    for (H, W, T) in ((128, 192, 2), (512, 512, 1)):
        labels = torch.stack([torch.cat([synth.make_inputs(spec, 1, H, W, 70 + 10 * t + b)[0] for b in range(4)]) for t in range(T)]).cuda()
        dains = torch.stack([torch.cat([synth.make_inputs(spec, 1, H, W, 70 + 10 * t + b)[1] for b in range(4)]) for t in range(T)]).cuda()
        key = torch.cat([synth.make_inputs(spec, 1, H, W, 60 + b)[2] for b in range(4)]).cuda()
        try:
            for n in (4, 1):
                G.set_plan_batch(n)
                assert G.plan_batch == n and G._lib.rib_get_plan_batch(G._h) == n
                i4, m4, f4 = [t.clone() for t in G.chain(key, labels, dains)]
                i3, m3, f3 = [t.clone() for t in G.chain(key[:3], labels[:, :3], dains[:, :3])]
                assert torch.equal(f3, f4[:, :3]) and torch.equal(m3, m4[:, :3]) and torch.equal(i3, i4[:, :3]), (H, W, n, "B=3 vs B=4")
                for b in range(4):
                    i1, m1, f1 = G.chain(key[b:b + 1], labels[:, b:b + 1], dains[:, b:b + 1])
                    assert torch.equal(f1, f4[:, b:b + 1]) and torch.equal(m1, m4[:, b:b + 1]) and torch.equal(i1, i4[:, b:b + 1]), (H, W, n, b)
                # the single-frame entry follows the same plans
                img, mask = G(labels[0, 1:2], None, dains[0, 1:2], key[1:2])
                assert torch.equal(img, i4[0, 1:2]) and torch.equal(mask, m4[0, 1:2])
                Gc = G.clone()
                assert Gc.plan_batch == n
                assert torch.equal(Gc.chain(key[:2], labels[:, :2], dains[:, :2])[2], f4[:, :2])
            R = oracle(spec, sd)
            oi, om = R(labels[0, :1].cpu(), None, dains[0, :1].cpu(), key[:1].cpu())
            assert float((i4[0, :1].cpu() - oi).abs().max()) < TOL and float((m4[0, :1].cpu() - om).abs().max()) < TOL
        finally:
            G.set_plan_batch(0)
        # default policy: every batch its own choices - close, not necessarily equal
        f4d = G.chain(key, labels, dains)[2]
        assert float((f4d - f4).abs().max()) < 1e-4


def test_driver_frames_do_not_depend_on_grouping_or_world_size(tmp_path, monkeypatch):
    """The folder driver's default (reproducible=True): ragged groups, batch 1 against batch 4 tables aside, a 2-rank split of
    the same folder and a small in-flight window all write byte-identical files; the call-wide back-pressure window holds over
    several clips (it used to be per clip) and the generator gets its own plan policy back."""
    import numpy as np
    from PIL import Image
    from render_in_between_amd import evaluator as ev
    from tests.test_driver import _write_example
    root = str(tmp_path)
    n = sum(_write_example(root, n_key=k, rate=4, H=32, W=48, clip=c, seed=i) for i, (c, k) in enumerate((("clipA", 6), ("clipB", 3), ("clipC", 4))))
    spec, sd, G = build("full", 0)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
    one = ev.Evaluator(cfg, batch=4, chunk=2).evaluate_from_folder(G, *dirs, os.path.join(root, "one"))
    assert len(one) == n == 21 + 9 + 13 and G.plan_batch == 0
    # two ranks, one after the other in this process: their shares are disjoint, complete and the same bytes
    two = []
    for r in range(2):
        two += ev.Evaluator(cfg, batch=4, chunk=2).evaluate_from_folder(G, *dirs, os.path.join(root, "two"), rank=r, world=2)
    assert sorted(os.path.relpath(f, os.path.join(root, "two")) for f in two) == sorted(os.path.relpath(f, os.path.join(root, "one")) for f in one)
    # a window of 2 units over 3 clips (10 units in all): never more in flight, same bytes
    monkeypatch.setattr(ev, "MAX_UNITS_IN_FLIGHT", 2)
    E = ev.Evaluator(cfg, batch=4, chunk=2, lanes=2)
    win = E.evaluate_from_folder(G, *dirs, os.path.join(root, "win"))
    assert 1 <= E.timings["peak_units_in_flight"] <= 2 and E.timings["units"] >= 8
    for f in one:
        rel = os.path.relpath(f, os.path.join(root, "one"))
        a = np.asarray(Image.open(f))
        for other in ("two", "win"):
            assert np.array_equal(a, np.asarray(Image.open(os.path.join(root, other, rel)))), (other, rel)
    # without the policy a ragged group runs another table: still within one uint8 step
    loose = ev.Evaluator(cfg, batch=4, chunk=2, reproducible=False).evaluate_from_folder(G, *dirs, os.path.join(root, "loose"))
    for fa, fb in zip(one, loose):
        a, b = np.asarray(Image.open(fa)).astype(int), np.asarray(Image.open(fb)).astype(int)
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 2e-3, fb


def test_driver_pipeline_stages_can_be_driven_one_at_a_time(tmp_path):
    """The folder driver is a `_FolderPipeline` of explicit stages (round 6): plan a clip, open its units
    (staging blocks + decode tasks), wait for a unit's decodes, upload, render, sink.  Driven by hand here, unit by unit and in
    both file-side modes, they write the bytes the whole call writes, hand every shared block back, and a unit's staging view
    is gone before its block returns to the free list."""
    import numpy as np
    from PIL import Image
    from concurrent.futures import Future, ThreadPoolExecutor
    from render_in_between_amd import evaluator as ev
    from tests.test_driver import _write_example
    root = str(tmp_path)
    n = _write_example(root, n_key=4, rate=4, H=32, W=48)        # 3 segments of 3 frames + 4 key frames
    spec, sd, G = build("full", 0)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
    whole = ev.Evaluator(cfg, batch=2, chunk=2, lanes=1).evaluate_from_folder(G, *dirs, os.path.join(root, "whole"))
    assert len(whole) == n == 13
    for mode in ("process", "thread"):
        E = ev.Evaluator(cfg, batch=2, chunk=2, lanes=1, io_mode=mode)
        E._pool, E._finishers = ThreadPoolExecutor(4), ThreadPoolExecutor(4)
        out = os.path.join(root, "staged_" + mode)
        with E._plan_policy(G, True):
            pipe = ev._FolderPipeline(E, G, 0, 1, None)
            assert pipe.native and pipe.gpu_labels and (pipe.procs is not None) == (mode == "process")
            clip = pipe.plan_clip("clipA", *dirs, out)
            # groups of 2 + 1 segments, chunks of 2 + 1 steps: four units
            assert [(len(m), c1 - c0) for _, m, c0, c1 in clip.units] == [(2, 2), (2, 1), (1, 2), (1, 1)] and clip.keys == [0, 4, 8, 12]
            st = torch.cuda.current_stream(G.device)
            for ui in range(len(clip.units)):
                pipe.open_units(clip, ui)                          # no decode-ahead: exactly this unit
                assert clip.opened == ui + 1 and ui in clip.stage and (ui in clip.stage_blk) == (mode == "process")
                poses, gt = pipe.wait_decoded(clip, ui)
                assert len(poses) == len(clip.unit_frames(ui)) and (gt is not None) == (clip.units[ui][2] == 0)
                lab, dn, gtd, ready = pipe.upload(clip, ui, G, st, poses, gt)
                assert lab.shape[:2] == dn.shape[:2] == (clip.units[ui][3] - clip.units[ui][2], len(clip.units[ui][1])) and lab.shape[2] == 22
                r = pipe.render(clip, ui, G, st, lab, dn, gtd, ready)
                names = pipe.sink(clip, ui, r, 0.0, 0.0)           # on this thread: returns when the unit's files exist
                assert names == [clip.names[i] for i in clip.unit_frames(ui)] and all(os.path.exists(f) for f in names)
                assert ui not in clip.stage and ui not in clip.stage_blk and r["keep"] is None
                for j, i in enumerate(clip.unit_frames(ui)):
                    f = Future()
                    f.set_result(names)
                    clip.futs[i] = (f, j)
            for k in clip.keys:
                pipe.submit_load(clip, k)                          # key frames pass through
            written = pipe.drain()
        assert [os.path.relpath(f, out) for f in written] == [os.path.relpath(f, os.path.join(root, "whole")) for f in whole]
        for fa, fb in zip(whole, written):
            assert np.array_equal(np.asarray(Image.open(fa)), np.asarray(Image.open(fb))), (mode, fb)
        del pipe, clip
        ev._shm_trim()
        assert all(len(v) <= ev._SHM_KEEP for v in ev._SHM_FREE.values())
        assert sum(len(v) for v in ev._SHM_FREE.values()) == len(ev._SHM_ALL)      # every block is back on a free list


# the half-storage mode's promise on a [-1, 1] frame (max-abs <= 3e-2 / mean <= 3e-3 at >= 600 frames/s)
F16_MAX, F16_MEAN = 2.6e-2, 2e-3      # = 1.5 x measured (1.7e-2 worst max, 1.3e-3 worst mean over the sizes and the 32-frame chain)


def _assert_on_rounding_model(spec, sd, inputs, fmt, got):
    """The 16-bit bounds are model-derived (round 6): oracle/precision_model.py applies the format's roundings where the kernels
    apply them; the GPU error must sit on it - mean within 20 %, max (a tail statistic) within a factor 1.6.  The constants
    above stay as a tripwire.  tests/test_precision_model.py makes the same comparison on the CPU against the committed figures."""
    from oracle import precision_model
    m = precision_model.predict(spec, sd, *inputs, fmt=fmt)
    for k in ("mean_abs_img", "mean_abs_mask"):
        assert 0.8 <= got[k] / m[k] <= 1.2, (fmt, k, got[k], m[k])
    for k in ("max_abs_img", "max_abs_mask"):
        assert 1 / 1.6 <= got[k] / m[k] <= 1.6, (fmt, k, got[k], m[k])


def _scaled_checkpoint(sd, s):
    """The same checkpoint with every FOLDED filter s times larger (what a trained checkpoint may bring: SURVEY f-3):
    spectral-norm convolutions through their u vector (W / (u . W v) grows by s when u shrinks by s), the others directly."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".weight_u"):
            out[k] = v / s
        elif k.endswith(".layers.conv.weight"):
            out[k] = v * s
        else:
            out[k] = v
    return out


def test_half_mode_stays_finite_or_fails_loudly_on_larger_filters():
    """DESIGN 6 admitted that the half mode's range argument had only seen He-scaled synthetic filters.  Folded filters 4x and
    16x larger: the mode either renders finite frames or REFUSES the checkpoint when it is loaded (rib_finalize_weights checks
    the folded filters against 65504, Generator.load_state_dict runs a full-range probe frame) - it never hands out NaN
    frames silently.  bf16 (fp32's exponent range) and fp32 take the same checkpoints and agree with the oracle."""
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    sd0 = synth.make_state_dict(spec, 0)
    label, fake, prev = synth.make_inputs(spec, 1, 128, 128, 9)
    outcome = {}
    for s in (4.0, 16.0, 4096.0):
        sd = _scaled_checkpoint(sd0, s)
        oimg, omask = oracle(spec, sd)(label, None, fake, prev)
        assert bool(torch.isfinite(oimg).all())
        G32 = rib.Generator(cfg).eval(); G32.load_state_dict(sd)
        i32, m32 = G32(label, None, fake, prev)
        # (every layer s times steeper: the tanh head saturates and single pixels near a zero crossing may flip - compare means)
        assert bool(torch.isfinite(i32).all()) and bool(torch.isfinite(m32).all()), s
        if s <= 16:       # (beyond that the fp32 network itself is chaotic: 4096^5 through the un-normalised condition encoder)
            assert float((i32.cpu() - oimg).abs().mean()) <= 1e-3 and float((m32.cpu() - omask).abs().mean()) <= 1e-3, s
        Gb = rib.Generator(cfg, compute_dtype="bf16").eval(); Gb.load_state_dict(sd)
        ib, mb = Gb(label, None, fake, prev)
        assert bool(torch.isfinite(ib).all()) and bool(torch.isfinite(mb).all()), s
        Gh = rib.Generator(cfg, compute_dtype="f16").eval()
        try:
            Gh.load_state_dict(sd)
        except (FloatingPointError, Exception) as e:      # noqa: BLE001
            from render_in_between_amd import _native
            assert isinstance(e, (FloatingPointError, _native.RibError)), repr(e)
            assert "range" in str(e), str(e)
            outcome[s] = "refused: " + type(e).__name__
            continue
        ih, mh = Gh(label, None, fake, prev)
        assert bool(torch.isfinite(ih).all()) and bool(torch.isfinite(mh).all()), s
        outcome[s] = "finite, img max-abs %.2e" % float((ih.cpu() - oimg).abs().max())
    with open("gpurun_out/f16_range_stress.json", "w") as f:
        json.dump({str(k): v for k, v in outcome.items()}, f, indent=1)
    assert outcome[4096.0].startswith("refused")          # 0.4 x 4096 is far inside half's range, the activations it drives are not


def test_half_storage_mode_meets_its_bound():
    """RIB_DTYPE_F16: the 16-bit storage layouts and kernels of the bf16 mode with IEEE half elements
    (v_mfma_f32_32x32x16_f16).  Same bytes, same launches, 8x smaller rounding unit: the CPU model of the roundings
    (tools/probes/bf16_policy_sim.py) predicts 9e-3 max / 1e-3 mean.  Outputs finite (half's range: 65504) at every size;
    deterministic; blob round trip."""
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    sd = synth.make_state_dict(spec, 0)
    G = rib.Generator(cfg, compute_dtype="f16").eval()
    G.load_state_dict(sd)
    R = oracle(spec, sd)
    rep = {}
    for (B, H, W, seed) in ((1, 256, 256, 2), (2, 48, 80, 3), (1, 512, 512, 6)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, seed)
        img, mask = G(label, None, fake, prev)
        assert img.dtype == torch.float32 and bool(torch.isfinite(img).all()) and bool(torch.isfinite(mask).all())
        oimg, omask = R(label, None, fake, prev)
        r = {"max_abs_img": float((img.cpu() - oimg).abs().max()), "max_abs_mask": float((mask.cpu() - omask).abs().max()),
             "mean_abs_img": float((img.cpu() - oimg).abs().mean()), "mean_abs_mask": float((mask.cpu() - omask).abs().mean())}
        rep["%dx%dx%d" % (B, H, W)] = r
        assert r["max_abs_img"] <= F16_MAX and r["max_abs_mask"] <= F16_MAX and r["mean_abs_img"] <= F16_MEAN and r["mean_abs_mask"] <= F16_MEAN, (B, H, W, r)
        if H <= 256:      # the bound that means something: the kernels sit on the CPU model of the format's roundings
            _assert_on_rounding_model(spec, sd, (label, fake, prev), "f16", r)
    with open("gpurun_out/parity_f16.json", "w") as f:
        json.dump(rep, f)
    label, fake, prev = synth.make_inputs(spec, 1, 64, 64, 5)
    a = [t.clone() for t in G(label, None, fake, prev)]
    b = G(label, None, fake, prev)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    G2 = rib.Generator(cfg, compute_dtype="f16").import_weights(G.export_weights())
    c = G2(label, None, fake, prev)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    from render_in_between_amd import _native
    with pytest.raises(_native.RibError):                       # a bf16 handle's blob has the same size but another mode in its header
        rib.Generator(cfg, compute_dtype="bf16").import_weights(G.export_weights())


def test_bf16_storage_mode_error_is_bounded():
    """BASELINE configs[2] mode: bf16 NHWC activations and filters in HBM / LDS, bf16 matrix-core operands, fp32
    accumulation, fp32 InstanceNorm statistics (of the rounded tensors) and fp32 SPADE arithmetic.  The reference has
    no counterpart (fp32 only), so the tolerance is this mode's own: every stored tensor is rounded to 8 mantissa bits
    (2^-9 relative) and the error grows to ~1.5 % over the ~40 layers of the deepest path (tools/precision_debug.py).
    That is the FORMAT's error, not the kernels': a CPU model that rounds exactly the tensors and filters the kernels
    round (tools/probes/bf16_policy_sim.py) lands on the same 1.0e-1 max / 8.7e-3 mean, and no cheap subset of the
    roundings carries it.  Bounds = 1.5 x the measured error (written to gpurun_out/), so a regression of the
    arithmetic - an fp32 accumulation or a statistic lost to bf16 - fails the test."""
    cfg = rib.hsm_gen_config()
    spec = rib.GenSpec.from_cfg(cfg)
    sd = synth.make_state_dict(spec, 0)
    G = rib.Generator(cfg, compute_dtype="bf16").eval()
    G.load_state_dict(sd)
    R = oracle(spec, sd)
    rep = {}
    for (B, H, W, seed) in ((1, 256, 256, 2), (2, 48, 80, 3), (1, 16, 16, 4)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, seed)
        img, mask = G(label, None, fake, prev)
        assert img.dtype == torch.float32 and mask.dtype == torch.float32          # the boundary stays fp32 NCHW
        oimg, omask = R(label, None, fake, prev)
        d_img = float((img.cpu() - oimg).abs().max()); d_mask = float((mask.cpu() - omask).abs().max())
        m_img = float((img.cpu() - oimg).abs().mean()); m_mask = float((mask.cpu() - omask).abs().mean())
        rep["%dx%dx%d" % (B, H, W)] = {"max_abs_img": d_img, "max_abs_mask": d_mask, "mean_abs_img": m_img, "mean_abs_mask": m_mask}
        if H >= 48:      # (a 1x1 deepest map makes InstanceNorm degenerate: compared loosely in fp32 too)
            # measured (round 3): img 9.6e-2 / 1.10e-1 max, 8.2e-3 / 9.9e-3 mean; mask 2.8e-2 / 3.0e-2 max, 3.6e-3 / 4.0e-3 mean
            assert d_img <= 1.7e-1 and d_mask <= 4.5e-2 and m_img <= 1.5e-2 and m_mask <= 6e-3, (B, H, W, d_img, d_mask, m_img, m_mask)
            _assert_on_rounding_model(spec, sd, (label, fake, prev), "bf16", rep["%dx%dx%d" % (B, H, W)])
        assert bool(torch.isfinite(img).all()) and bool(torch.isfinite(mask).all())
    with open("gpurun_out/parity_bf16_256.json", "w") as f:
        json.dump(rep, f)
    # deterministic, and a bf16 handle fed only the exported blob of another bf16 handle reproduces it bit for bit
    label, fake, prev = synth.make_inputs(spec, 1, 64, 64, 5)
    a = [t.clone() for t in G(label, None, fake, prev)]
    b = G(label, None, fake, prev)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    G2 = rib.Generator(cfg, compute_dtype="bf16").import_weights(G.export_weights())
    c = G2(label, None, fake, prev)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    from render_in_between_amd import _native
    with pytest.raises(_native.RibError):                                          # an fp32 handle's blob has another layout
        rib.Generator(cfg).import_weights(G.export_weights())
    # the fp32 default is untouched by the existence of the mode
    G32 = rib.Generator(cfg).eval(); G32.load_state_dict(sd)
    label, fake, prev = synth.make_inputs(spec, 1, 256, 256, 2)
    i32, m32 = G32(label, None, fake, prev)
    oimg, omask = R(label, None, fake, prev)
    assert float((i32.cpu() - oimg).abs().max()) <= TOL


def test_instance_norm_statistics_survive_a_large_channel_mean():
    """|mean| >> std in a normalised tensor: down_first gets a bias of +100 on every channel, so the tensor the first
    SPADE normalises has mean ~100 and std ~1 (mean^2 / var ~ 1e4).  The producers accumulate sum(x), sum(x^2) in fp64
    from the element level, so E[x^2] - mean^2 keeps the variance; fp32 per-tile partials used to lose ~1e-3 of it here.
    (At this offset the fp32 convolution outputs themselves differ by ~1e-5 between any two fp32 implementations, hence
    the north star's 1e-3 and not the usual 2e-4.)"""
    spec, sd, _ = build("full", 0)
    sd2 = dict(sd)
    sd2["down_first.layers.conv.bias"] = sd["down_first.layers.conv.bias"] + 100.0
    G = rib.Generator(rib.hsm_gen_config()).eval()
    G.load_state_dict(sd2)
    for (H, W) in ((64, 64), (256, 256)):
        label, fake, prev = synth.make_inputs(spec, 1, H, W, 21)
        img, mask = G(label, None, fake, prev)
        oimg, omask = oracle(spec, sd2)(label, None, fake, prev)
        d = (float((img.cpu() - oimg).abs().max()), float((mask.cpu() - omask).abs().max()))
        assert d[0] <= NORTH_STAR_TOL and d[1] <= NORTH_STAR_TOL, (H, W, d)


def test_long_autoregressive_chain_stays_within_tolerance():
    """Error accumulation over a 12-step device-side chain (prev <- fused frame) vs the CPU oracle's
    frame-by-frame loop: the 1e-3 fp32 bar must hold at the END of the chain, not only per frame."""
    from oracle import generator_ref
    spec, sd, G = build("full", 0)
    H = W = 64
    T = 12
    key = synth.smooth_image(spec, 1, H, W, 300)
    labels = torch.stack([synth.make_inputs(spec, 1, H, W, 300 + t)[0] for t in range(T)])
    dains = torch.stack([synth.smooth_image(spec, 1, H, W, 400 + t) for t in range(T)])
    _, _, fuses = G.chain(key, labels, dains, want_all=False)
    _, _, ofuses = generator_ref.autoregressive_segment(oracle(spec, sd), key, list(labels), list(dains))
    d_last = float((fuses[-1].cpu() - ofuses[-1]).abs().max())
    d_max = max(float((fuses[t].cpu() - ofuses[t]).abs().max()) for t in range(T))
    with open("gpurun_out/parity_chain12_64.json", "w") as f:
        json.dump({"steps": T, "max_abs_last_frame": d_last, "max_abs_any_frame": d_max}, f)
    assert d_max <= NORTH_STAR_TOL, (d_last, d_max)


def test_full_1024_against_oracle():
    """BASELINE configs[4] resolution (1024x1024; one sample of the batch): every pixel vs the CPU
    oracle.  Guards the large-map index arithmetic and the 1M-element InstanceNorm statistics
    (fp32 per-tile partials + fp64 finalize, SURVEY §7 'hard parts')."""
    spec, sd, G = build("full", 0)
    label, fake, prev = synth.make_inputs(spec, 1, 1024, 1024, 77)
    img, mask = G(label, None, fake, prev)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    oimg, omask = oracle(spec, sd)(label, None, fake, prev)
    d_img = float((img.cpu() - oimg).abs().max()); d_mask = float((mask.cpu() - omask).abs().max())
    with open("gpurun_out/parity_1024.json", "w") as f:
        json.dump({"max_abs_img": d_img, "max_abs_mask": d_mask, "tolerance": NORTH_STAR_TOL}, f)
    assert d_img <= TOL and d_mask <= TOL, (d_img, d_mask)
    G._ws.clear()          # release the 4 GB workspace of this shape
    torch.cuda.empty_cache()


# ---- label rasteriser: rib_rasterise, bit-exact -----------------------------
def _raster(G, frames, H, W):
    from render_in_between_amd import rasterise
    return rasterise.rasterise_labels(G, frames, H, W).cpu()


def _oracle_raster(frames, H, W):
    from tests.test_driver import oracle_labels
    return oracle_labels(frames, H, W)


def test_rasteriser_matches_reference_fixtures_bit_for_bit(golden_dir):
    """GPU label maps == the reference's own _generate_skeleton / _generate_pose_map outputs
    (tests/golden/raster_*.npz, made by make_golden_raster.py from the reference's functions)."""
    spec, sd, G = build("full", 0)
    for n in "abcde":
        g = np.load(os.path.join(golden_dir, "raster_%s.npz" % n))
        H, W = [int(v) for v in g["size"]]
        lm = [tuple(v) for v in g["landmarks"]]
        got = _raster(G, [(lm, list(g["conf"]))], H, W)[0].numpy()
        want_sk = ((g["skeleton"].astype(np.float32) / 255.0 - 0.5) / 0.5).transpose(2, 0, 1)
        assert got.shape == (22, H, W)
        assert np.array_equal(got[:3], want_sk), n
        assert np.array_equal(got[3:], g["pose_map"]), n


def test_rasteriser_matches_oracle_on_hard_cases():
    """Batches of frames; joints on the borders (reflecting gaussian, clamped strokes), short limbs
    (overlapping end discs), integer-valued joints (knife-edge truncations), missing joints."""
    spec, sd, G = build("full", 0)
    rng = np.random.default_rng(5)
    for (H, W, mode) in [(64, 64, "frac"), (96, 160, "int"), (128, 128, "border"), (48, 80, "tiny"), (256, 192, "frac")]:
        frames = []
        for t in range(6):
            if mode == "border":
                xy = np.stack([rng.choice([0.2, 1.7, W - 1.3, W - 0.4, W / 2], 19) + rng.uniform(0, 0.2, 19),
                               rng.choice([0.3, 2.2, H - 2.6, H - 0.2, H / 3], 19) + rng.uniform(0, 0.2, 19)], 1)
            elif mode == "tiny":
                xy = np.array([W / 2, H / 2]) + rng.uniform(-6, 6, (19, 2))
            else:
                xy = np.stack([rng.uniform(0, W, 19), rng.uniform(0, H, 19)], 1)
            xy = np.round(xy) if mode == "int" else np.round(xy, 3)
            conf = rng.uniform(0.2, 1, 19)
            conf[rng.integers(0, 19, 3)] = 0.0
            frames.append(([tuple(v) for v in xy], list(conf)))
        got = _raster(G, frames, H, W)
        want = _oracle_raster(frames, H, W)
        assert torch.equal(got[:, 3:], want[:, 3:]), (mode, "heat-maps")
        assert torch.equal(got[:, :3], want[:, :3]), (mode, "skeleton", int((got[:, :3] != want[:, :3]).sum()))


def test_rasteriser_rejects_bad_tables():
    from render_in_between_amd import rasterise, _native
    spec, sd, G = build("full", 0)
    w, r = rasterise.gaussian_weights(5)
    strokes = np.zeros((1, 18), rasterise.STROKE_DTYPE)
    peaks = np.full((1, 19, 2), -1, np.int32)
    out = G.rasterise(strokes, peaks, w, r, 32, 48)
    assert out.shape == (1, 22, 32, 48) and float(out[:, :3].max()) == -1.0 and float(out[:, 3:].abs().max()) == 0.0
    bad = peaks.copy(); bad[0, 0] = (48, 3)
    with pytest.raises(_native.RibError):
        G.rasterise(strokes, bad, w, r, 32, 48)
    with pytest.raises(_native.RibError):
        G.rasterise(strokes, peaks[:, :5], w, r, 32, 48)          # 3 + 5 != label_nc


def test_every_kernel_variant_is_correct_wherever_it_fits():
    """Forces, on representative launches (3x3 with a fused 1x1 shortcut, stride 2, upsampled gather,
    1x1, 16-column, SPADE), every tile variant x split-K that the plan accepts - including the
    in-workgroup split-K (KW) and three-slices-per-barrier (TB) twins - and checks the frame against
    the plan's default choice.  A variant that does not fit a launch is rejected by the plan builder."""
    import ctypes as C
    from render_in_between_amd import _native
    spec, sd, _ = build("full", 0)
    G = rib.Generator(rib.hsm_gen_config(), use_tuning=False).eval()
    G.load_state_dict(sd)
    lib, h = G._lib, G._h
    B, H, W = 1, 64, 64
    label, fake, prev = synth.make_inputs(spec, B, H, W, 3)
    img0, mask0 = [t.clone() for t in G(label, None, fake, prev)]
    R = oracle(spec, sd)
    ri, rm = R(label, None, fake, prev)
    assert (img0.cpu() - ri).abs().max() < TOL and (mask0.cpu() - rm).abs().max() < TOL
    ops = ["down_1.conv_block_1", "ref_embedding.down_1", "flow_network_temp.up_flow.3", "res_0.conv_block_0",
           "flow_network_temp.res_flow.0.conv_block_s", "conv_img", "down_0.0.spade", "up_3.1.spade", "flow_network_temp.down_lbl.0"]
    g12 = (C.c_int * 12)()
    tried = accepted = 0
    seen_kw = seen_tb = 0
    for name in ops:
        for vi in range(lib.rib_num_variants()):
            if lib.rib_variant_info(vi, g12) != 0:
                continue                                   # bf16 twins are covered by the bf16 test
            for ks in (1, 2, 4):
                tried += 1
                assert lib.rib_set_choice(h, B, H, W, name.encode(), vi, ks) == 0
                try:
                    img, mask = G(label, None, fake, prev)
                except _native.RibError:
                    continue                               # does not fit this launch
                accepted += 1
                seen_kw += g12[10] > 1
                seen_tb += g12[11] > 1
                e = max(float((img - img0).abs().max()), float((mask - mask0).abs().max()))
                assert e < 5e-5, (name, list(g12), ks, e)
        assert lib.rib_set_choice(h, B, H, W, name.encode(), -1, 1) == 0
    assert accepted > 150 and seen_kw > 10 and seen_tb > 10, (tried, accepted, seen_kw, seen_tb)


def test_head_convolutions_agree_across_their_three_kernels(monkeypatch):
    """conv_img (16 -> 3) and conv_mask.0 (32 -> 1) run as k_conv_head (matrix cores, the nine taps as GEMM columns; the
    mask head also writes the driver's blend); with RIB_NO_HEADCONV as k_conv_small (direct, vector ALUs); with
    RIB_NO_SMALLCONV through k_igemm's 16-column path.  Same frame all three ways, at odd sizes too, and the fused
    blend equals forward + rib_blend bit for bit."""
    spec, sd, _ = build("full", 0)
    for (B, H, W, seed) in ((1, 64, 64, 1), (2, 48, 80, 2), (1, 256, 256, 3), (1, 16, 16, 4)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, seed)
        outs, launches = [], []
        for env in (None, "RIB_NO_HEADCONV", "RIB_NO_SMALLCONV"):
            monkeypatch.delenv("RIB_NO_HEADCONV", raising=False)
            monkeypatch.delenv("RIB_NO_SMALLCONV", raising=False)
            if env:
                monkeypatch.setenv(env, "1")
            G = rib.Generator(rib.hsm_gen_config()).eval(); G.load_state_dict(sd)
            i1, m1, f1 = [t.clone() for t in G.forward_blend(label, None, fake, prev)]
            i2, m2 = G(label, None, fake, prev)
            assert torch.equal(i1, i2) and torch.equal(m1, m2)
            assert torch.equal(f1, G.blend(i2, m2, fake))                 # fused blend (or the fallback launch) == rib_blend
            outs.append((i1, m1)); launches.append(G.num_launches(B, H, W))
            info = [o for o in G.launch_info(B, H, W) if o["name"] in ("conv_img", "flow_network_temp.conv_mask.0")]
            assert all(("head" in o["tile"]) == (env is None) for o in info), (env, info)
            torch.cuda.synchronize()
            del G
        monkeypatch.delenv("RIB_NO_SMALLCONV", raising=False)

        # (without the head kernel conv_img writes its NHWC copy again and pack.img9 is back: +1; k_igemm may split K at these sizes)
        assert 0 <= launches[1] - launches[0] <= 1 and abs(launches[1] - launches[2]) <= 1
        for (i, m) in outs[1:]:
            e = max(float((outs[0][0] - i).abs().max()), float((outs[0][1] - m).abs().max()))
            assert e < (2e-6 if H >= 48 else 1e-5), (B, H, W, e)     # (a 1x1 deepest map amplifies rounding: see the edge-shape test)


def test_upsample_convolutions_as_phase_convolutions_match_oracle_at_odd_sizes():
    """The mask network's Upsample(2) -> 3x3 convolutions run as four 2x2 phase convolutions of the
    half-resolution map (4/9 of the MACs, filters summed at fold time).  Sizes whose source maps do not
    tile evenly (H/8 not a multiple of the 8 / 16-pixel tiles) exercise the tile borders."""
    spec, sd, G = build("full", 0)
    R = oracle(spec, sd)
    for (B, H, W, seed) in ((1, 48, 80, 5), (2, 80, 48, 6), (1, 176, 112, 7)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, seed)
        img, mask = G(label, None, fake, prev)
        ri, rm = R(label, None, fake, prev)
        assert float((img.cpu() - ri).abs().max()) < TOL and float((mask.cpu() - rm).abs().max()) < TOL, (B, H, W)


def test_chain_with_batched_label_work_matches_the_per_frame_chain(monkeypatch):
    """rib_chain runs the label-only launches (pack.label, down_first, the mask network's label branch) once for
    the whole segment at batch T*B; with RIB_NO_LABEL_BATCH they run per frame.  Same frames either way, bit for
    bit (the batched launches use the frame plan's kernel choices), and they match the oracle loop."""
    spec, sd, G = build("full", 0)
    T, H, W = 5, 64, 96
    labels = torch.stack([synth.make_inputs(spec, 1, H, W, 40 + t)[0] for t in range(T)])
    dains = torch.stack([synth.make_inputs(spec, 1, H, W, 40 + t)[1] for t in range(T)])
    key = synth.make_inputs(spec, 1, H, W, 39)[2]
    monkeypatch.delenv("RIB_NO_LABEL_BATCH", raising=False)
    i1, m1, f1 = [t.clone() for t in G.chain(key, labels, dains)]
    assert G._lib.rib_chain_workspace_bytes(G._h, T, 1, H, W) > G._lib.rib_workspace_bytes(G._h, 1, H, W)
    monkeypatch.setenv("RIB_NO_LABEL_BATCH", "1")
    assert G._lib.rib_chain_workspace_bytes(G._h, T, 1, H, W) == G._lib.rib_workspace_bytes(G._h, 1, H, W)
    i2, m2, f2 = G.chain(key, labels, dains)
    torch.cuda.synchronize()
    # the batched launches follow the frame plan's kernel choices: bit-identical frames
    assert torch.equal(f1, f2) and torch.equal(m1, m2) and torch.equal(i1, i2)
    R = oracle(spec, sd)
    from oracle import generator_ref
    prev = key
    for t in range(T):
        oi, om = R(labels[t], None, dains[t], prev)
        prev = generator_ref.blend(oi, om, dains[t])
        assert float((f1[t].cpu() - prev).abs().max()) < TOL, t


def test_kernel_time_profiling_counts_every_launch():
    """bench.py's roofline: rib_profile_begin_kernels binds a (start, stop) event pair to every dispatch.  Every launch of the
    plan is counted in its class, the kernels' own times are positive and add up to roughly the forward's wall time (they
    exclude the gaps between dependent launches and include nothing else), the interval-event mode still works, and
    profiling leaves the frame untouched."""
    import time
    spec, sd, G = build("full", 0)
    B, H, W = 1, 256, 256
    label, fake, prev = [t.cuda() for t in synth.make_inputs(spec, B, H, W, 5)]
    want = [t.clone() for t in G(label, None, fake, prev)]
    n = G.num_launches(B, H, W)
    for _ in range(3):
        G(label, None, fake, prev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        G(label, None, fake, prev)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / 10 * 1e3
    for kernels in (True, False):
        G.profile_begin(kernels=kernels)
        for _ in range(4):
            got = G(label, None, fake, prev)
        prof = G.profile_collect()
        assert sum(v["launches"] for v in prof.values()) == 4 * n, (kernels, prof)
        assert prof["igemm"]["launches"] > 0 and all(v["ms"] >= 0 for v in prof.values())
        per_fwd = sum(v["ms"] for v in prof.values()) / 4
        assert 0.5 * wall_ms < per_fwd < (1.6 if kernels else 3.0) * wall_ms, (kernels, per_fwd, wall_ms)
        assert all(torch.equal(a, b) for a, b in zip(want, got))
    got = G(label, None, fake, prev)              # profiling is off again
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(want, got))


def test_chain_graph_replay_is_bit_identical():
    """rib_chain as ONE HIP graph launch.  The first call with a given shape and set of tensors captures
    the segment's launches, later calls replay the graph: same kernels, parameters and order, so the frames equal the
    launch-by-launch path bit for bit - also with batch 2, other inputs in the same tensors, and after the option is
    switched off again.  Generator.chain alternates two output sets while the option is on: 2 captures, then replays."""
    spec, sd, G = build("full", 0)
    T, B, H, W = 4, 2, 64, 96
    labels = torch.stack([synth.make_inputs(spec, B, H, W, 60 + t)[0] for t in range(T)]).cuda()
    dains = torch.stack([synth.make_inputs(spec, B, H, W, 60 + t)[1] for t in range(T)]).cuda()
    key = synth.make_inputs(spec, B, H, W, 59)[2].cuda()
    want = [t.clone() for t in G.chain(key, labels, dains)]
    torch.cuda.synchronize()
    G.set_graph_replay(True)
    side = torch.cuda.Stream()          # (the NULL stream cannot be captured)
    try:
        with torch.cuda.stream(side):
            for k in range(5):
                got = G.chain(key, labels, dains)
                torch.cuda.synchronize()
                assert all(torch.equal(a, b) for a, b in zip(want, got)), k
            st = G.graph_stats()
            assert st == {"captures": 2, "replays": 3}, st
            # new values in the SAME input tensors: the replayed graph reads them (nothing was baked in but addresses)
            labels2 = torch.stack([synth.make_inputs(spec, B, H, W, 80 + t)[0] for t in range(T)]).cuda()
            keep = labels.clone()
            labels.copy_(labels2)
            got2 = [t.clone() for t in G.chain(key, labels, dains)]
            assert G.graph_stats()["replays"] == 4
            G.set_graph_replay(False)
            want2 = G.chain(key, labels, dains)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(want2, got2))
            assert not torch.equal(want2[2], want[2])
            labels.copy_(keep)
            torch.cuda.synchronize()
    finally:
        G.set_graph_replay(False)


def test_full_size_properties_512():
    """Size-independent properties at BASELINE.json's full size (512x512), where the oracle is only run once:
    batch independence (a sample's frame does not depend on its batch mates), determinism, label_prev is dead,
    the mask is a probability, the blend is the convex combination of evaluator.py:256-258, quantise is idempotent
    on already quantised values, and the uint8 frame is what the oracle's quantiser makes of the same floats."""
    from oracle import generator_ref
    spec, sd, G = build("full", 0)
    label, fake, prev = synth.make_inputs(spec, 2, 512, 512, 11)
    i2, m2 = [t.clone() for t in G(label, None, fake, prev)]
    j2, n2 = G(label, torch.zeros_like(label), fake, prev)
    assert torch.equal(i2, j2) and torch.equal(m2, n2)
    for b in range(2):
        i1, m1 = G(label[b:b + 1], None, fake[b:b + 1], prev[b:b + 1])
        assert float((i1 - i2[b:b + 1]).abs().max()) <= 5e-5 and float((m1 - m2[b:b + 1]).abs().max()) <= 5e-5
    assert float(i2.abs().max()) <= 1.0 and 0.0 <= float(m2.min()) and float(m2.max()) <= 1.0
    fuse = G.blend(i2, m2, fake)
    lo = torch.minimum(i2, fake.to(i2.device)) - 1e-6
    hi = torch.maximum(i2, fake.to(i2.device)) + 1e-6
    assert bool(((fuse >= lo) & (fuse <= hi)).all())                     # convex combination, channel by channel
    q = G.quantise(fuse)
    assert np.array_equal(q.cpu().numpy(), np.stack([generator_ref.quantise_uint8(fuse[b:b + 1].cpu()) for b in range(2)]).reshape(q.shape))
    back = (q.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
    q2 = G.quantise(back + 1e-4)      # already on the uint8 grid (nudged off the truncation edge): unchanged
    assert torch.equal(q, q2)


# ---- round 2: the configurations of BASELINE.json that had no GPU test, and the reference-pinned quantiser ----
def test_quantise_equals_the_reference_bytes(golden_dir):
    """rib_quantise vs the bytes the reference's own tensor2images produced (tests/golden/quant_ref.npz, made by
    make_golden_quant.py from PGNR/utils/utils.py:122-147): bit for bit on the reference's chain frame and on a tensor
    of knife-edge values (every k/255 edge and its fp32 neighbours, out-of-range values, +-inf)."""
    spec, sd, G = build("full", 0)
    g = np.load(os.path.join(golden_dir, "quant_ref.npz"))
    c = np.load(os.path.join(golden_dir, "chain3_128.npz"))
    q = G.quantise(torch.from_numpy(c["fuse_last"])).cpu().numpy()[0]
    assert q.dtype == np.uint8 and np.array_equal(q, g["chain_last_quant"])
    e = G.quantise(torch.from_numpy(g["edge_in"])).cpu().numpy()[0]
    assert np.array_equal(e, g["edge_quant"]), int((e != g["edge_quant"]).sum())


def test_config3_32_frame_chain_512_fp32_and_bf16_against_the_oracle_loop():
    """BASELINE configs[2]: one 32-frame autoregressive segment at 512x512, batch 1 (prev <- fused frame on the device),
    against the CPU oracle's frame-by-frame loop (evaluator.py:238-262).  fp32: the north star's 1e-3 must hold on the
    LAST frame, after 32 steps of feedback.  bf16 mode: its own stated bound (there is no reference counterpart)."""
    from oracle import generator_ref
    spec, sd, G = build("full", 0)
    H = W = 512
    T = 32
    key = synth.smooth_image(spec, 1, H, W, 3100)
    labels = torch.stack([synth.make_inputs(spec, 1, H, W, 3100 + t)[0] for t in range(T)])
    dains = torch.stack([synth.smooth_image(spec, 1, H, W, 3200 + t) for t in range(T)])
    _, _, fuses = G.chain(key, labels, dains, want_all=False)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    _, _, ofuses = generator_ref.autoregressive_segment(oracle(spec, sd), key, list(labels), list(dains))
    d = [float((fuses[t].cpu() - ofuses[t]).abs().max()) for t in range(T)]
    rep = {"workload": "512x512, 32-frame chain, batch 1", "fp32": {"max_abs_last_frame": d[-1], "max_abs_any_frame": max(d), "tolerance": NORTH_STAR_TOL}}
    Gb = rib.Generator(rib.hsm_gen_config(), compute_dtype="bf16").eval()
    Gb.load_state_dict(sd)
    _, _, fb = Gb.chain(key, labels, dains, want_all=False)
    db = [float((fb[t].cpu() - ofuses[t]).abs().max()) for t in range(T)]
    mb = [float((fb[t].cpu() - ofuses[t]).abs().mean()) for t in range(T)]
    rep["bf16"] = {"max_abs_last_frame": db[-1], "max_abs_any_frame": max(db), "mean_abs_last_frame": mb[-1], "mean_abs_worst_frame": max(mb),
                   "bound_max_abs": 2.0e-1, "bound_mean_abs": 1.3e-2}      # 1.5 x measured (1.30e-1 worst frame, 8.4e-3 worst mean)
    with open("gpurun_out/parity_config3_chain32_512.json", "w") as f:
        json.dump(rep, f, indent=1)
    # half storage (round 3): the same 16-bit kernels with 11 significant bits; the accurate 16-bit mode
    Gh = rib.Generator(rib.hsm_gen_config(), compute_dtype="f16").eval()
    Gh.load_state_dict(sd)
    _, _, fh = Gh.chain(key, labels, dains, want_all=False)
    dh = [float((fh[t].cpu() - ofuses[t]).abs().max()) for t in range(T)]
    mh = [float((fh[t].cpu() - ofuses[t]).abs().mean()) for t in range(T)]
    rep["f16"] = {"max_abs_last_frame": dh[-1], "max_abs_any_frame": max(dh), "mean_abs_last_frame": mh[-1], "mean_abs_worst_frame": max(mh),
                  "bound_max_abs": F16_MAX, "bound_mean_abs": F16_MEAN}
    with open("gpurun_out/parity_config3_chain32_512.json", "w") as f:
        json.dump(rep, f, indent=1)
    del Gh
    assert bool(torch.isfinite(fh).all())
    assert max(dh) <= F16_MAX and max(mh) <= F16_MEAN, rep["f16"]
    assert max(d) <= NORTH_STAR_TOL, rep["fp32"]
    assert max(db) <= 2.0e-1 and max(mb) <= 1.3e-2, rep["bf16"]
    # no build-up through the recurrence: the last frame is no worse than the worst one of the first four
    assert db[-1] <= 1.5 * max(db[:4]) and mb[-1] <= 1.5 * max(mb[:4]), (db[:4], db[-1], mb[:4], mb[-1])
    del Gb


def test_config5_1024_batch4_matches_the_batch1_run():
    """BASELINE configs[4]: 1024x1024, batch 4.  The B=4 plan (its own tuned tile choices, 4x workspace, blockIdx.z
    ranges) against the B=1 plan sample by sample - which test_full_1024_against_oracle pins to the oracle at every
    pixel - to fp32 summation-order tolerance; plus batch independence and determinism at this size."""
    spec, sd, G = build("full", 0)
    label, fake, prev = synth.make_inputs(spec, 4, 1024, 1024, 77)        # sample 0 == the oracle test's inputs
    i4, m4 = [t.clone() for t in G(label, None, fake, prev)]
    j4, n4 = G(label, None, fake, prev)
    assert torch.equal(i4, j4) and torch.equal(m4, n4)
    rep = {}
    for b in (0, 3):
        i1, m1 = G(label[b:b + 1], None, fake[b:b + 1], prev[b:b + 1])
        rep[b] = (float((i1 - i4[b:b + 1]).abs().max()), float((m1 - m4[b:b + 1]).abs().max()))
    with open("gpurun_out/parity_config5_1024_b4.json", "w") as f:
        json.dump({"max_abs_vs_batch1 (img, mask)": rep, "tolerance": 1e-4}, f)
    # (each plan is within 4e-5 of the oracle at this size - test_full_1024_against_oracle - with its own tile choices and
    # Winograd GEMM shapes, so the two may differ by up to twice that)
    assert all(v[0] <= 1e-4 and v[1] <= 1e-4 for v in rep.values()), rep
    assert float(i4.abs().max()) <= 1.0 and 0.0 <= float(m4.min()) and float(m4.max()) <= 1.0
    G._ws.clear()
    # SURVEY 8(d) config 5 is "fp32 AND bf16": the same batch through the bf16 storage mode (its own plan: 64-channel
    # chunks, its own tuning table) against the fp32 run above, which is oracle-pinned; samples 0 and 3; the mode's own
    # bound (1.5 x measured, as in test_bf16_storage_mode_error_is_bounded)
    Gb = rib.Generator(rib.hsm_gen_config(), compute_dtype="bf16").eval()
    Gb.load_state_dict(sd)
    ib, mb = Gb(label, None, fake, prev)
    brep = {}
    for b in (0, 3):
        brep[b] = {"max_abs_img": float((ib[b] - i4[b]).abs().max()), "mean_abs_img": float((ib[b] - i4[b]).abs().mean()),
                   "max_abs_mask": float((mb[b] - m4[b]).abs().max()), "mean_abs_mask": float((mb[b] - m4[b]).abs().mean())}
    with open("gpurun_out/parity_config5_1024_b4.json", "w") as f:
        json.dump({"max_abs_vs_batch1 (img, mask)": rep, "tolerance": 1e-4,
                   "bf16_vs_fp32_batch4": brep, "bf16_bounds": {"max_abs_img": 2.5e-1, "mean_abs_img": 1.5e-2, "max_abs_mask": 5.5e-2, "mean_abs_mask": 6.2e-3}}, f)
    assert bool(torch.isfinite(ib).all()) and bool(torch.isfinite(mb).all())
    # measured (round 3): img 1.64e-1 / 1.49e-1 max, 1.01e-2 / 1.00e-2 mean; mask 3.6e-2 / 3.4e-2 max, 4.1e-3 mean (4 M pixels per sample:
    # the maximum over 16x the pixels of a 256x256 frame sits further out in the same error distribution)
    assert all(v["max_abs_img"] <= 2.5e-1 and v["mean_abs_img"] <= 1.5e-2 and v["max_abs_mask"] <= 5.5e-2 and v["mean_abs_mask"] <= 6.2e-3 for v in brep.values()), brep
    del Gb
    torch.cuda.empty_cache()


def test_driver_lanes_follow_a_weight_reload(tmp_path):
    """Evaluator keeps clones of the generator for its lanes; after load_state_dict on the model the clones must get
    the new weights too (they used to keep the old blob: segments alternated between old and new weights)."""
    from PIL import Image
    from render_in_between_amd import evaluator as ev
    from tests.test_driver import _write_example
    root = str(tmp_path)
    n = _write_example(root, n_key=4, rate=2, H=32, W=48)        # 3 independent segments -> 3 lanes
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    spec = rib.GenSpec.from_cfg(cfg.gen)
    dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
    G = rib.Generator(rib.hsm_gen_config()).eval()
    G.load_state_dict(synth.make_state_dict(spec, 0))
    E = ev.Evaluator(cfg, lanes=3, batch=1)
    a = E.evaluate_from_folder(G, *dirs, os.path.join(root, "a"))
    v0 = G.weights_version
    G.load_state_dict(synth.make_state_dict(spec, 5))             # new checkpoint into the same object
    assert G.weights_version == v0 + 1
    b = E.evaluate_from_folder(G, *dirs, os.path.join(root, "b"))
    G2 = rib.Generator(rib.hsm_gen_config()).eval()
    G2.load_state_dict(synth.make_state_dict(spec, 5))
    c = ev.Evaluator(cfg, lanes=1, batch=1).evaluate_from_folder(G2, *dirs, os.path.join(root, "c"))
    assert len(a) == len(b) == len(c) == n
    differs = 0
    for fa, fb, fc in zip(a, b, c):
        xb = np.asarray(Image.open(fb))
        assert np.array_equal(xb, np.asarray(Image.open(fc))), fb      # every segment rendered with the NEW weights
        differs += not np.array_equal(xb, np.asarray(Image.open(fa)))
    assert differs >= 3                                                # and the new weights do change the generated frames


@pytest.mark.parametrize("wino_m", [4, 2, 0])
def test_winograd_path_agrees_with_the_direct_convolutions(monkeypatch, wino_m):
    """The deep 3x3 layers (>= 256 input channels, maps <= 128x128) run in the Winograd domain: k_wino(4)_in, one 36-way
    (F(4x4, 3x3), RIB_WINO_M=4) or 16-way (F(2x2, 3x3), RIB_WINO_M=2) batched 1x1 k_igemm, k_wino(4)_out with the fused
    epilogue (residual, activation, statistics); by default the plan picks F(4x4) for maps of >= 256 4x4 tiles and F(2x2)
    below.  With RIB_NO_WINO they run as direct implicit GEMMs.  Same frame either
    way to fp32 summation-order tolerance - at map sizes that leave partial output tiles too (a 3x5 / 11x7 deepest
    map), with a batch, and per tap against the oracle."""
    spec, sd, _ = build("full", 0)
    R = oracle(spec, sd)
    gname = (".wino4",) if wino_m == 4 else (".wino",) if wino_m == 2 else (".wino", ".wino4")      # 0: the plan's own choice per layer
    for (B, H, W, seed) in ((1, 64, 64, 1), (2, 48, 80, 2), (1, 256, 256, 3), (1, 176, 112, 4)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, seed)
        monkeypatch.delenv("RIB_NO_WINO", raising=False)
        if wino_m:
            monkeypatch.setenv("RIB_WINO_M", str(wino_m))
        G1 = rib.Generator(rib.hsm_gen_config()).eval(); G1.load_state_dict(sd)
        names = [o["name"] for o in G1.launch_info(B, H, W)]
        assert any(n.endswith(gname) for n in names) and any(n.endswith(".wino_in") for n in names), (B, H, W)
        if (H, W) == (64, 64):
            G1.enable_taps()
        i1, m1 = [t.clone() for t in G1(label, None, fake, prev)]
        if (H, W) == (64, 64):
            taps = G1.read_taps(B, H, W)
            otaps = {}
            R(label, None, fake, prev, taps=otaps)
            for k, v in taps.items():
                assert float((v - otaps[k]).abs().max()) / max(1.0, float(otaps[k].abs().max())) <= TOL, k
        monkeypatch.setenv("RIB_NO_WINO", "1")
        G2 = rib.Generator(rib.hsm_gen_config()).eval(); G2.load_state_dict(sd)
        assert not any(".wino" in o["name"] for o in G2.launch_info(B, H, W))
        i2, m2 = G2(label, None, fake, prev)
        torch.cuda.synchronize()
        e = max(float((i1 - i2).abs().max()), float((m1 - m2).abs().max()))
        oi, om = R(label, None, fake, prev)
        print("winograd m=%d %s: vs direct %.2e, img vs oracle %.2e (direct %.2e)" % (wino_m, (B, H, W), e, float((i1.cpu() - oi).abs().max()),
                                                                                    float((i2.cpu() - oi).abs().max())))
        assert e < (5e-5 if wino_m == 2 else 1e-4), (B, H, W, e)
        assert float((i1.cpu() - oi).abs().max()) <= TOL and float((m1.cpu() - om).abs().max()) <= TOL, (B, H, W)
        del G1, G2
    monkeypatch.delenv("RIB_NO_WINO", raising=False)
    monkeypatch.delenv("RIB_WINO_M", raising=False)


def test_first_layers_read_the_callers_tensors_in_place(monkeypatch):
    """The four first-layer 3x3 convolutions (ref_embedding.conv_first 6 -> 64, flow_network_temp.down_img.0 9 -> 32 /
    down_lbl.0 22 -> 32, down_first 22 -> 16) run as k_conv_lowc: the halo tile gathered from the caller's NCHW tensors,
    K over the real channels, no pack launches, conv_img without its NHWC copy.  Against the same frame with
    RIB_NO_LOWC=1 (pack + k_igemm over zero-padded channels) and against the oracle, per tap at 64x64; sizes whose
    width is not a multiple of the 32-pixel tile, a batch, and a 5-step chain (the labels-only batch plan uses the
    same kernel)."""
    spec, sd, _ = build("full", 0)
    R = oracle(spec, sd)
    for (B, H, W, seed) in ((1, 64, 64, 11), (2, 48, 80, 12), (1, 176, 112, 13), (1, 256, 256, 14)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, seed)
        monkeypatch.delenv("RIB_NO_LOWC", raising=False)
        G1 = rib.Generator(rib.hsm_gen_config()).eval(); G1.load_state_dict(sd)
        info = G1.launch_info(B, H, W)
        names = [o["name"] for o in info]
        assert not any(n.startswith("pack.") for n in names), names[:6]
        lowc = [o["name"] for o in info if o["tile"].startswith("lowc")]
        assert sorted(lowc) == sorted(["ref_embedding.conv_first", "flow_network_temp.down_img.0", "flow_network_temp.down_lbl.0", "down_first"])
        if (H, W) == (64, 64):
            G1.enable_taps()
        i1, m1 = [t.clone() for t in G1(label, None, fake, prev)]
        if (H, W) == (64, 64):
            taps = G1.read_taps(B, H, W)
            otaps = {}
            R(label, None, fake, prev, taps=otaps)
            for k in ("cond_0", "down_first", "mask.lbl_0.raw", "mask.img_0.raw", "mask.cat.raw"):
                assert float((taps[k] - otaps[k]).abs().max()) / max(1.0, float(otaps[k].abs().max())) <= 2e-5, k
        monkeypatch.setenv("RIB_NO_LOWC", "1")
        G2 = rib.Generator(rib.hsm_gen_config()).eval(); G2.load_state_dict(sd)
        n2 = [o["name"] for o in G2.launch_info(B, H, W)]
        assert "pack.label" in n2 and "pack.img9" in n2 and "pack.embed_in" in n2
        i2, m2 = G2(label, None, fake, prev)
        torch.cuda.synchronize()
        e = max(float((i1 - i2).abs().max()), float((m1 - m2).abs().max()))
        assert e < 5e-5, (B, H, W, e)
        oi, om = R(label, None, fake, prev)
        assert float((i1.cpu() - oi).abs().max()) <= TOL and float((m1.cpu() - om).abs().max()) <= TOL, (B, H, W)
        del G2
        for tw in ("16", "32"):                      # both tile shapes of every instantiation (the plan mixes them by layer)
            monkeypatch.delenv("RIB_NO_LOWC", raising=False)
            monkeypatch.setenv("RIB_LOWC_TW", tw)
            G3 = rib.Generator(rib.hsm_gen_config()).eval(); G3.load_state_dict(sd)
            assert all(("8x%s tile" % tw) in o["tile"] for o in G3.launch_info(B, H, W) if o["tile"].startswith("lowc"))
            i3, m3 = G3(label, None, fake, prev)
            torch.cuda.synchronize()
            assert max(float((i1 - i3).abs().max()), float((m1 - m3).abs().max())) < 5e-5, (B, H, W, tw)
            del G3
        monkeypatch.delenv("RIB_LOWC_TW", raising=False)
        if (H, W) == (48, 80):
            # chain: frame t of the segment == the per-frame call on (label_t, dain_t, fused_{t-1})
            monkeypatch.delenv("RIB_NO_LOWC", raising=False)
            T = 5
            labels = torch.stack([synth.make_inputs(spec, B, H, W, 100 + t)[0] for t in range(T)])
            dains = torch.stack([synth.make_inputs(spec, B, H, W, 100 + t)[1] for t in range(T)])
            key = prev.cuda()
            _, _, fused = G1.chain(key, labels.cuda(), dains.cuda())
            p = key
            for t in range(T):
                _, _, f = G1.forward_blend(labels[t].cuda(), None, dains[t].cuda(), p)
                assert torch.equal(f, fused[t]), t
                p = f.clone()
        del G1
    monkeypatch.delenv("RIB_NO_LOWC", raising=False)


def test_dma_staged_kernel_variants_agree_with_the_register_staged_ones():
    """Round 3: k_igemm instantiations whose operand tiles are staged by LDS-DMA (global_load_lds_dwordx4; filters always,
    the input tile where no prologue transforms it: unpadded XOR-swizzled LDS rows, zero padding read from a page of zeros,
    stride-2 de-interleaving done by the source addresses) exist beside the register-staged ones and are picked by
    measured choices only.  Here every convolution / fused-SPADE launch that admits one is pinned to its first fitting DMA
    variant: same frame (summation order differs with the tile geometry only), odd sizes and batch included."""
    import ctypes as C
    from render_in_between_amd import _native
    spec, sd, G0 = build("full", 0)
    lib = _native.lib()
    g12 = (C.c_int * 12)()
    kinds = {}
    for i in range(lib.rib_num_variants()):
        if lib.rib_variant_info(i, g12) == 0 and g12[11] in (100, 109):      # 100: one slice per barrier; 109: a whole chunk (tile + nine slices) per barrier
            kinds.setdefault(g12[11], []).append(i)
    assert len(kinds[100]) >= 8 and len(kinds[109]) >= 3      # (round 4 pruned the table to the entries some launch plan uses: 10 + 4)
    for (B, H, W), prefer in (((1, 128, 128), 109), ((2, 48, 80), 100), ((1, 256, 256), 109), ((1, 128, 128), 100)):
        dma_idx = kinds[prefer] + kinds[209 - prefer]
        label, fake, prev = synth.make_inputs(spec, B, H, W, 31)
        i0, m0 = [t.clone() for t in G0(label, None, fake, prev)]
        G1 = rib.Generator(rib.hsm_gen_config(), use_tuning=False).eval()
        G1.load_state_dict(sd)
        pinned = {}
        for info in G1.launch_info(B, H, W):
            if info["class"] not in (0, 1) or "tile " not in info["tile"] or "gemm" in info["tile"] or "wino" in info["tile"]:
                continue
            for vi in dma_idx:
                lib.rib_set_choice(G1._h, B, H, W, info["name"].encode(), vi, 1)
                if lib.rib_workspace_bytes(G1._h, B, H, W) > 0:
                    pinned[info["name"]] = vi
                    break
                lib.rib_set_choice(G1._h, B, H, W, info["name"].encode(), -1, 1)
        assert len(pinned) >= 15, (B, H, W, len(pinned))      # (the fused-shortcut, upsample, first-layer and head launches have no DMA twin)
        G1._ws.clear()
        i1, m1 = G1(label, None, fake, prev)
        staged = [x["tile"] for x in G1.launch_info(B, H, W) if x["name"] in pinned]
        assert all(("tb100" in t or "tb109" in t) for t in staged) and any("tb%d" % prefer in t for t in staged), staged[:3]
        d = (float((i1 - i0).abs().max()), float((m1 - m0).abs().max()))
        assert d[0] <= 5e-5 and d[1] <= 5e-5, (B, H, W, d, len(pinned))
        oi, om = oracle(spec, sd)(label, None, fake, prev)
        assert float((i1.cpu() - oi).abs().max()) <= TOL and float((m1.cpu() - om).abs().max()) <= TOL
        del G1


def test_every_gemm_dma_tile_computes_the_same_frame():
    """k_gemm_dma (the Winograd position GEMMs and the condition-level gamma/beta GEMMs) has five tiles, 64x64 ... 128x128.
    Every GEMM launch of the plan is pinned to each tile in turn: same frame as the default plan, odd sizes (ragged M)
    included.  (Round 3 also measured a three-stage LDS ring for the small tiles and an XCD-aware workgroup order: neither
    moved the GEMMs - DESIGN "Round 3" - and neither is in the library.)"""
    import ctypes as C
    from render_in_between_amd import _native
    spec, sd, G0 = build("full", 0)
    lib = _native.lib()
    g12 = (C.c_int * 12)()
    tiles = [i for i in range(lib.rib_num_variants()) if lib.rib_variant_info(i, g12) == 0 and g12[0] == 0]      # FRW = 0: a k_gemm_dma tile
    assert len(tiles) == 5, tiles
    for (B, H, W) in ((1, 256, 256), (2, 80, 112)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, 37)
        i0, m0 = [t.clone() for t in G0(label, None, fake, prev)]
        names = [x["name"] for x in G0.launch_info(B, H, W) if "gemm (LDS-DMA" in x["tile"]]
        assert len(names) >= 10, names
        seen = set()
        for vi in tiles:
            G1 = rib.Generator(rib.hsm_gen_config(), use_tuning=False).eval()
            G1.load_state_dict(sd)
            for n in names:
                lib.rib_set_choice(G1._h, B, H, W, n.encode(), vi, 1)
            assert lib.rib_workspace_bytes(G1._h, B, H, W) > 0, vi
            seen.update(x["tile"].split(",")[0] for x in G1.launch_info(B, H, W) if x["name"] in names)
            i1, m1 = G1(label, None, fake, prev)
            d = (float((i1 - i0).abs().max()), float((m1 - m0).abs().max()))
            assert d[0] <= 5e-5 and d[1] <= 5e-5, (B, H, W, vi, d)
            del G1
        assert len(seen) == 5, seen


def test_split_bf16_products_are_opt_in_and_fp32_grade():
    """rib_set_products(RIB_PRODUCTS_BF16X3) (round 6): the k_gemm_dma launches form each fp32 product from six bf16 matrix-core
    products of three-way split operands.  Off by default; when on, the frame differs from the default's by fp32 rounding
    noise only and holds the same 2e-4 against the oracle (tools/products_error.py ranks the two against an fp64 oracle)."""
    spec, sd, G0 = build("full", 0)
    assert G0.products == "f32"
    G1 = rib.Generator(rib.hsm_gen_config(), products="bf16x3").eval()
    G1.load_state_dict(sd)
    for (B, H, W) in ((1, 256, 256), (2, 80, 112)):
        label, fake, prev = synth.make_inputs(spec, B, H, W, 41)
        assert any("gemm (LDS-DMA" in x["tile"] for x in G1.launch_info(B, H, W))
        i0, m0 = [t.clone() for t in G0(label, None, fake, prev)]
        i1, m1 = G1(label, None, fake, prev)
        assert not torch.equal(i0, i1)                     # other instructions ran ...
        d = (float((i1 - i0).abs().max()), float((m1 - m0).abs().max()))
        assert d[0] <= 5e-5 and d[1] <= 5e-5, (B, H, W, d)     # ... and moved nothing beyond fp32 rounding noise
        oi, om = oracle(spec, sd)(label, None, fake, prev)
        assert float((i1.cpu() - oi).abs().max()) <= TOL and float((m1.cpu() - om).abs().max()) <= TOL
    i2, _ = G1(label, None, fake, prev)
    assert torch.equal(i1, i2)                             # deterministic
    # the setting belongs to the handle, not to a plan: an autoregressive segment (rib_chain, launch by launch and replayed as
    # a captured graph) renders what the per-frame loop renders, and switching it drops the captured segments
    T, B, H, W = 3, 2, 64, 64
    frames = [synth.make_inputs(spec, B, H, W, 60 + t) for t in range(T)]
    labels = torch.stack([f[0] for f in frames]).cuda(); dains = torch.stack([f[1] for f in frames]).cuda(); key = frames[0][2].cuda()
    p_, want = key, []
    for t in range(T):
        p_ = G1.forward_blend(labels[t], None, dains[t], p_)[2].clone()
        want.append(p_)
    got = G1.chain(key, labels, dains, want_all=True)[2]
    assert all(torch.equal(got[t], want[t]) for t in range(T))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        G1.set_graph_replay(True)
        for k in range(4):           # (Generator.chain alternates two output sets while the option is on: two captures, then replays)
            a = G1.chain(key, labels, dains, want_all=True)[2].clone()
            st.synchronize()
            assert torch.equal(a, got), k
        assert G1.graph_stats()["replays"] >= 1
        from render_in_between_amd import _native
        _native.check(G1._h, _native.lib().rib_set_products(G1._h, 0))   # back to exact fp32: the captured segment must not survive
        c = G1.chain(key, labels, dains, want_all=True)[2].clone()
        st.synchronize()
        _native.check(G1._h, _native.lib().rib_set_products(G1._h, 1))
        G1.set_graph_replay(False)
    assert not torch.equal(c, got) and float((c - got).abs().max()) <= 5e-5
    with pytest.raises(ValueError):
        rib.Generator(rib.hsm_gen_config(), compute_dtype="bf16", products="bf16x3")


def test_sixteen_channel_spade_layout_agrees_with_the_pair_layout(monkeypatch):
    """down_0.1 / up_0.1 modulate 16 channels: by default one [gamma(16) | beta(16)] MFMA fragment per wave (k_igemm<SPADE,
    NF = 1>, the halves exchange rows with shuffles); with RIB_NO_SPADE16 the pair layout every other SPADE uses.  Same
    frame, and the modulated tensors (taps `.y1`) agree with the oracle's."""
    spec, sd, _ = build("full", 0)
    R = oracle(spec, sd)
    for (B, H, W, seed) in ((1, 96, 96, 21), (2, 48, 80, 22), (1, 176, 112, 23)):      # (maps of <= 4096 pixels run GEMM + modulate instead)
        label, fake, prev = synth.make_inputs(spec, B, H, W, seed)
        monkeypatch.delenv("RIB_NO_SPADE16", raising=False)
        G1 = rib.Generator(rib.hsm_gen_config()).eval(); G1.load_state_dict(sd)
        one = [o for o in G1.launch_info(B, H, W) if o["name"] in ("down_0.1.spade", "up_0.1.spade")]
        assert len(one) == 2 and all("BN 32" in o["tile"] for o in one), one
        if (H, W) == (96, 96):
            G1.enable_taps()
        i1, m1 = [t.clone() for t in G1(label, None, fake, prev)]
        if (H, W) == (96, 96):
            taps = G1.read_taps(B, H, W)
            otaps = {}
            R(label, None, fake, prev, taps=otaps)
            for k in ("down_0.y1", "up_0.y1", "down_0", "up_0"):
                assert float((taps[k] - otaps[k]).abs().max()) / max(1.0, float(otaps[k].abs().max())) <= 2e-5, k
        monkeypatch.setenv("RIB_NO_SPADE16", "1")
        G2 = rib.Generator(rib.hsm_gen_config()).eval(); G2.load_state_dict(sd)
        assert all("BN 32" not in o["tile"] for o in G2.launch_info(B, H, W) if o["name"] in ("down_0.1.spade", "up_0.1.spade"))
        i2, m2 = G2(label, None, fake, prev)
        torch.cuda.synchronize()
        assert max(float((i1 - i2).abs().max()), float((m1 - m2).abs().max())) < 2e-5, (B, H, W)
        del G1, G2
    monkeypatch.delenv("RIB_NO_SPADE16", raising=False)
