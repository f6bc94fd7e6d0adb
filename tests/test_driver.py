"""Host logic of the sequence driver (no GPU): directory contract, sample-rate inference, key-frame
pass-through, segment splitting, PNG naming and quantisation.  The generator is replaced by the
CPU oracle standing in behind the reference's call protocol (allowed: tests may use the oracle)."""
import json
import os

import numpy as np
import pytest
import torch

import render_in_between_amd as rib
from render_in_between_amd import evaluator as ev, rasterise, synth
from oracle import generator_ref, rasterise_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MID_CFG = dict(num_filters=16, max_num_filters=64, mask=dict(num_filters=32, max_num_filters=64),
               embed=dict(num_filters=32, max_num_filters=64))


def _write_example(root, n_key=2, rate=2, H=32, W=48, clip="clipA", seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    n = (n_key - 1) * rate + 1
    for d in ("inputs", "DAIN", "Predict_motion"):
        os.makedirs(os.path.join(root, d, clip))
    for k in range(n_key):
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "inputs", clip, "%04d.png" % k))
    for i in range(n):
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(os.path.join(root, "DAIN", clip, "f%03d.png" % i))
        body = []
        for j in range(25):
            body += [float(rng.uniform(4, W - 4)), float(rng.uniform(4, H - 4)), 0.9]
        hand = [10.0, 10.0, 0.9] * 21
        with open(os.path.join(root, "Predict_motion", clip, "f%03d_keypoints.json" % i), "w") as f:
            json.dump({"people": [{"pose_keypoints_2d": body, "hand_left_keypoints_2d": hand, "hand_right_keypoints_2d": hand}]}, f)
    return n


def test_sample_rate_and_segments():
    assert ev.sample_rate_of(9, 3) == 4 and ev.sample_rate_of(3, 2) == 2 and ev.sample_rate_of(65, 3) == 32
    keys, segs = ev.split_segments(9, 4)
    assert keys == [0, 4, 8] and segs == [(0, [1, 2, 3]), (4, [5, 6, 7])]
    assert ev.sample_rate_of(9, 3) == generator_ref.sample_rate_of(9, 3)


def oracle_labels(frames, H, W, sigma=5, t1=0.001, t2=0.001):
    """label_fn for Evaluator built on the CPU oracle (tests only)."""
    out = []
    for lm, conf in frames:
        sk = rasterise_ref.skeleton_image(lm, conf, H, W, t1, t2)
        pm = rasterise_ref.pose_map(lm, conf, H, W, sigma, t1)
        sk_t = torch.from_numpy((sk.astype(np.float32) / 255.0 - 0.5) / 0.5).permute(2, 0, 1)
        out.append(torch.cat([sk_t, torch.from_numpy(pm)], dim=0))
    return torch.stack(out).contiguous()


def stroke_points(st):
    """numpy statement of what k_skeleton does with one rib_stroke (csrc/raster.hip.h)."""
    n = int(st["n"])
    if n == 0:
        return None, None
    lin = np.arange(n) * float(st["step"]) + float(st["start"])
    if n > 1:
        lin[-1] = float(st["stop"])
    u = lin.astype(int); v = (float(st["a"]) * lin + float(st["b"])).astype(int)
    return (v, u) if st["swap"] else (u, v)


def test_host_tables_match_the_reference_rules():
    # json reader against the reference's own output (golden), incl. two people and nobody
    for n in "abcde":
        g = np.load(os.path.join(GOLD, "raster_%s.npz" % n))
        assert np.array_equal(rasterise.read_json_keypoint(os.path.join(GOLD, "raster_json", "pose_%s.json" % n)), g["keypoints"])
    # gaussian kernel == the one scipy builds (response of gaussian_filter1d to a delta, interior)
    from scipy import ndimage
    w, r = rasterise.gaussian_weights(5)
    d = np.zeros(101); d[50] = 1
    assert r == 20 and np.array_equal(ndimage.gaussian_filter1d(d, 5)[50:71], w)
    # limb lines: curve samples == interpPoints (oracle restatement, pinned to the reference) for
    # OpenPose-like fractional joints AND for integer-valued joints (knife-edge truncations)
    rng = np.random.default_rng(3)
    for it in range(400):
        x = np.round(rng.uniform(1, 255, 2), 3); y = np.round(rng.uniform(1, 255, 2), 3)
        if it % 2:
            x, y = np.round(x), np.round(y)
        if it % 7 == 0:
            y[1] = y[0]
        if it % 11 == 0:
            x[1] = x[0]
        pts = np.zeros((2, 2)); pts[:, 0] = x; pts[:, 1] = y
        st = rasterise.stroke_table(pts, edges=[[0, 1]])[0]
        gx, gy = stroke_points(st)
        rx, ry = rasterise_ref._interp_points(x, y)
        if rx is None or not rx.size:
            assert gx is None
        else:
            assert np.array_equal(gx, rx) and np.array_equal(gy, ry), (x, y)
    # joints: thresholds, frame bounds, foot threshold, peaks
    lm = [(10.7, 5.2), (-1.0, 3.0), (47.9, 31.9), (48.0, 3.0), (5.0, 5.0)]
    conf = [0.9, 0.9, 0.9, 0.9, 0.0005]
    pts = rasterise.valid_points(lm, conf, 32, 48)
    assert np.array_equal(pts, [[10.7, 5.2], [0, 0], [47.9, 31.9], [0, 0], [0, 0]])
    assert np.array_equal(rasterise.peak_table(lm, conf, 32, 48), [[10, 5], [-1, -1], [47, 31], [-1, -1], [-1, -1]])
    assert rasterise.stroke_table(pts, edges=[[0, 1], [0, 2]])["n"].tolist() == [0, 37]    # off joint: limb skipped


def test_evaluate_from_folder_matches_oracle_loop(tmp_path):
    root = str(tmp_path)
    n = _write_example(root, n_key=3, rate=2)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    spec = rib.GenSpec.from_cfg(cfg.gen)
    R = generator_ref.RefGenerator(spec, synth.make_state_dict(spec, 2))
    calls = []

    class Model:                               # the reference's object protocol
        def eval(self):
            return self

        def __call__(self, label, label_prev, dain, prev):
            calls.append(label.shape)
            return R(label, label_prev, dain, prev)

    E = ev.Evaluator(cfg, label_fn=oracle_labels)
    out = os.path.join(root, "out", "Generated_frames")
    written = E.evaluate_from_folder(Model(), os.path.join(root, "inputs"), os.path.join(root, "DAIN"),
                                     os.path.join(root, "Predict_motion"), out)
    assert len(written) == n == 5 and len(calls) == 2            # 3 key frames pass through
    assert [os.path.basename(w) for w in written] == ["f%03d.png" % i for i in range(5)]
    from PIL import Image
    # key frame 2 is the resized ground-truth image, re-quantised
    key, _ = E.load_image(os.path.join(root, "inputs", "clipA", "0001.png"))
    assert np.array_equal(np.asarray(Image.open(written[2])), generator_ref.quantise_uint8(key.unsqueeze(0)))
    # generated frame 1 == oracle step from key frame 0
    k0, osz = E.load_image(os.path.join(root, "inputs", "clipA", "0000.png"))
    d1, _ = E.load_image(os.path.join(root, "DAIN", "clipA", "f001.png"))
    l1 = oracle_labels([E.load_pose(os.path.join(root, "Predict_motion", "clipA", "f001_keypoints.json"), osz)], 32, 48)[0]
    img, mask = R(l1.unsqueeze(0), None, d1.unsqueeze(0), k0.unsqueeze(0))
    want = generator_ref.quantise_uint8(generator_ref.blend(img, mask, d1.unsqueeze(0)))
    assert np.array_equal(np.asarray(Image.open(written[1])), want)
    assert l1.shape == (22, 32, 48) and float(l1[:3].min()) >= -1 and float(l1[3:].max()) <= 1
    # a model that only speaks the reference protocol and no label_fn: loud error, no host fallback
    with pytest.raises(RuntimeError, match="rasteriser"):
        ev.Evaluator(cfg).evaluate_from_folder(Model(), os.path.join(root, "inputs"), os.path.join(root, "DAIN"),
                                               os.path.join(root, "Predict_motion"), out + "2")


def test_gt_dir_and_keypoint_scaling_follow_the_reference(tmp_path):
    """evaluator.py:205-219: with a gt_dir the key frames / chain starts are gtlist[i]; the keypoints are resized
    together with the key (or gt) image, i.e. they scale by ITS size, not by the DAIN frame's; gen_vid is refused."""
    from PIL import Image
    root = str(tmp_path)
    n = _write_example(root, n_key=2, rate=2, H=32, W=48)
    rng = np.random.default_rng(9)
    os.makedirs(os.path.join(root, "gt", "clipA"))
    for i in range(n):                      # ground-truth frames at TWICE the model size: joints must halve
        Image.fromarray(rng.integers(0, 255, (64, 96, 3), dtype=np.uint8)).save(os.path.join(root, "gt", "clipA", "g%03d.png" % i))
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    spec = rib.GenSpec.from_cfg(cfg.gen)
    R = generator_ref.RefGenerator(spec, synth.make_state_dict(spec, 2))
    seen = []

    class Model:
        def eval(self):
            return self

        def __call__(self, label, label_prev, dain, prev):
            seen.append((label.clone(), prev.clone()))
            return R(label, label_prev, dain, prev)

    E = ev.Evaluator(cfg, label_fn=oracle_labels)
    dirs = [os.path.join(root, d) for d in ("inputs", "DAIN", "Predict_motion")]
    written = E.evaluate_from_folder(Model(), *dirs, os.path.join(root, "o"), gt_dir=os.path.join(root, "gt"))
    assert len(written) == n == 3 and len(seen) == 1
    g0, size0 = E.load_image(os.path.join(root, "gt", "clipA", "g000.png"))
    assert size0 == (96, 64)
    assert torch.equal(seen[0][1], g0.unsqueeze(0))                       # the chain starts from gtlist[0], not inputs/0000.png
    want = oracle_labels([E.load_pose(os.path.join(root, "Predict_motion", "clipA", "f001_keypoints.json"), (96, 64))], 32, 48)
    assert torch.equal(seen[0][0], want)                                  # joints scaled by the gt image's size (x0.5)
    assert np.array_equal(np.asarray(Image.open(written[2])),
                          generator_ref.quantise_uint8(E.load_image(os.path.join(root, "gt", "clipA", "g002.png"))[0].unsqueeze(0)))
    with pytest.raises(NotImplementedError):
        E.evaluate_from_folder(Model(), *dirs, os.path.join(root, "o2"), gen_vid=True)


def test_inference_cli_surface():
    import importlib.util
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "render-in-between_amd", "inference.py")
    src = open(p).read()
    for flag in ("--config", "--save-dir", "--input-dir", "--seed"):
        assert flag in src
    cfg = rib.get_config(os.path.join(os.path.dirname(p), "configs", "HSM.yaml"))
    assert cfg.model_pretrain_G.endswith("netG_epoch006.pth") and cfg.model_width == 480 and cfg.model_height == 320
    assert rib.GenSpec.from_cfg(cfg.gen) == rib.GenSpec.from_cfg(rib.hsm_gen_config())
    spec = importlib.util.spec_from_file_location("rib_inference", p)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    with pytest.raises((ValueError, RuntimeError)):     # missing checkpoint -> ValueError (no GPU here -> RuntimeError first)
        mod.load_generator(cfg)


def test_inference_cli_exposes_the_drivers_choices():
    """the driver's real trade-offs are flags of the CLI, not constructor arguments - PNG level (default: the
    reference's bytes), batch-invariant plans (default on), batch, dtype, ranks - and every rank ends with one summary line that
    names frames, rate, the launch thread's phases, the in-flight window, workers and the CPU budget."""
    import importlib.util
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "render-in-between_amd", "inference.py")
    src = open(p).read()
    for flag in ("--png-level", "--reproducible", "--batch", "--dtype", "--gpus"):
        assert flag in src, flag
    spec = importlib.util.spec_from_file_location("rib_inference2", p)
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    from render_in_between_amd.evaluator import Evaluator
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(), model_height=320, model_width=480)
    E = Evaluator(cfg)                                         # defaults: the reference's PNG bytes, batch-invariant plans
    assert E.reproducible and E.png_compress_level is None and E.default_batch() == 8
    E.timings = {"frames": 129, "wall": 0.6, "load": 0.02, "rasterise": 0.13, "generate": 0.01, "save": 0.43, "units": 8, "peak_units_in_flight": 8}
    line = mod.summary_line(E, 1, 2)
    for part in ("[rank 1/2]", "129 frames", "215.0 frames/s", "load 0.02 s", "save tail 0.43 s", "8 units, <= 8 in flight", "PNG level reference (zlib 6)", "batch 8", "batch-invariant plans"):
        assert part in line, (part, line)
    E2 = Evaluator(cfg, png_compress_level=1, reproducible=False, batch=2)
    E2.timings = {"frames": 0}
    assert "PNG level 1, batch 2, per-batch plans" in mod.summary_line(E2)


def test_cubic_resize_follows_the_opencv_definition():
    """The folder driver's resize (render-in-between_amd/resize.py, vectorised) against the scalar restatement of
    OpenCV's 8-bit INTER_CUBIC in oracle/resize_ref.py (written independently): bit-exact on enlargements, reductions,
    non-integer ratios, grey and RGB; plus properties of the definition: equal sizes copy, constants stay constant,
    overshoot saturates, no low-pass on reduction (period-3 stripes reduced 3x stay solid; PIL's BICUBIC averages them)."""
    from oracle import resize_ref
    from render_in_between_amd import resize as rz
    rng = np.random.default_rng(5)
    for (h0, w0, c, h, w) in ((9, 7, 3, 16, 16), (23, 31, 3, 16, 16), (12, 20, 1, 19, 33), (40, 40, 3, 16, 32), (5, 5, 3, 5, 9), (64, 48, 3, 32, 16)):
        a = rng.integers(0, 256, size=(h0, w0, c), dtype=np.uint8)
        if c == 1:
            a = a[:, :, 0]
        got = rz.resize_cubic_u8(a, w, h)
        want = resize_ref.resize_cubic_u8(a, w, h)
        assert got.shape == want.shape and got.dtype == np.uint8
        assert np.array_equal(got, want), (h0, w0, c, h, w, int(np.abs(got.astype(int) - want.astype(int)).max()))
    a = rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8)
    assert np.array_equal(rz.resize_cubic_u8(a, 16, 16), a)
    for v in (0, 1, 127, 254, 255):
        assert np.all(rz.resize_cubic_u8(np.full((11, 13, 3), v, np.uint8), 32, 16) == v)
    # overshoot is clipped, not wrapped: a hard edge enlarged 4x stays within [0, 255] and reaches both ends
    edge = np.zeros((8, 8), np.uint8); edge[:, 4:] = 255
    up = rz.resize_cubic_u8(edge, 32, 32)
    assert up.min() == 0 and up.max() == 255 and np.all(up[:, :12] == 0) and np.all(up[:, 20:] == 255)
    # reduction takes 4 taps at the source pitch (no area filter): reducing 3x lands on source pixel 3*dx + 1 exactly, so
    # stripes of period 3 come out solid; PIL's BICUBIC widens its kernel on reduction and returns their mean
    st = np.zeros((48, 48), np.uint8); st[:, 1::3] = 255
    down = rz.resize_cubic_u8(st, 16, 16)
    assert np.all(down == 255)
    from PIL import Image
    pil = np.asarray(Image.fromarray(st).resize((16, 16), Image.BICUBIC))
    assert 60 < pil.mean() < 110


def test_cubic_resize_agrees_with_atens_bicubic_to_one_grey_level():
    """A third-party anchor for the frame resize (cv2 itself is not in this image, so OpenCV's 8-bit INTER_CUBIC stays formally
    unpinned): ATen's `upsample_bicubic2d` implements the same definition in float - Keys weights with A = -0.75, half-pixel
    centres, replicated border, no low-pass on reduction (torch documents it as matching OpenCV's INTER_CUBIC) - so the 11-bit
    fixed-point restatement must land within ONE grey level of it on every pixel and be identical on most, on enlargements,
    reductions (1080p -> 512x512, 720p -> the reference's 320x480) and odd ratios."""
    from render_in_between_amd import resize as rz
    rng = np.random.default_rng(7)
    for (h0, w0, h, w) in ((1080, 1920, 512, 512), (720, 1280, 320, 480), (256, 256, 512, 512), (300, 500, 320, 480), (64, 48, 32, 16), (37, 53, 80, 96)):
        a = rng.integers(0, 256, size=(h0, w0, 3), dtype=np.uint8)
        got = rz.resize_cubic_u8(a, w, h).astype(int)
        t = torch.from_numpy(a.astype(np.float32)).permute(2, 0, 1)[None]
        ref = torch.nn.functional.interpolate(t, size=(h, w), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
        d = np.abs(got - np.clip(np.rint(ref), 0, 255).astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.1, (h0, w0, h, w, int(d.max()), float((d > 0).mean()))


def test_load_image_resizes_like_the_reference_transform(tmp_path):
    """`Evaluator.load_image` on a file that is not at the model size: the OpenCV-style cubic resize (default) equals the
    scalar restatement followed by ToTensor + Normalize(.5, .5); `resize="pil"` keeps PIL's bicubic; both report the
    original size (the keypoints are scaled by it)."""
    from PIL import Image
    from oracle import resize_ref
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, size=(45, 70, 3), dtype=np.uint8)
    path = str(tmp_path / "k.png")
    Image.fromarray(a).save(path)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=32, model_width=48, gauss_sigma=5, skeleton_thres=0.001, foot_thres=0.001)
    E = ev.Evaluator(cfg, label_fn=oracle_labels)
    t, size0 = E.load_image(path)
    assert size0 == (70, 45) and tuple(t.shape) == (3, 32, 48)
    want = resize_ref.resize_cubic_u8(a, 48, 32).astype(np.float32) / 255.0
    assert torch.equal(t, torch.from_numpy((want - 0.5) / 0.5).permute(2, 0, 1).contiguous())
    u8, _ = E.load_image_u8(path)
    assert np.array_equal(u8.numpy(), resize_ref.resize_cubic_u8(a, 48, 32))
    Ep = ev.Evaluator(cfg, label_fn=oracle_labels, resize="pil")
    tp, _ = Ep.load_image(path)
    pil = np.asarray(Image.fromarray(a).resize((48, 32), Image.BICUBIC), dtype=np.float32) / 255.0
    assert torch.equal(tp, torch.from_numpy((pil - 0.5) / 0.5).permute(2, 0, 1).contiguous())
    assert not torch.equal(t, tp)
    with pytest.raises(ValueError):
        ev.Evaluator(cfg, resize="area")


def test_group_segments_batches_equal_lengths_only():
    """Evaluator.group_segments: a chain of batch B advances all its samples together, so only segments with the same
    number of frames share one; order is kept, a group closes at `batch` members, leftovers form smaller groups."""
    segs = [(0, [1, 2, 3]), (4, [5, 6, 7]), (8, [9]), (10, [11, 12, 13]), (14, [15, 16, 17]), (18, [19, 20, 21])]
    assert ev.Evaluator.group_segments(segs, 2) == [[0, 1], [2], [3, 4], [5]]
    assert ev.Evaluator.group_segments(segs, 1) == [[0], [1], [2], [3], [4], [5]]
    assert ev.Evaluator.group_segments(segs, 8) == [[0, 1, 3, 4, 5], [2]]
    assert ev.Evaluator.group_segments([], 4) == []
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=320, model_width=480)
    assert ev.Evaluator(cfg).default_batch() == 8
    cfg.model_height = cfg.model_width = 512
    assert ev.Evaluator(cfg).default_batch() == 4


def test_io_worker_processes_return_what_the_threads_compute(tmp_path):
    """The file-side work of the native pipeline runs in forkserver worker processes (io_worker.py): same functions, same
    results as in-process calls - decoded frame, normalised key frame, rasteriser tables - and PNG bytes equal to PIL's own."""
    from PIL import Image
    from render_in_between_amd import io_worker
    root = str(tmp_path)
    _write_example(root, n_key=2, rate=2)
    dain = os.path.join(root, "DAIN", "clipA", "f001.png")
    key = os.path.join(root, "inputs", "clipA", "0000.png")
    pose = os.path.join(root, "Predict_motion", "clipA", "f001_keypoints.json")
    args = (dain, key, pose, True, True, 48, 32, "cv2", 0.001, 0.001)
    want = io_worker.load_frame(*args)
    pool = ev._process_pool(2)
    assert ev._process_pool(2) is pool                      # one pool per size and process
    got = pool.submit(io_worker.load_frame, *args).result()
    assert np.array_equal(got[0], want[0]) and got[0].dtype == np.uint8 and got[0].shape == (32, 48, 3)
    assert np.array_equal(got[1], want[1]) and got[1].dtype == np.float32 and got[1].shape == (3, 32, 48)
    assert all(np.array_equal(a, b) for a, b in zip(got[2], want[2]))
    E = ev.Evaluator(rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=32, model_width=48))
    assert torch.equal(E.load_image(key)[0], torch.from_numpy(want[1]))
    # the shared-memory variants the pipeline uses: pixels go through a named block, never through the result pipe
    blk = ev._shm_get(2 * 32 * 48 * 3)
    try:
        off = 32 * 48 * 3
        got = pool.submit(io_worker.load_frame_shm, blk.name, off, *args).result()
        assert got[0] is None and np.array_equal(blk.t.numpy()[off:2 * off].reshape(32, 48, 3), want[0])
        assert np.array_equal(io_worker.normalised_chw(got[1]), want[1])
        assert all(np.array_equal(a, b) for a, b in zip(got[2], want[2]))
        shm_png = os.path.join(root, "s.png")
        assert pool.submit(io_worker.save_png_shm, blk.name, off, 32, 48, shm_png, None).result() == shm_png
    finally:
        ev._shm_put(blk)
    assert ev._shm_get(2 * 32 * 48 * 3) is blk              # blocks are reused by size
    ev._shm_put(blk)
    out = os.path.join(root, "w.png")
    assert pool.submit(io_worker.save_png, want[0], out, None).result() == out
    assert open(out, "rb").read() == open(shm_png, "rb").read()
    ref = os.path.join(root, "r.png")
    Image.fromarray(want[0]).save(ref)
    assert open(out, "rb").read() == open(ref, "rb").read()


def test_cpu_budget_is_positive_and_bounded_by_the_affinity_mask():
    n = ev.cpu_budget()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    E = ev.Evaluator(rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=32, model_width=48))
    assert 1 <= E.io_threads <= max(1, n - 1) or n == 1
