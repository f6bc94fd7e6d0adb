"""tools/verify_checkpoint.py (the real netG_epoch006.pth is not in the tree): the one-command day-one
check, exercised on seed-defined checkpoints of the reference's exact key set - a good file, one that a DataParallel run
saved, and three broken ones.  CPU steps only here (load, fold, and - in the build container - the reference's own outputs)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import render_in_between_amd as rib                    # noqa: E402
from render_in_between_amd import synth                # noqa: E402
from tools import verify_checkpoint as vc              # noqa: E402


def _run(path, tmp_path, capsys):
    rc = vc.main([path, "--out", str(tmp_path / "gold"), "--sizes", "64"])
    return rc, json.loads(capsys.readouterr().out)


def test_day_one_check_on_good_and_broken_checkpoints(tmp_path, capsys):
    spec = rib.GenSpec.from_cfg(rib.hsm_gen_config())
    sd = synth.make_state_dict(spec, 0)
    good = str(tmp_path / "netG.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, good)      # as utils.py:115-116 / 101-105 expect to find it
    rc, rep = _run(good, tmp_path, capsys)
    assert rc == 0 and rep["ok"] and rep["load"]["tensors"] == 372 and rep["fold"]["ok"] and rep["fold"]["fits_ieee_half"]
    assert rep["fold"]["max_rel_diff_vs_oracle_fold"] <= 1e-6 and rep["fold"]["convolutions"] == 62
    if os.path.isdir("/root/reference"):          # the build container: the reference itself ran on the file
        assert rep["goldens"]["ok"] and rep["goldens"]["cases"]["64"]["oracle_vs_reference"]["img"] <= 1e-4
        assert os.path.exists(str(tmp_path / "gold" / "real_64.npz"))
    else:
        assert "skipped" in rep["goldens"]
    assert "skipped" in rep["gpu"] or rep["gpu"]["ok"]
    # a missing tensor, an extra one, a wrong shape, a NaN: each is a hard failure with a message that names it
    for name, edit in (("missing", lambda d: d.pop("down_first.layers.conv.bias")),
                       ("extra", lambda d: d.__setitem__("net_D.weight", torch.zeros(1))),
                       ("shape", lambda d: d.__setitem__("conv_img.layers.conv.bias", torch.zeros(4))),
                       ("nan", lambda d: d.__setitem__("conv_img.layers.conv.bias", torch.full((3,), float("nan"))))):
        bad = dict(sd)
        edit(bad)
        path = str(tmp_path / (name + ".pth"))
        torch.save(bad, path)
        rc, rep = _run(path, tmp_path, capsys)
        assert rc == 1 and not rep["ok"] and not rep["load"]["ok"], name
        assert rep["load"]["missing"] or rep["load"]["unexpected"] or rep["load"]["shape_mismatches"] or rep["load"]["non_finite_tensors"], name
    rc, rep = _run(str(tmp_path / "nowhere.pth"), tmp_path, capsys)
    assert rc == 1 and "No checkpoint found" in rep["load"]["error"]                 # utils.py:113
