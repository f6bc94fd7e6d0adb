#!/usr/bin/env python3
"""Drop-in for HMM/inference.py: same flags, same config keys, same directory contract
(<pose-dir>/<clip>/*.json  ->  <save-dir>/Predict_motion/<clip>/ and <save-dir>/Linear_motion/<clip>/).

    python render-in-between_amd/motion/inference.py --pose-dir ../example/input_pose
                    [--config configs/motion.yaml] [--save-dir ../example/test] [--upsample-rate 8]

It builds only what inference needs: the transformer and the positional encoding (no trainer,
optimiser, discriminator or AMASS h5 file; HMM/models/trainer.py:46-113).
"""
import argparse
import os
import random
import sys

import numpy as np
import torch
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))

import render_in_between_amd  # noqa: E402,F401
from render_in_between_amd.config import AttrDict  # noqa: E402
from render_in_between_amd.motion import model, pose_io  # noqa: E402
from render_in_between_amd.motion.spec import MotionSpec  # noqa: E402


def get_config(path):
    with open(path, "r") as stream:
        return AttrDict(yaml.load(stream, Loader=yaml.FullLoader))


def load_model(config, device=None):
    """trainer.pos_encode / trainer.transformer with the checkpoint (HMM/models/trainer.py:64-74)."""
    spec = MotionSpec.from_cfg(config)
    transformer = model.MotionTransformer(spec, device=device)
    path = config.model_pretrain
    if os.path.isfile(path):
        checkpoint = torch.load(path, map_location="cpu")
        print("=> Loaded checkpoint '{}'".format(path))
    else:
        raise ValueError("=> No checkpoint found at '{}'".format(path))
    transformer.load_state_dict(checkpoint)
    return model.ModelInference(model.PositionEmbeddingSine1D(spec.pos_hidden_dim // 2, normalize=True), transformer)


def main(opts):
    random.seed(opts.seed)
    np.random.seed(opts.seed)
    torch.manual_seed(opts.seed)
    config = get_config(opts.config)
    config.out_dir = opts.save_dir
    evaluator = pose_io.Evaluator(config)
    evaluator.set_model(load_model(config))
    subfolders = [f for f in sorted(os.listdir(opts.pose_dir)) if os.path.isdir(os.path.join(opts.pose_dir, f))]
    for sub in subfolders:
        save_path = {"pred_dir": os.path.join(opts.save_dir, "Predict_motion", sub),
                     "linear_dir": os.path.join(opts.save_dir, "Linear_motion", sub)}
        evaluator.interpolate_openpose(os.path.join(opts.pose_dir, sub), sample_rate=opts.upsample_rate, save_dir=save_path)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="motion transformer inference (MI355X)")
    parser.add_argument("--config", type=str, default=os.path.join(os.path.dirname(_HERE), "configs", "motion.yaml"), help="Path to the config file.")
    parser.add_argument("--save-dir", type=str, default="../example/test", help="outputs path")
    parser.add_argument("--pose-dir", type=str, help="input low FPS pose path")
    parser.add_argument("--upsample-rate", type=int, default=8, help=" N-1 frames to insert between two frames")
    parser.add_argument("--seed", type=int, default=123)
    main(parser.parse_args())
