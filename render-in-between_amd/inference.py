#!/usr/bin/env python3
"""Drop-in for PGNR/inference.py: same flags, same config keys, same directory contract.

    python render-in-between_amd/inference.py --input-dir ../example [--config configs/HSM.yaml]
                                               [--save-dir ../example] [--seed 123]

Unlike the reference it builds only what inference needs: the generator (no discriminator, VGG
perceptual loss, optimisers or h5 dataset; PGNR/models/trainer.py:61-113).
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

import render_in_between_amd as rib                                   # noqa: E402
from render_in_between_amd.evaluator import Evaluator                 # noqa: E402


def load_generator(config, device=None):
    """trainer.net_G with its checkpoint (PGNR/models/trainer.py:61,67; utils/utils.py:107-119)."""
    net_G = rib.Generator(config.gen, device=device)
    path = config.model_pretrain_G
    if os.path.isfile(path):
        checkpoint = torch.load(path, map_location="cpu")
        print("=> Loaded checkpoint '{}'".format(path))
    else:
        raise ValueError("=> No checkpoint found at '{}'".format(path))
    net_G.load_state_dict(checkpoint)
    return net_G


def main(opts):
    random.seed(opts.seed)
    np.random.seed(opts.seed)
    torch.manual_seed(opts.seed)
    config = rib.get_config(opts.config)
    config.out_dir = opts.save_dir
    config.eval_dir = opts.save_dir
    net_G = load_generator(config)
    evaluator = Evaluator(config)
    train_dir = os.path.join(opts.input_dir, "inputs")
    dain_dir = os.path.join(opts.input_dir, "DAIN")
    pose_dir = os.path.join(opts.input_dir, "Predict_motion")
    save_dir = os.path.join(opts.save_dir, "Generated_frames")
    evaluator.evaluate_from_folder(net_G, train_dir, dain_dir, pose_dir, save_dir, gt_dir=None, gen_vid=False)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="pose-guided neural rendering inference (MI355X)")
    parser.add_argument("--config", type=str, default=os.path.join(_HERE, "configs", "HSM.yaml"), help="Path to the config file.")
    parser.add_argument("--save-dir", type=str, default="../example", help="outputs path")
    parser.add_argument("--input-dir", type=str, required=True, help="input low FPS frames and pose input")
    parser.add_argument("--seed", type=int, default=123)
    main(parser.parse_args())
