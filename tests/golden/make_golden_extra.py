"""Two more fixtures made by the imported REFERENCE generator (round 6), added without touching the existing ones:

    python tests/golden/make_golden_extra.py          (build container only: needs /root/reference)

  full_b3_96x160   batch 3, non-square, odd multiples of 32 (the batched / ragged shapes the folder driver produces)
  full_1024        BASELINE configs[4]'s frame size, 1024x1024, stored at every 16th pixel (+ the summaries of the whole frame):
                   until now that size was compared with the oracle only

Same recipe as make_golden.py (seed-defined checkpoint and inputs from the build's own synth module, the reference's
outputs stored, the oracle asserted <= 1e-5 from the reference); the entries are merged into golden_report.json.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import render_in_between_amd as rib      # noqa: E402
import make_golden                       # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    full = rib.hsm_gen_config()
    report = {}
    make_golden.run_case("full_b3_96x160", full, 8, 3, 96, 160, 2, report)
    # (a million pixels: the max over the frame of two fp32 evaluations that sum in different orders is 1.6e-5; tolerance 2.5e-5 here)
    make_golden.run_case("full_1024", full, 9, 1, 1024, 1024, 16, report, tol=2.5e-5)
    path = os.path.join(HERE, "golden_report.json")
    with open(path) as f:
        merged = json.load(f)
    merged.update(report)
    with open(path, "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
