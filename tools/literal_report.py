#!/usr/bin/env python3
"""Writes tests/golden/literal_vs_recipe.json: per configuration, how far the frames and hit indices of the bit-exact
recipes are from those of the reference's literal arithmetic (tests/literal_study.py).  CPU only.

    python tools/literal_report.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ray_tracing_amd as rta          # host-side constructors only (no GPU needed)
from oracle import oracle as orc
from literal_study import cases, study


def main():
    out = {"what": "ORC_ARITH_RECIPE (bit-identical to the HIP kernels) against ORC_ARITH_LITERAL (no FMA, IEEE divide, v / sqrt(v.v), "
                   "libm sinf/cosf, Sphere::hit's distance in double): same scenes, seeds and draws, oracle/rt_oracle.c",
           "configs": {}}
    for name, (cfg, grid_n) in cases(rta.scenes, full=True).items():
        out["configs"][name] = study(orc, cfg, grid_n)
        r = out["configs"][name]
        print(f"{name:11s} first hits {r['first_hit_agreement_percent']:8.4f} %  segments {r['segment_hit_agreement_percent']:8.4f} %  "
              f"max-abs {r['image_max_abs']:.4f}  PSNR {r['image_psnr_db']} dB", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "literal_vs_recipe.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
