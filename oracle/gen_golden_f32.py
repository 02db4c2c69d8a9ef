#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/f32_reference.npz from the REAL reference (oracle/_ref/libtds_ref.so):
for float-representable input records x (the committed golden inputs rounded to float), the outputs of
  * the reference's own float instantiation  TinyAlgebra<float, FloatUtils>  (oracle/ref_harness_f32.cpp)  -> y32
  * the reference's double path on the same x                                                               -> y64
so that the GPU box (no /root/reference there) can hold a float kernel against what the reference itself does in
float (tests/test_f32.py).  Run where /root/reference exists:  python oracle/gen_golden_f32.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import tds_amd  # noqa: E402
import reflib  # noqa: E402

# committed model name -> name the reference harness builds it from
MODELS = {"pendulum5": "pendulum5.urdf", "cartpole": "cartpole.urdf", "ant": "ant", "laikago": "laikago",
          "pendulum5_plane": "pendulum5.urdf+plane"}
N = 64


def main():
    out = {}
    for name, refname in MODELS.items():
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        m = tds_amd.load_model(name)
        x = g["x"][:N].astype(np.float32).astype(np.float64)
        rf = reflib.RefSimF32(refname, m.input_dim, m.output_dim, m.dt)
        rd = reflib.RefSim(refname)
        if refname.endswith("urdf") or "+plane" in refname:
            rd.set_dt(m.dt)
        out[name + "_x"] = x
        out[name + "_y32"] = rf.step(x)
        out[name + "_y64"] = rd.step(x)
        rf.close()
        rd.close()
        e = np.max(np.abs(out[name + "_y32"] - out[name + "_y64"]) / np.maximum(np.abs(out[name + "_y64"]), 1e-3))
        print(f"{name}: reference float path vs its double path: {e:.3e}", file=sys.stderr)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "f32_reference.npz"), **out)


if __name__ == "__main__":
    main()
