"""Oracle-only (CPU) probe: how chaotic is a full BDM trajectory on procedural weights, as a function of the scale of
the denoisers' last layer?  Shape 1 of the batch is shape 0 with its initial cloud moved by one float32 ulp; both see
the same noise.  Prints the rel-L2 distance between the two along the trajectory (the oracle's SELF-sensitivity).
Used once to choose the head scale of tests/test_hip_full_trajectory.py; not product code.

    python tools/chaos_probe.py --scale 0.01 --points 1024 [--merging]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--milestones", type=str, default="1000,968,936,872,128,64,32,0")
    ap.add_argument("--roll", type=int, default=16)
    ap.add_argument("--merging", action="store_true")
    ap.add_argument("--hip", action="store_true", help="run the twin pair on the HIP path instead of the oracle (fast proxy)")
    a = ap.parse_args()
    import trajectory_case as case
    c = case.build(a.points, a.scale, [int(v) for v in a.milestones.split(",")], a.roll, merging=a.merging, twin=True)
    t0 = time.time()
    out = case.run_hip(c) if a.hip else case.run_oracle(c, progress=True)
    d = float((out[1] - out[0]).norm() / out[0].norm())
    print(f"{'HIP' if a.hip else 'oracle'} scale {a.scale}{' merging' if a.merging else ''}: final self-sensitivity (1 ulp) = {d:.3e}"
          f"   [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
