"""Does the launch tape reproduce the eager loop on a (short) C2 blending trajectory?  Prints max |tape - eager| for a few
setups.  Usage: python tools/tape_check.py [--batch 16] [--points 4096] [--profiler] [--no-pool]   (env: BDM_TAPE_NATIVE, BDM_HOIST)"""
import argparse
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--profiler", action="store_true")
    ap.add_argument("--no-pool", action="store_true")
    ap.add_argument("--milestones", type=str, default="1000,968,936")
    args = ap.parse_args()
    from bdm_amd import model as M, ops, tape
    if args.no_pool:
        tape.POOL = False
    from bdm_amd.config import ProjectConfig
    from bdm_amd.data import SyntheticShapes
    from bdm_amd.model import get_model
    from bdm_amd.pvd import prepare_pvd_model
    from bdm_amd.sampling import batch_streams, bdm_blending
    from bdm_amd.utils.procedural import fill_module_
    device = torch.device("cuda", 0)
    cfg = ProjectConfig()
    cfg.dataset.max_points = args.points
    cfg.aux_run.roll_step = 16
    cfg.aux_run.milestones = [int(v) for v in args.milestones.split(",")]
    cfg.run.rng = "per_shape"
    torch.manual_seed(cfg.run.seed)
    model = fill_module_(get_model(cfg).eval(), seed=cfg.run.seed).to(device)
    pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, device)
    batch = next(iter(SyntheticShapes(list(range(args.batch)), args.batch, seed=cfg.run.seed, image_size=224, num_points=args.points))).to(device)

    def run(tape):
        M.TAPE_STEPS = "1" if tape else "0"
        model._cond_cache = None
        out = bdm_blending(None, batch, cfg, model, pvd, streams=batch_streams(cfg, batch, device, sample_idx=1)).points_padded()
        torch.cuda.synchronize()
        return out.clone()

    prof = None
    if args.profiler:
        from bdm_amd.profiling import KernelClassProfiler
        prof = KernelClassProfiler(every=32).install()
    e = run(False)
    e2 = run(False)
    t = run(True)
    t2 = run(True)
    if prof:
        prof.remove()
    g = getattr(model, "_tape_cache", None)
    print("eager finite", bool(torch.isfinite(e).all()), "| eager vs eager", float((e - e2).abs().max()),
          "| tape vs eager", float((t - e).abs().max()), "| tape vs tape", float((t2 - t).abs().max()),
          "| tape off:", None if g is None else g["off"], "| python entries:", None if g is None or g["tape"] is None else g["tape"].python_entries,
          "| len", None if g is None or g["tape"] is None else len(g["tape"]))
    if g is not None and g["tape"] is not None and os.environ.get("TAPE_DUMP"):
        with open(os.environ["TAPE_DUMP"], "w") as f:
            for i, (fn, a) in enumerate(g["tape"].calls):
                nat = g["tape"].native.get(i)
                nm = getattr(fn, "__name__", str(fn))
                if nm in ("_py", "_into"):
                    nm += ":" + (nat[0] if nat else getattr(a[1], "__name__", str(a[1]))) + (":" + str(getattr(a[2], "__name__", a[2])) if nm.startswith("_into") else "")
                f.write(nm + "\n")
    ops.poll_h2_saturation()


if __name__ == "__main__":
    main()
