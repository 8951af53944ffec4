"""RCCL sanity on a 1-GPU box: the collectives the sampling path uses (barrier, max all_reduce, all_gather of clouds,
broadcast_object_list) on a world-size-1 NCCL (= RCCL) group, through bdm_amd.distributed's own helpers.  The multi-rank logic is
covered by the gloo tests; this covers the backend the 8-GPU run uses.  python tools/rccl_world1_check.py"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdm_amd import distributed as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
D.barrier()
print("max_over_ranks:", D.max_over_ranks(1.25, dev))
local = torch.randn(4, 128, 3, device=dev)
# force the collective path although world == 1
out = [torch.empty_like(local)]
dist.all_gather(out, local)
assert torch.equal(out[0], local)
box = ["run_dir"]; dist.broadcast_object_list(box, src=0); assert box == ["run_dir"]
t = torch.ones(3, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); torch.cuda.synchronize()
print("backend", dist.get_backend(), "ok")
dist.destroy_process_group()
