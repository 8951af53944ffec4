import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p): print(p, open(p).read().strip())
os.system("lscpu | head -20; cat /proc/loadavg")
from bdm_amd.pvd import prepare_pvd_model
from oracle import ref_net
pvd = prepare_pvd_model({"model": None, "nc": 3, "embed_dim": 64, "attention": True, "dropout": 0.1}, "cpu")
sd = pvd.state_dict()
x = torch.randn(1, 3, 4096) * 0.5
t = torch.tensor([500])
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    ref_net.pvcnn_forward(sd, x, t, prefix="model.module.")
    t0 = time.perf_counter(); ref_net.pvcnn_forward(sd, x, t, prefix="model.module."); print(nt, "threads", time.perf_counter() - t0, "s")
