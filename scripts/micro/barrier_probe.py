import os, time, torch, torch.distributed as dist
lr = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
t = torch.zeros(1, device="cuda")
for name, fn in (("dist.barrier()", lambda: dist.barrier()), ("all_reduce+sync", lambda: (dist.all_reduce(t), torch.cuda.synchronize())),
                 ("dist.barrier(device_ids)", lambda: dist.barrier(device_ids=[lr]))):
    ts = []
    for i in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, ["%.3f" % x for x in ts], flush=True)
dist.destroy_process_group()
