import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29631")
os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
t=time.time(); dist.init_process_group("nccl", device_id=dev); print("init", time.time()-t, dist.get_backend(), flush=True)
x=torch.ones(1,device=dev); dist.all_reduce(x); print("allreduce", x.item(), flush=True)
loc=torch.rand(1024,800,device=dev); out=torch.empty(1024,800,device=dev)
s=torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    dist.all_gather_into_tensor(out, loc)
    w=dist.all_gather_into_tensor(out, loc, async_op=True); w.wait()
torch.cuda.synchronize(); print("gather equal", torch.equal(out,loc), flush=True)
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    a.record()
    for _ in range(20): dist.all_gather_into_tensor(out, loc)
    b.record()
torch.cuda.synchronize(); print("us per gather", a.elapsed_time(b)/20*1e3)
dist.destroy_process_group(); print("ok")
