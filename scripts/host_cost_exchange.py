import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
flat=torch.zeros(201740,device=dev); g=torch.randn(32,70,28,28,device=dev)
comm=torch.cuda.Stream()
def T(name, fn, n=500):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    h=time.perf_counter()-t; torch.cuda.synchronize(); tot=time.perf_counter()-t
    print(f"{name:40s} host {h/n*1e6:7.1f} us   total {tot/n*1e6:7.1f} us")
cur=torch.cuda.current_stream()
T("record_event", lambda: cur.record_event())
T("record_stream", lambda: g.record_stream(comm))
def ctx():
    with torch.cuda.stream(comm): pass
T("stream ctx", ctx)
ev=cur.record_event()
T("wait_event", lambda: comm.wait_event(ev))
T("fill (reshape+slice+copy)", lambda: flat[:201740].copy_(g.reshape(-1)[:201740]))
T("all_reduce sync", lambda: dist.all_reduce(flat, op=dist.ReduceOp.AVG))
def ar_side():
    with torch.cuda.stream(comm): dist.all_reduce(flat, op=dist.ReduceOp.AVG)
T("all_reduce on side stream", ar_side)
def ar_async():
    w=dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True); w.wait()
T("all_reduce async+wait", ar_async)
dist.destroy_process_group()
