"""GPU box: does c10d's watchdog abort the process when a capture that contains a collective starts while it still holds the
work of an EAGER collective?  (HIP: hipEventQuery on an event whose stream has meanwhile entered a capture ->
hipErrorCapturedEvent.)  usage: c10d_capture_probe.py <seconds to wait before the capture>   — prints 'survived' or dies."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
import torch, torch.distributed as dist
wait = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0), rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
for rep in range(20):
    dist.all_reduce(x, async_op=True).wait()          # eager: its work sits in the watchdog's list until the next poll
    torch.cuda.synchronize()
    time.sleep(wait)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        dist.all_reduce(x, async_op=True).wait()      # RCCL's stream joins the capture
        time.sleep(0.25)                                # the capture stays open over two watchdog periods
    g.replay()
    torch.cuda.synchronize()
print("survived", wait, flush=True)
dist.destroy_process_group()
