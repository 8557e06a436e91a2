"""Can an RCCL all-reduce (issued on a side stream, as GradReducer does) be captured into a hipGraph and replayed?"""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
x = torch.ones(1 << 20, device=dev)
side = torch.cuda.Stream()
dist.all_reduce(x)  # warm-up (communicator creation must happen outside capture)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
y = torch.zeros_like(x)
with torch.cuda.graph(g):
    y.copy_(x * 2)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        w = dist.all_reduce(y, op=dist.ReduceOp.AVG, async_op=True)
    w.wait()
    torch.cuda.current_stream().wait_stream(side)
    z = y + 1
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("captured all-reduce replay ok:", float(z[0]))
dist.destroy_process_group()
