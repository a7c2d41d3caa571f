"""Can a kernel on another stream start in the MIDDLE of a replayed hipGraph?  (trainer.TrainStep wants to start the
all-reduce of a gradient bucket as soon as the backward pass has produced it, while the rest of the captured backward pass
keeps running.)  A torch.cuda.Event(external=True) recorded during capture becomes an event-record NODE of the graph; after
graph.replay() another stream waits for it.  Prints when the side stream's kernel finished relative to the graph."""
import torch

dev = torch.device("cuda", 0)
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
c = torch.empty_like(a)
flag = torch.zeros(1, device=dev)
probe = torch.zeros(1, device=dev)
ev = torch.cuda.Event(external=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        torch.mm(a, b, out=c)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    flag.add_(1.0)          # "bucket ready"
    ev.record()             # external event-record node
    for _ in range(40):     # the rest of the backward pass: ~40 x 1 ms
        torch.mm(a, b, out=c)
side = torch.cuda.Stream()
for it in range(3):
    t0, t_side, t_main = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    t0.record()
    g.replay()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        probe.copy_(flag)   # must see this replay's increment
        t_side.record(side)
    t_main.record()
    torch.cuda.synchronize()
    print(f"replay {it}: side kernel done at {t0.elapsed_time(t_side):.2f} ms, graph done at {t0.elapsed_time(t_main):.2f} ms, "
          f"flag seen by side = {probe.item():.0f} (expect {it + 1})")
