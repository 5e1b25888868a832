"""GPU probe: does a hipGraph holding the vendor (MIOpen) 3-channel stem convolution read memory it does not own?
Capture fwd / wgrad of the stem shape, replay, then fill every byte the caching allocator can reach with NaN and replay
again: the outputs of a self-contained graph cannot change."""
import os, sys
import torch

dev = torch.device("cuda:0")
if os.environ.get("DET"):
    torch.backends.cudnn.deterministic = True
torch.manual_seed(0)
N, CO = 256, 64
x = torch.rand(N, 3, 32, 32, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(CO, 3, 3, 3, device=dev) * 0.2).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
gy = torch.randn(N, CO, 32, 32, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def body():
    y = torch.ops.aten.convolution(x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1)
    gw = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False])[1]
    return y, gw


for _ in range(3):
    body()
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
    y, gw = body()
g.replay()
torch.cuda.synchronize()
y0, gw0 = y.clone(), gw.clone()
print("reference replay: y nan", int(torch.isnan(y0.float()).sum()), "gw nan", int(torch.isnan(gw0.float()).sum()), flush=True)

keep, sz = [], 1 << 28
base = torch.cuda.memory_reserved()
while sz >= 512:
    t = torch.empty(sz // 4, device=dev)
    if torch.cuda.memory_reserved() > base:
        del t
        base = torch.cuda.memory_reserved()
        sz //= 2
        continue
    t.fill_(float("nan"))
    keep.append(t)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("after poison: y equal", torch.equal(y.view(torch.int16), y0.view(torch.int16)), "gw equal",
      torch.equal(gw.view(torch.int16), gw0.view(torch.int16)), "y nan", int(torch.isnan(y.float()).sum()),
      "gw nan", int(torch.isnan(gw.float()).sum()), flush=True)
