#!/usr/bin/env python
"""Pure ATen, no kernel of this repository: graphs whose intermediates are LARGE and short-lived (every layer's temporaries are freed and
their blocks reused inside the graph's private pool), three graphs replayed concurrently on their own streams.  Do the replays
reproduce the eager results?  (PMN_PROBE_KEEP=1: every temporary is kept alive instead, no reuse.)"""
import os
import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
S, R, N = 3, int(os.environ.get("PMN_PROBE_ROUNDS", "30")), int(os.environ.get("PMN_PROBE_N", "4096"))
keep = os.environ.get("PMN_PROBE_KEEP", "0") == "1"
xs = [torch.randn(N, N, device=dev) for _ in range(S)]
w = [torch.randn(N, N, device=dev) * (0.7 / N ** 0.5) for _ in range(3)]


def net(x, held):
    for i in range(24):
        a = x @ w[i % 3]
        b = torch.tanh(a)
        c = b * 1.01
        x = c + 0.1 * x
        if keep:
            held += [a, b, c, x]
    return x


want = [net(x, []).clone() for x in xs]
torch.cuda.synchronize()
graphs, outs, helds, streams = [], [], [], [torch.cuda.Stream(dev) for _ in range(S)]
for k in range(S):
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    held = []
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        o = net(xs[k], held)
    graphs.append(g)
    outs.append(o)
    helds.append(held)
torch.cuda.synchronize()
bad = {}
for r in range(R):
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            graphs[k].replay()
    torch.cuda.synchronize()
    for k in range(S):
        if not torch.equal(outs[k], want[k]):
            bad.setdefault(k, []).append((r, int((outs[k] != want[k]).sum())))
print(f"pure-ATen graphs, {N}x{N} temporaries ({'kept alive' if keep else 'freed and reused'}), replayed concurrently x{R} "
      f"[GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'unset')}]:", {k: (len(v), v[:3]) for k, v in bad.items()} if bad else "every replay equals eager")
