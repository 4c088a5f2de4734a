#!/usr/bin/env python3
"""Does this runtime accept `external` events as nodes of a captured graph, and what do they measure?"""
import torch
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 24, device=dev)
s = torch.cuda.Stream(dev)
s.wait_stream(torch.cuda.current_stream())
for ext in (True, False):
    e0 = torch.cuda.Event(enable_timing=True, external=ext)
    e1 = torch.cuda.Event(enable_timing=True, external=ext)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                e0.record(s)
                for _ in range(20):
                    x.add_(1.0)
                e1.record(s)
        torch.cuda.synchronize()
        for rep in range(3):
            o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            o0.record(s)
            with torch.cuda.stream(s):
                g.replay()
            o1.record(s)
            torch.cuda.synchronize()
            print("external", ext, "replay", rep, "inside %.1f us" % (e0.elapsed_time(e1) * 1e3), "outside %.1f us" % (o0.elapsed_time(o1) * 1e3))
    except Exception as e:
        print("external", ext, "FAILED:", type(e).__name__, str(e)[:300])
        torch.cuda.synchronize()
