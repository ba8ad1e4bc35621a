"""Matching when list2 outgrows the Infinity Cache: 50 000 queries against 0.2 ... 3.2 million rows
(bf16 fragments 51 MB ... 820 MB; the MALL holds 256 MB).  Descriptors are generated on the device
(non-negative, unit norm, every 4th query a noisy copy of a row); 200 queries are checked against the
exact scan each time.  Prints ms, ns per 1000 pair evaluations and the share of the bf16 MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import stereo_amd
from stereo_amd import api

ctx = stereo_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(5)
n1 = 50_000
for n2 in (200_000, 400_000, 800_000, 1_600_000, 3_200_000):
    d2 = torch.rand((n2, 128), device="cuda", generator=g).pow_(2)
    d2 /= d2.norm(dim=1, keepdim=True)
    d1 = torch.rand((n1, 128), device="cuda", generator=g).pow_(2)
    idx = torch.randint(0, n2, (n1 // 4,), device="cuda", generator=g)
    d1[::4] = d2[idx] + 0.01 * torch.randn((n1 // 4, 128), device="cuda", generator=g)
    d1 = (d1.abs() / d1.norm(dim=1, keepdim=True)).contiguous()
    s1 = torch.full((n1,), 800.0, device="cuda"); s2 = torch.full((n2,), 900.0, device="cuda")
    o = [torch.empty(n1, dtype=dt, device="cuda") for dt in (torch.int32, torch.int32, torch.float32, torch.float32, torch.uint8)]
    f = lambda: ctx.match_dev(d1, s1, d2, s2, 1000.0, *o)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    st = ctx.match_stats()
    k = 200
    ex = [torch.empty(k, dtype=t_.dtype, device="cuda") for t_ in o]
    ctx.set_match_algo(api.MATCH_EXACT)
    ctx.match_dev(d1[:k].contiguous(), s1[:k].contiguous(), d2, s2, 1000.0, *ex)
    ctx.set_match_algo(api.MATCH_AUTO)
    torch.cuda.synchronize()
    same = all(torch.equal(a[:k].view(torch.uint8), b.view(torch.uint8)) for a, b in zip(o, ex))
    flops = 2.0 * n1 * n2 * 128
    print("%6d x %8d  fragments %4d MB  %8.3f ms  %.3f ns per 1000 pairs  %.3f of bf16 peak  cand/query %.1f  overflow %d  first %d equal exact: %s"
          % (n1, n2, n2 * 256 >> 20, ms, ms * 1e6 / (n1 * float(n2)) * 1e3, flops / (ms * 1e-3) / 2.5e15,
             st["candidates_per_query"], st["overflow_queries"], k, same), flush=True)
    del d1, d2, s1, s2, o, ex
    torch.cuda.empty_cache()
