"""Differential fuzz of the matching prefilter against the exact GPU scan: random shapes (few to many query
blocks, short to long list2), ordinary and clustered descriptors, five outputs compared as bits.
python tools/exp/match_fuzz.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import stereo_amd
from stereo_amd import api, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = stereo_amd.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
rng = np.random.RandomState(2026)
bad = 0
for k in range(rounds):
    n1 = int(2 ** rng.uniform(6, 16.5))
    n2 = int(2 ** rng.uniform(9, 17.5))
    if k % 3 == 2:
        d1, s1, d2, s2 = synth.descriptor_sets_clustered(k, n1, n2, centres=max(2, n2 // int(2 ** rng.uniform(3, 9))),
                                                         noise=10.0 ** rng.uniform(-4, -2))
    else:
        d1, s1, d2, s2 = synth.descriptor_sets(k, n1, n2)
    t = [torch.from_numpy(a).cuda() for a in (d1, s1, d2, s2)]
    outs = lambda: [torch.empty(n1, dtype=dt, device="cuda") for dt in (torch.int32, torch.int32, torch.float32, torch.float32, torch.uint8)]
    a, b = outs(), outs()
    ctx.set_match_algo(api.MATCH_MFMA_BF16X1)
    ctx.match_dev(t[0], t[1], t[2], t[3], 1000.0, *a)
    torch.cuda.synchronize()
    st = ctx.match_stats()
    ctx.set_match_algo(api.MATCH_EXACT)
    ctx.match_dev(t[0], t[1], t[2], t[3], 1000.0, *b)
    torch.cuda.synchronize()
    ctx.set_match_algo(api.MATCH_AUTO)
    same = all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(a, b))
    bad += not same
    print("%2d  %6d x %6d  %s  cand/query %.1f max %d overflow %d  %s" % (
        k, n1, n2, "clustered" if k % 3 == 2 else "ordinary ", st["candidates_per_query"], st["candidates_max"],
        st["overflow_queries"], "equal" if same else "DIFFERENT"), flush=True)
print("different:", bad)
sys.exit(1 if bad else 0)
