"""Which HIP runtime serves whom?  Loads libstereo_hip.so and torch in either order in this one
process (argv[1] = 'lib-first' or 'torch-first'), creates a context, runs a tiny kernel on both
sides and prints every libamdhip64 / libhsa-runtime64 mapping of the process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

order = sys.argv[1] if len(sys.argv) > 1 else "lib-first"


def maps():
    seen = []
    for line in open("/proc/self/maps"):
        p = line.split()[-1]
        if ("amdhip64" in p or "hsa-runtime64" in p or "libstereo_hip" in p or "libhiprtc" in p) and p not in seen:
            seen.append(p)
    return seen


def use_lib():
    import stereo_amd
    from stereo_amd import synth
    ctx = stereo_amd.Context(0)
    l, r = synth.rectified_pair(1, 16, 64, 8)
    out = ctx.disparity(l, r, win=3, max_disp=8)
    print("libstereo_hip ok, checksum", int(out.sum()), flush=True)
    return ctx


def use_torch():
    import torch
    try:
        torch.cuda.init()
        t = torch.ones(8, device="cuda")
        print("torch ok:", torch.version.hip, float(t.sum().item()), flush=True)
    except Exception as e:  # noqa
        print("torch FAILED:", type(e).__name__, e, flush=True)


print("order:", order, "| HIP_VISIBLE_DEVICES", os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES",
      os.environ.get("ROCR_VISIBLE_DEVICES"), "LD_LIBRARY_PATH", os.environ.get("LD_LIBRARY_PATH"))
if order == "lib-first":
    c = use_lib(); print("\n".join(maps())); use_torch()
else:
    use_torch(); print("\n".join(maps())); c = use_lib()
print("--- final mappings"); print("\n".join(maps()))
