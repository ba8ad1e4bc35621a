"""The real backend (nccl = RCCL) on a one-GPU box: world size 1, a grouped send/receive to
self of what stereo_amd.dist puts on the wire.  Shows (a) that a 16-bit integer tensor is
refused by torch's NCCL type map -- why gather_rows sends bytes -- and (b) that the byte view
of a u16 row band goes through and views back unchanged.  Run by tests/test_gpu_configs.py."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from stereo_amd.dist import _wire
    band = (torch.arange(270 * 3840, device="cuda", dtype=torch.int32) % 65536).to(torch.uint16).reshape(270, 3840)
    got = torch.empty_like(band)
    # send first: RCCL pairs a receive from self with a send already queued in the group (the
    # other order copies nothing on this version); between real ranks each side posts one kind only
    ops = [dist.P2POp(dist.isend, _wire(band), 0), dist.P2POp(dist.irecv, _wire(got), 0)]
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    torch.cuda.synchronize()
    same = bool(torch.equal(got.view(torch.uint8), band.view(torch.uint8)))
    print("bytes round trip equal:", same)
    # last: a refused batch leaves the group half-open on this torch version (a later batch on
    # the same group then copies nothing), and the product never posts such a tensor
    refused = False
    try:
        t = band.view(torch.int16)
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, 0), dist.P2POp(dist.irecv, torch.empty_like(t), 0)]):
            q.wait()
    except Exception as e:  # "Input tensor data type is not supported for NCCL process group"
        refused = True
        print("int16 refused:", str(e).splitlines()[0])
    print("int16 refused by the backend:", refused)
    dist.destroy_process_group()
    print("WIRE OK" if same else "WIRE FAIL")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
