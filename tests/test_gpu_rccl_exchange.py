"""-m gpu: the RCCL leg of the halo exchange on the one GPU we have: torch.distributed backend "nccl" (= RCCL) with
world_size 1, HaloExchange wrapping raw HIP device pointers (__cuda_array_interface__) and exchanging a slot with
itself (peer = own rank).  Checks pointer wrapping is zero-copy and that isend/irecv on the wrapped buffers moves the
bytes; the multi-rank slot logic itself is covered by tests/test_multirank_gloo.py (gloo) and
tests/test_gpu_multirank_emulated.py (two tse_ctx on one GPU)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_halo_exchange_on_rccl_self_peer():
    import torch
    import torch.distributed as dist
    from transport_se_amd.driver import HaloExchange
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nlyr, ncol = 7, 12
        send = torch.arange(ncol * nlyr, dtype=torch.float64, device="cuda:0") + 0.5
        recv = torch.zeros(ncol * nlyr, dtype=torch.float64, device="cuda:0")
        sched = dict(send=[(0, 1, 5), (0, 6, 7)], recv=[(0, 1, 5), (0, 6, 7)])   # two slots, both to/from rank 0
        ex = HaloExchange(sched, "cuda:0", dist, torch)
        # zero-copy wrap of a raw device pointer
        w = ex._wrap(send.data_ptr(), send.numel())
        assert w.data_ptr() == send.data_ptr() and torch.equal(w, send)
        torch.cuda.synchronize()
        assert ex(send.data_ptr(), recv.data_ptr(), nlyr) == 0
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
        # kind 1 (compact min/max exchange): other slot lengths over the same buffers
        ex.set_minmax_layout([2, 3], [2, 3])
        recv.zero_()
        assert ex(send.data_ptr(), recv.data_ptr(), nlyr, 1) == 0
        torch.cuda.synchronize()
        assert torch.equal(recv[:5 * nlyr], send[:5 * nlyr]) and not recv[5 * nlyr:].any()
    finally:
        dist.destroy_process_group()
