"""GPU: libvorta_sp.so on a REAL RCCL communicator of one rank (the one-GPU box cannot hold two: RCCL refuses two ranks on a
device): id, init, the three exchanges, destroy through the C ABI -- on one rank every exchange is the identity map of the
reference (all_to_all_4D / all_gather with P = 1); the multi-rank index maps are pinned on the CPU (tests/test_sp_abi.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_world_of_one_through_the_c_abi():
    from vorta_amd.ulysses.rccl_abi import SpComm
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    uid = SpComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = SpComm(0, 1, uid)
    try:
        assert (comm.rank, comm.size) == (0, 1)
        g = torch.Generator(device=dev).manual_seed(3)
        for dtype in (torch.bfloat16, torch.float16, torch.float32):
            x = torch.randn((2, 6, 40, 128), generator=g, device=dev).to(dtype)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # the caller's stream is honoured
                y = comm.seq2head(x)
                z = comm.head2seq(y)
                t = comm.allgather_heads(x)
            side.synchronize()
            assert torch.equal(y, x) and torch.equal(z, x) and torch.equal(t, x)
        b = torch.randint(0, 255, (1, 4, 16, 128), device=dev, dtype=torch.uint8)  # e4m3 / int8 rows travel as bytes
        assert torch.equal(comm.seq2head(b), b)
        with pytest.raises(ValueError):
            comm.seq2head(x.transpose(1, 2))
    finally:
        comm.destroy()
