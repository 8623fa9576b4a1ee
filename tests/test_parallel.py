"""Multi-GPU plumbing, exercised on CPU: stripe arithmetic and the N>1 gather over gloo (world 2, 3)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as graft


def _partition_cls():
    graft.load_package()
    from haskell_path_tracer_amd.parallel import StripePartition, gather_color
    return StripePartition, gather_color


@pytest.mark.parametrize("height,stripe,parts", [(1080, 8, 8), (1080, 8, 3), (600, 8, 8), (61, 4, 2), (5, 8, 4), (2160, 16, 8)])
def test_stripes_cover_every_row_exactly_once(height, stripe, parts):
    SP, _ = _partition_cls()
    seen = np.zeros(height, np.int32)
    for p in range(parts):
        part = SP(height, parts, p, stripe)
        rows = part.global_rows()
        assert rows.size == part.local_rows and np.all(np.diff(rows) > 0)
        seen[rows] += 1
        owner = (rows // stripe) % parts
        assert np.all(owner == p)
    assert np.all(seen == 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, stripe, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    SP, gather_color = _partition_cls()
    part = SP(height, world, rank, stripe)
    rows = torch.as_tensor(part.global_rows())
    cols = torch.arange(width)
    # colour of pixel (y, x), channel c := 10000 c + 100 y + x  -- any mix-up shows
    local = torch.stack([10000.0 * c + 100.0 * rows[:, None] + cols[None, :] for c in range(3)]).float()
    full = gather_color(local, part, dst=0)
    if rank == 0:
        ys = torch.arange(height)
        want = torch.stack([10000.0 * c + 100.0 * ys[:, None] + cols[None, :] for c in range(3)]).float()
        torch.save(torch.equal(full, want), out_path)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height,stripe", [(2, 64, 8), (2, 61, 8), (3, 50, 4)])
def test_gather_color_over_gloo(tmp_path, world, height, stripe):
    out = str(tmp_path / "ok.pt")
    mp.spawn(_worker, args=(world, _free_port(), height, 24, stripe, out), nprocs=world, join=True)
    assert torch.load(out) is True


def test_gather_color_in_the_8_gpu_shape(tmp_path):
    """The shape `bench.py --gpus 8` has: a 3840x2160 image in 10-row stripes over 8 ranks, 270 rows and 12.4 MB of colour
    per rank, eight gather buffers on the root -- over gloo, on the CPU."""
    out = str(tmp_path / "ok8.pt")
    mp.spawn(_worker, args=(8, _free_port(), 2160, 3840, 10, out), nprocs=8, join=True)
    assert torch.load(out) is True


def test_gather_color_single_rank_is_a_row_scatter():
    SP, gather_color = _partition_cls()
    part = SP(10, 1, 0, 4)
    local = torch.arange(3 * 10 * 5, dtype=torch.float32).reshape(3, 10, 5)
    assert torch.equal(gather_color(local, part), local)
