"""The read-out collective on real RCCL (backend "nccl"), as far as one GPU allows: a one-rank communicator
runs the same dist.gather, snapshot and side-stream code that bench.py uses with N ranks (the N > 1 data
movement itself is covered by the gloo tests in test_parallel.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import __graft_entry__ as graft
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
graft.load_package()
from haskell_path_tracer_amd.parallel import ColorGatherer, StripePartition
part = StripePartition(40, 1, 0, 8)
g = ColorGatherer(part, 24, torch.float32, "cuda", dst=0, force_collective=True)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ok = True
for k in range(5):
    color = torch.full((3, 40, 24), float(k), device="cuda") + torch.arange(24, device="cuda")
    full = g.overlapped(color)
    g.wait(); torch.cuda.synchronize()
    ok &= bool(torch.equal(full, color))
dist.destroy_process_group()
print("RCCL_OK" if ok else "RCCL_MISMATCH")
''' % ROOT


def test_gather_on_rccl_single_rank():
    res = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300)
    assert "RCCL_OK" in res.stdout, res.stdout + res.stderr
