"""Row-stripe partition of the image over ranks and the one collective of the path.

Pixels are independent (`render Inline` is a per-element map, src/Scene/Trace.hs:193-200), so
the image is cut into stripes of `stripe_rows` rows dealt round-robin to the ranks; every rank
keeps its colour + RNG planes resident and renders without any data-path exchange.  The only
collective is the gather of the three colour planes at read-out (what graphicsLoop reads,
app/Main.hs:350) -- torch.distributed gather, which is RCCL over xGMI with backend "nccl" and
gloo in the CPU tests.  The RNG planes never move.
"""
import numpy as np

DEFAULT_STRIPE_ROWS = 8


class StripePartition:
    """Same arithmetic as ptmi_set_partition / ptmi_global_row (include/ptmi.h)."""

    def __init__(self, height, n_parts, part, stripe_rows=DEFAULT_STRIPE_ROWS):
        if height <= 0 or n_parts <= 0 or not (0 <= part < n_parts) or stripe_rows <= 0:
            raise ValueError("bad partition")
        self.height, self.n_parts, self.part, self.stripe_rows = height, n_parts, part, stripe_rows

    @staticmethod
    def rows_of(height, stripe_rows, n_parts, part):
        cycle = stripe_rows * n_parts
        rows = (height // cycle) * stripe_rows
        rem = min(height % cycle - part * stripe_rows, stripe_rows)
        return rows + max(rem, 0)

    @property
    def local_rows(self):
        return self.rows_of(self.height, self.stripe_rows, self.n_parts, self.part)

    def global_rows(self, part=None):
        part = self.part if part is None else part
        n = self.rows_of(self.height, self.stripe_rows, self.n_parts, part)
        lr = np.arange(n, dtype=np.int64)
        s = self.stripe_rows
        return ((lr // s) * self.n_parts + part) * s + lr % s

    def max_rows(self):
        return max(self.rows_of(self.height, self.stripe_rows, self.n_parts, p) for p in range(self.n_parts))


def gather_color(color_local, partition, dst=0, group=None):
    """Gather every rank's [3, local_rows, W] colour tensor to `dst` and reassemble [3, H, W].

    Returns the full image on `dst`, None elsewhere.  Ranks may hold different row counts
    (H not a multiple of stripe_rows * n_parts): tensors are padded to the largest.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    assert world == partition.n_parts and rank == partition.part
    width = color_local.shape[2]
    if world == 1:
        full = torch.empty((3, partition.height, width), dtype=color_local.dtype, device=color_local.device)
        full[:, torch.as_tensor(partition.global_rows(), device=color_local.device), :] = color_local
        return full
    if color_local.is_cuda and dist.get_backend(group) == "gloo":
        color_local = color_local.cpu()          # CPU rehearsal of the N > 1 path; RCCL takes the device tensor
    max_rows = partition.max_rows()
    send = color_local
    if color_local.shape[1] != max_rows:
        send = torch.zeros((3, max_rows, width), dtype=color_local.dtype, device=color_local.device)
        send[:, :color_local.shape[1], :] = color_local
    send = send.contiguous()
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty((3, partition.height, width), dtype=color_local.dtype, device=color_local.device)
    for p in range(world):
        rows = torch.as_tensor(partition.global_rows(p), device=color_local.device)
        full[:, rows, :] = bufs[p][:, :rows.numel(), :]
    return full
