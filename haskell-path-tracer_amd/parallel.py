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


class ColorGatherer:
    """The read-out collective with its buffers and row indices allocated once (a bench step or a
    display loop calls it every frame): gather [3, local_rows, W] from every rank to `dst`, reassemble
    [3, H, W] there.  Ranks may hold different row counts (H not a multiple of stripe_rows * n_parts):
    tensors are padded to the largest.  With backend "gloo" and CUDA tensors the data takes a CPU detour
    (rehearsal of the N > 1 path on a one-GPU box); RCCL takes the device tensors directly."""

    def __init__(self, partition, width, dtype, device, dst=0, group=None, force_collective=False):
        import torch
        import torch.distributed as dist

        self.partition, self.width, self.dst, self.group = partition, width, dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert self.world == partition.n_parts and self.rank == partition.part
        self.via_cpu = (self.world > 1 and torch.device(device).type == "cuda" and dist.get_backend(group) == "gloo")
        dev = "cpu" if self.via_cpu else device
        self.max_rows = partition.max_rows()
        self.send = None
        if self.world > 1 and partition.local_rows != self.max_rows:
            self.send = torch.zeros((3, self.max_rows, width), dtype=dtype, device=dev)
        self.bufs, self.full, self.rows = None, None, None
        if self.rank == dst:
            self.full = torch.empty((3, partition.height, width), dtype=dtype, device=dev)
            self.rows = [torch.as_tensor(partition.global_rows(p), device=dev) for p in range(self.world)]
            if self.world > 1 or force_collective:      # force_collective: run the collective even with one rank (tests)
                self.bufs = [torch.empty((3, self.max_rows, width), dtype=dtype, device=dev) for _ in range(self.world)]

    def overlapped(self, color_local):
        """Pipelined form for a render loop on a GPU: snapshot the colour planes on the current (render) stream,
        then gather the snapshot on a private stream, so the transfer over xGMI runs beside the next render
        launch.  Call wait() before reading the result or at the end of the loop."""
        import torch

        if self.bufs is None and self.rank == self.dst and self.world == 1 or self.via_cpu or not color_local.is_cuda:
            return self(color_local)
        if not hasattr(self, "_comm"):
            self._comm = torch.cuda.Stream()
            self._snap = torch.empty_like(color_local)
            self._ready = torch.cuda.Event()
        torch.cuda.current_stream().wait_stream(self._comm)     # the previous gather has finished reading the snapshot
        self._snap.copy_(color_local)
        self._ready.record()
        with torch.cuda.stream(self._comm):
            self._comm.wait_event(self._ready)
            out = self(self._snap)
        return out

    def wait(self):
        import torch
        if hasattr(self, "_comm"):
            torch.cuda.current_stream().wait_stream(self._comm)

    def __call__(self, color_local):
        import torch.distributed as dist

        if self.world == 1 and self.bufs is None:
            self.full[:, self.rows[0], :] = color_local
            return self.full
        if self.via_cpu:
            color_local = color_local.cpu()
        send = color_local
        if self.send is not None:
            self.send[:, :color_local.shape[1], :] = color_local
            send = self.send
        dist.gather(send.contiguous(), gather_list=self.bufs, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        for p in range(self.world):
            self.full[:, self.rows[p], :] = self.bufs[p][:, :self.rows[p].numel(), :]
        return self.full


def gather_color(color_local, partition, dst=0, group=None):
    """One-shot form of ColorGatherer: returns the full image on `dst`, None elsewhere."""
    return ColorGatherer(partition, color_local.shape[2], color_local.dtype, color_local.device, dst, group)(color_local)
