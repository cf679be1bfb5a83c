"""Channel-sharded (and, for one long stream, time-sharded) multi-GPU driver: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).

The path shards by independent channels/streams: every rank runs the single-GPU engine on its own
channels and keeps its spectrograms device-local -- NO data-path collective.  The only exchange the reference's
semantics can require is its AbsMean mix when the mixed channels live on different GPUs
(Spectrogram.cpp:68-76: sum over all channels, divide by the channel count, then 10*log10): each rank sums the
linear power of its local channels (JSG_MIX_SUM), one all-reduce(sum) of [frames][bins] float32 follows, and the
divide + dB tail runs on the reduced sums (jsg_db_from_power_launch).  Differs from the single-GPU result only by
float32 re-association of the channel sum.
"""
from __future__ import annotations

import numpy as np


def shard_channels(n_channels: int, world: int, rank: int) -> range:
    """Contiguous, balanced partition of channel indices; the first (n_channels % world) ranks get one more."""
    if not (0 <= rank < world) or n_channels < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_channels, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_frames(n_frames: int, feedblocks: int, world: int, rank: int) -> range:
    """Time-axis partition of ONE long stream ("also legal"): contiguous runs of frames, cut only at
    multiples of `feedblocks` because the reference restarts its frame offsets at every fft-size block
    (Spectrogram.cpp:50-55), balanced in blocks; the first (blocks % world) ranks get one more block."""
    if not (0 <= rank < world) or n_frames < 0 or feedblocks <= 0:
        raise ValueError("bad shard request")
    blocks = (n_frames + feedblocks - 1) // feedblocks
    base, extra = divmod(blocks, world)
    b0 = rank * base + min(rank, extra)
    b1 = b0 + base + (1 if rank < extra else 0)
    return range(min(b0 * feedblocks, n_frames), min(b1 * feedblocks, n_frames))


def frame_span(frames: range, n: int, hop: int, feedblocks: int) -> range:
    """Samples a run of frames reads (frames.start a multiple of feedblocks): its own new samples plus the halo of
    n - hop samples it shares with the following rank.  Frame j starts at (j // feedblocks)*n + (j % feedblocks)*hop."""
    if len(frames) == 0:
        return range(0, 0)
    if frames.start % feedblocks:
        raise ValueError("a shard must start on a block boundary")
    last = frames.stop - 1
    return range((frames.start // feedblocks) * n, (last // feedblocks) * n + (last % feedblocks) * hop + n)


class GpuBackend:
    """Compute backend on the local MI355X through libjsg.so (the product path)."""

    def __init__(self, n: int, hop: int, window: np.ndarray, feedblocks: int | None = None, device=None):
        import torch
        import jadespectrogram_amd as jsg
        self.torch, self.jsg = torch, jsg
        self.n, self.hop, self.H = n, hop, n // 2 + 1
        self.feedblocks = feedblocks if feedblocks is not None else max(1, n // hop)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.plan = jsg.Plan(n, window)
        self.pitch = (self.H + 31) // 32 * 32

    def _frames(self, n_samples: int) -> int:
        return ((n_samples - self.n) // self.n) * self.feedblocks + self.feedblocks if n_samples >= self.n else 0

    def to_device(self, samples: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(samples, dtype=np.float32)).to(self.device)

    def partial_power(self, d_samples, n_frames: int):
        """Sum over the local channels of |X|^2 -> [n_frames][pitch] float32 on the device (no divide, no log).
        A rank without local channels contributes zeros (it still has to enter the all-reduce)."""
        out = self.torch.zeros((n_frames, self.pitch), dtype=self.torch.float32, device=self.device)
        if d_samples.shape[0] == 0:
            return out
        self.jsg.stft_db(self.plan, d_samples, self.hop, n_frames, out, feedblocks=self.feedblocks,
                         mix_mode=self.jsg.capi.MIX_SUM, linear_out=True)
        return out

    def finish_db(self, d_power, total_channels: int):
        self.jsg.spectrogram.db_from_power(d_power, d_power, float(total_channels))
        return d_power

    def per_channel_db(self, d_samples, n_frames: int):
        C = d_samples.shape[0]
        out = self.torch.empty((C, n_frames, self.pitch), dtype=self.torch.float32, device=self.device)
        self.jsg.stft_db(self.plan, d_samples, self.hop, n_frames, out, feedblocks=self.feedblocks,
                         mix_mode=self.jsg.capi.MIX_PER_CHANNEL)
        return out

    def to_host(self, d_tensor) -> np.ndarray:
        return d_tensor[..., :self.H].cpu().numpy()


class ShardedSpectrogram:
    """Spectrogram of `n_channels` channels spread over the ranks of a torch.distributed process group."""

    def __init__(self, n_channels: int, backend, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_channels = n_channels
        self.local = shard_channels(n_channels, self.world, self.rank)
        self.backend = backend

    def local_channels(self) -> range:
        return self.local

    def per_channel(self, local_samples, n_frames: int):
        """Independent spectrograms of the local channels; stays on this rank's device.  No communication."""
        return self.backend.per_channel_db(self.backend.to_device(local_samples), n_frames)

    def time_sharded(self, stream_samples, n_frames: int):
        """One long stream [channels][samples] (every rank sees the same host array, e.g. a memory-mapped file) cut
        along time: this rank computes the per-channel columns of shard_frames(...) from its sample span (own samples +
        halo) and keeps them on its device.  Returns (frames, columns).  No communication."""
        b = self.backend
        frames = shard_frames(n_frames, b.feedblocks, self.world, self.rank)
        span = frame_span(frames, b.n, b.hop, b.feedblocks)
        if len(frames) == 0:
            return frames, None
        if span.stop > stream_samples.shape[1]:
            raise ValueError(f"{n_frames} frames need {span.stop} samples, the stream holds {stream_samples.shape[1]}")
        return frames, b.per_channel_db(b.to_device(stream_samples[:, span.start:span.stop]), len(frames))

    def absmean(self, local_samples, n_frames: int):
        """The reference's AbsMean column over ALL channels; every rank ends up with the full result."""
        if self.world > 1:   # every rank must ask for the same columns, or the reduce would mix different frames
            import torch
            nf = torch.tensor([n_frames, -n_frames], dtype=torch.int64,
                              device=getattr(self.backend, "device", None) if self.dist.get_backend(self.group) == "nccl" else "cpu")
            self.dist.all_reduce(nf, op=self.dist.ReduceOp.MAX, group=self.group)
            if int(nf[0]) != n_frames or int(nf[1]) != -n_frames:
                raise ValueError("absmean: n_frames differs between ranks")
        power = self.backend.partial_power(self.backend.to_device(local_samples), n_frames)
        if self.world > 1:
            self.dist.all_reduce(power, op=self.dist.ReduceOp.SUM, group=self.group)   # the one exchange step
        return self.backend.finish_db(power, self.n_channels)


def max_over_ranks(value: float, device=None) -> float:
    """max over ranks of a host scalar (bench timing); identity without a process group."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
