"""The N>1 path on CPU: world_size-2 gloo process group.  The GPU compute is replaced by an oracle-backed stand-in
(tests may use the oracle); what is tested is the sharding, the single exchange step and the timing reduction."""
import os
import socket

import numpy as np
import pytest


class OracleBackend:
    """CPU stand-in with the GpuBackend interface (TEST ONLY)."""

    def __init__(self, oracle, n, hop, window):
        import torch
        self.torch, self.o, self.n, self.hop, self.win = torch, oracle, n, hop, window
        self.H = n // 2 + 1
        self.feedblocks = n // hop

    def to_device(self, samples):
        return np.ascontiguousarray(samples, dtype=np.float32)

    def _power(self, samples, n_frames):
        idx = (np.arange(n_frames) * self.hop)[:, None] + np.arange(self.n)[None, :]
        frames = (samples[:, idx] * self.win[None, None, :]).astype(np.float32)
        return self.o.power_spectrum(frames)                     # [C][F][H] float32

    def partial_power(self, samples, n_frames):
        p = self._power(samples, n_frames)
        acc = np.zeros(p.shape[1:], np.float32)
        for c in range(p.shape[0]):
            acc = (acc + p[c]).astype(np.float32)
        return self.torch.from_numpy(acc)

    def finish_db(self, power, total):
        return self.torch.from_numpy(self.o.to_db((power.numpy() / np.float32(total)).astype(np.float32)))

    def per_channel_db(self, samples, n_frames):
        return self.torch.from_numpy(self.o.to_db(self._power(samples, n_frames)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_channels, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from oracle import jsg_oracle as oracle
    from jadespectrogram_amd.sharded import ShardedSpectrogram, max_over_ranks, shard_channels
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, hop, F = 1024, 512, 6
    win = oracle.window(oracle.WIN_HANN, n)
    x = oracle.synth_audio(n_channels, F * hop + n, seed=11)      # every rank can build the full input; uses its shard
    sh = ShardedSpectrogram(n_channels, OracleBackend(oracle, n, hop, win))
    mine = sh.local_channels()
    assert mine == shard_channels(n_channels, world, rank)
    mixed = sh.absmean(x[mine.start:mine.stop], F).numpy()
    per = sh.per_channel(x[mine.start:mine.stop], F).numpy()
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine.start, per))
    tmax = max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        q.put((mixed, gathered, tmax))
    dist.destroy_process_group()


def test_shard_partition_is_balanced_and_complete():
    from jadespectrogram_amd.sharded import shard_channels
    for C in (1, 2, 5, 8, 63, 64):
        for world in (1, 2, 3, 8):
            parts = [shard_channels(C, world, r) for r in range(world)]
            assert [c for p in parts for c in p] == list(range(C))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert list(shard_channels(64, 8, 3)) == list(range(24, 32))      # BASELINE configs[3]: 8 channels per GPU
    with pytest.raises(ValueError):
        shard_channels(4, 2, 2)


def test_world_size_2_absmean_and_per_channel(oracle):
    import torch.multiprocessing as mp
    world, C = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, q)) for r in range(world)]
    for p in procs:
        p.start()
    mixed, gathered, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0                                            # max over ranks of (1, 2)
    n, hop, F = 1024, 512, 6
    win = oracle.window(oracle.WIN_HANN, n)
    x = oracle.synth_audio(C, F * hop + n, seed=11)
    single = OracleBackend(oracle, n, hop, win)
    # per-channel: the concatenation of the shards equals the single-process run bit for bit
    ref_per = single.per_channel_db(x, F).numpy()
    got = np.concatenate([g[1] for g in sorted(gathered, key=lambda t: t[0])], axis=0)
    assert (got.view(np.uint32) == ref_per.view(np.uint32)).all()
    # AbsMean across shards: equal to the reference's sequential float sum up to float32 re-association
    ref_mix = oracle.to_db(oracle.mix_channels(single._power(x, F), oracle.MIX_ABSMEAN))
    assert np.abs(mixed - ref_mix).max() < 1e-4
