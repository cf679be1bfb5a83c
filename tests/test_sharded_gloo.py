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
    # the same stream cut along time instead: every rank takes a run of frames (+ halo) of all channels
    fr, cols = sh.time_sharded(x, F)
    tgathered = [None] * world
    dist.all_gather_object(tgathered, (fr.start, fr.stop, None if cols is None else cols.numpy()))
    tmax = max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        q.put((mixed, gathered, tmax, tgathered))
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


def test_time_shards_cover_every_frame_once_and_read_the_right_samples():
    from jadespectrogram_amd.sharded import frame_span, shard_frames
    for n, hop, fb in ((1024, 512, 2), (2048, 512, 4), (1024, 102, 10), (1024, 1024, 1)):
        for F in (0, 1, 7, 20, 4096, 4099):
            for world in (1, 2, 3, 8):
                parts = [shard_frames(F, fb, world, r) for r in range(world)]
                assert [j for p in parts for j in p] == list(range(F))
                assert all(p.start % fb == 0 for p in parts if len(p))
                blocks = [-(-len(p) // fb) for p in parts]
                assert max(blocks) - min(blocks) <= 1
                for p in parts:
                    span = frame_span(p, n, hop, fb)
                    starts = [(j // fb) * n + (j % fb) * hop for j in p]
                    if starts:
                        assert span.start == min(starts) and span.stop == max(starts) + n
                    else:
                        assert len(span) == 0
    with pytest.raises(ValueError):
        frame_span(range(1, 3), 1024, 512, 2)


def test_world_size_2_absmean_and_per_channel(oracle):
    import torch.multiprocessing as mp
    world, C = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, q)) for r in range(world)]
    for p in procs:
        p.start()
    mixed, gathered, tmax, tgathered = q.get(timeout=120)
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
    # time-axis shards: consecutive, complete, and their columns are the single-process columns bit for bit
    tg = sorted(tgathered, key=lambda t: t[0])
    assert tg[0][0] == 0 and tg[-1][1] == F and all(a[1] == b[0] for a, b in zip(tg, tg[1:]))
    got_t = np.concatenate([g[2] for g in tg if g[2] is not None], axis=1)
    assert (got_t.view(np.uint32) == ref_per.view(np.uint32)).all()
    # AbsMean across shards: equal to the reference's sequential float sum up to float32 re-association
    ref_mix = oracle.to_db(oracle.mix_channels(single._power(x, F), oracle.MIX_ABSMEAN))
    assert np.abs(mixed - ref_mix).max() < 1e-4


def _worker_edge(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    from oracle import jsg_oracle as oracle
    from jadespectrogram_amd.sharded import ShardedSpectrogram
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, hop, F = 1024, 512, 4
    win = oracle.window(oracle.WIN_HANN, n)
    x = oracle.synth_audio(1, F * hop + n, seed=5)               # ONE channel on two ranks: rank 1 has no channel
    sh = ShardedSpectrogram(1, OracleBackend(oracle, n, hop, win))
    mine = sh.local_channels()
    mixed = sh.absmean(x[mine.start:mine.stop], F).numpy()       # must not hang in the all-reduce
    # frame counts that differ between the ranks are refused on every rank (before any data is reduced)
    try:
        sh.absmean(x[mine.start:mine.stop], F - rank)
        refused = False
    except ValueError:
        refused = True
    # a stream that is too short for the requested frames is refused instead of being silently truncated
    try:
        sh.time_sharded(x[:, :(F - 1) * hop + n - 1], F)       # one sample short of what the last frame reads
        short_refused = False
    except ValueError:
        short_refused = True
    got = [None] * world
    dist.all_gather_object(got, (len(mine), refused, short_refused))
    if rank == 0:
        q.put((mixed, got))
    dist.destroy_process_group()


def test_world_size_2_empty_shard_and_mismatched_requests(oracle):
    """a rank without channels still enters the all-reduce with zeros; unequal n_frames and short streams raise."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edge, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mixed, got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [1, 0]
    assert all(g[1] for g in got)
    # only the last rank's run of frames reaches the end of the stream, so only that rank notices the missing sample
    assert got[1][2]
    n, hop, F = 1024, 512, 4
    win = oracle.window(oracle.WIN_HANN, n)
    x = oracle.synth_audio(1, F * hop + n, seed=5)
    single = OracleBackend(oracle, n, hop, win)
    ref = oracle.to_db(oracle.mix_channels(single._power(x, F), oracle.MIX_ABSMEAN))
    assert (mixed.view(np.uint32) == ref.view(np.uint32)).all()
