"""N>1 path on CPU: two gloo ranks shard a batch and gather the trajectories (world_size 2)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from etol_amd import batch
from etol_amd import workloads as W


def test_shard_range_partitions_every_batch():
    for n in (1, 7, 128, 1024, 1025):
        for world in (1, 2, 3, 8):
            blocks = [batch.shard_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [l - f for f, l in blocks]
            assert max(sizes) - min(sizes) <= 1
            for s in range(0, n, max(1, n // 13)):
                r = batch.owner_of(s, n, world)
                assert blocks[r][0] <= s < blocks[r][1]
    # config 4: 1024 scenarios over 8 GPUs = 128 each, instance s on GPU s // 128
    assert batch.shard_range(1024, 8, 3) == (384, 512)


def _worker(rank, world, port, n_inst, M, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = batch.shard_range(n_inst, world, rank)
    # each rank generates only its own scenarios (seeded by the global instance index)
    X, U, _ = W.quadrotor_batch(3, last - first, M, 2, first_instance=first)
    got = batch.gather_trajectories(torch.from_numpy(X), torch.from_numpy(U), n_inst, dst=0)
    if rank == 0:
        Xall, Uall, _ = W.quadrotor_batch(3, n_inst, M, 2, first_instance=0)
        ok.value = int(np.array_equal(got[0].numpy(), Xall) and np.array_equal(got[1].numpy(), Uall))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ok = ctx.Value("i", 0)
    n_inst, M = 5, 16          # odd count: blocks of 2 and 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_inst, M, ok)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ok.value == 1
