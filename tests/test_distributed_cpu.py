"""world_size-2 (and 4) gloo tests of the multi-GPU sharding logic on CPU: strips with restarted dither chains + gather to
rank 0 must reproduce the reference's own strip-threaded result; batches need no exchange.  The per-rank encoder is the
CPU oracle here (the HIP path takes its place on the GPU node; tests/test_gpu_parity.py::test_full_size_properties checks
that strip-restart == independent strips on the GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, size_y, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.bind import Oracle, PLANES
    from limg_amd import shard
    orc = Oracle()
    W = 128
    img = orc.photo_noise(W, size_y, 5)
    rows = shard.strip_rows(size_y, world)
    y0, y1 = rows[rank]
    mine = orc.encode3d(np.ascontiguousarray(img[y0:y1]), True)  # fresh chain per strip
    planes = {k: torch.from_numpy(mine[k].view(np.int32) if mine[k].dtype == np.uint32 else mine[k]) for k in PLANES}
    full = shard.gather_planes(planes, rows, W, dist, dst=0)
    # the max-over-ranks timing reduction bench.py does
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        np.savez(os.path.join(tmp, "out.npz"), **{k: v.numpy() for k, v in full.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,size_y", [(2, 64), (2, 200), (4, 264)])
def test_strip_sharded_equals_reference_strips(oracle, tmp_path, world, size_y):
    from oracle.bind import PLANES
    port = _free_port()
    mp.spawn(_worker, args=(world, port, size_y, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "out.npz"))
    img = oracle.photo_noise(128, size_y, 5)
    # the same partition evaluated by the oracle's own strip logic: thread_count == world
    from limg_amd import shard
    rows = shard.strip_rows(size_y, world)
    for k in PLANES:
        want = np.concatenate([oracle.encode3d(np.ascontiguousarray(img[y0:y1]), True)[k] for (y0, y1) in rows], axis=0)
        g = got[k].view(np.uint32) if got[k].dtype == np.int32 else got[k]
        assert np.array_equal(g, want), k
    pool = shard.equivalent_pool_threads(world)
    if pool:  # world == 4: identical to the reference run with a 1-thread pool (4 strips)
        want = oracle.encode3d(img, True, pool_threads=pool)
        for k in PLANES:
            g = got[k].view(np.uint32) if got[k].dtype == np.int32 else got[k]
            assert np.array_equal(g, want[k]), k


def test_plans():
    from limg_amd import shard
    assert shard.strip_rows(16384, 8) == [(i * 2048, (i + 1) * 2048) for i in range(8)]
    assert shard.strip_rows(618, 8)[-1] == (7 * 72, 618) and shard.strip_rows(618, 8)[0] == (0, 72)
    assert shard.strip_rows(40, 8) is None
    assert shard.equivalent_pool_threads(8) == 2 and shard.equivalent_pool_threads(2) is None
    assert shard.batch_assignment(64, 8, 3) == list(range(3, 64, 8))
    assert sorted(sum((shard.batch_assignment(10, 4, r) for r in range(4)), [])) == list(range(10))


def _bytes_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from limg_amd import shard
    n = 1000 + 37 * rank                       # every rank's stream has its own size
    buf = torch.zeros(4096, dtype=torch.uint8)  # worst-case buffer, only the first n bytes are meaningful
    buf[:n] = torch.arange(n, dtype=torch.int64).remainder(251).to(torch.uint8) + rank
    res = shard.gather_streams(None, buf, n, dist, dst=0)   # gloo group: same piece layout as limg_hip_gather_stream (limg_hip_host_gather_offsets), torch transfers
    if rank == 0:
        got, offs = res
        assert len(offs) == world + 1 and int(offs[0]) == 0
        for r in range(world):
            size = 1000 + 37 * r
            want = (torch.arange(size, dtype=torch.int64).remainder(251).to(torch.uint8) + r)
            assert int(offs[r]) % 16 == 0 and int(offs[r + 1]) - int(offs[r]) == (size + 15) // 16 * 16
            assert torch.equal(got[int(offs[r]): int(offs[r]) + size], want), r
        open(os.path.join(tmp, "ok"), "w").write("1")
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_variable_size_stream_gather(tmp_path, world):
    """What config 5's `--gather-stream` does between the ranks: only the used bytes of every rank's compact stream travel to rank 0, each piece to a
    16-byte aligned offset.  On the GPU node the same call goes through limg_hip_gather_stream (RCCL behind the C ABI)."""
    port = _free_port()
    mp.spawn(_bytes_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))


def _views_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # every rank's view travels through the process group's store: no collective on the group (what bench.collective_evidence does after its helper thread,
    # which may still sit inside RCCL, has been given up on)
    views = bench.exchange_views(dist, rank, world, [world, rank, 22605, 1 if rank == 1 else 0], timeout_s=20.0)
    assert [v[1] for v in views] == list(range(world)) and [v[3] for v in views] == [1 if r == 1 else 0 for r in range(world)], views
    # a second exchange uses fresh keys
    again = bench.exchange_views(dist, rank, world, [0, rank, 0, 0], timeout_s=20.0)
    assert [v[0] for v in again] == [0] * world and [v[1] for v in again] == list(range(world)), again
    # a rank that never answers is reported as hung, within the limit
    if rank == 0:
        import time
        t0 = time.time()
        late = bench.exchange_views(dist, rank, world, [1, 0, 1, 0], timeout_s=1.0)
        assert time.time() - t0 < 10 and late[0][:2] == [1, 0] and all(v[3] == 1 for v in late[1:]), late
        open(os.path.join(tmp, "ok"), "w").write("ok")
    else:
        import time
        time.sleep(3.0)  # (does not take part in the third exchange)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_evidence_views_travel_over_the_store(tmp_path, world):
    """bench.py --gpus N: after the RCCL evidence helper of ANY rank may have hung, the ranks agree on what they saw without another collective."""
    mp.spawn(_views_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))
