"""N > 1 path on CPU: two gloo ranks shard the frame stream, receive rank 0's parameters and
agree on the max-over-ranks time (the same helpers bench.py uses with backend nccl)."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import steroscopic_hardware_amd as sha
    dist, r, w, _ = sha.shard.init("gloo")
    assert (r, w) == (rank, world)
    block, maxdisp = sha.shard.broadcast_params(dist, r, 9 if r == 0 else -1, 64 if r == 0 else -1, torch.device("cpu"))
    seeds = sha.shard.frame_seeds(r, w, 4)
    sha.shard.barrier(dist)
    tmax = sha.shard.max_over_ranks(dist, 1.0 + r, torch.device("cpu"))
    # each rank generates its own shard independently (no data-path collective)
    L, R = sha.synth.synth_batch_torch(seeds[:1], 32, 8, "cpu")
    q.put((r, block, maxdisp, seeds, tmax, int(L.sum()) + int(R.sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [(b, d) for _, b, d, _, _, _ in res] == [(9, 64), (9, 64)]          # rank 0's params everywhere
    s0, s1 = res[0][3], res[1][3]
    assert s0 == [1000, 1002, 1004, 1006] and s1 == [1001, 1003, 1005, 1007]   # frame i -> rank i mod 2
    assert sorted(s0 + s1) == list(range(1000, 1008))                          # disjoint, complete
    assert res[0][4] == res[1][4] == 2.0                                       # max over ranks
    assert res[0][5] != res[1][5]                                              # different frames


def test_single_rank_needs_no_process_group():
    import torch
    import steroscopic_hardware_amd as sha
    assert sha.shard.frame_seeds(0, 1, 3) == [1000, 1001, 1002]
    assert sha.shard.frame_seeds(5, 8, 2) == [1005, 1013]
    # BASELINE config 4 on 8 GPUs: 1024 pairs in total, 128 per rank, every frame exactly once (bench.py's default split)
    assert sorted(i for r in range(8) for i in sha.shard.frame_indices(r, 8, 1024 // 8)) == list(range(1024))
    assert sha.shard.broadcast_params(None, 0, 9, 64, torch.device("cpu")) == (9, 64)
    assert sha.shard.max_over_ranks(None, 3.5, torch.device("cpu")) == 3.5
