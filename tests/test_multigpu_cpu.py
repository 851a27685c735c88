"""The N > 1 path of bench.py on CPU: two gloo ranks, 'replicas only' aggregation
(time = max over ranks, units = sum over ranks, no data-path collective)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, u = bench.aggregate(dist, 1.0 + rank, 100 * (rank + 1))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, t, u))


def test_aggregate_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t, u in res:
        assert t == pytest.approx(2.0)       # MAX over ranks
        assert u == pytest.approx(300.0)     # SUM over ranks


def test_batch_partition_rule():
    """icp_batch_*: registration i -> slot i mod n, entry i / n; the slots' counts add up; gather order = index order.
    Pure host function of the library (no device needed)."""
    import icp_amd
    for B, n in ((512, 8), (64, 8), (5, 8), (7, 2), (1, 1), (9, 4)):
        seen = {}
        for i in range(B):
            slot, idx, cnt = icp_amd.batch_partition(B, n, i)
            assert slot == i % n and idx == i // n
            assert cnt == len(range(slot, B, n))
            assert (slot, idx) not in seen
            seen[(slot, idx)] = i
        assert sum(icp_amd.batch_partition(B, n, s)[2] for s in range(min(B, n))) == B
    assert icp_amd.batch_partition(512, 8, 511) == (7, 63, 64)               # BASELINE config 4: 64 per GPU
    with pytest.raises(icp_amd.ICPError):
        icp_amd.batch_partition(4, 0, 0)
    with pytest.raises(icp_amd.ICPError):
        icp_amd.batch_partition(4, 2, 4)


def test_aggregate_single_process():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.aggregate(None, 0.5, 40) == (0.5, 40)
    from icp_amd import workloads as W
    assert W.algorithmic_bytes(16384, 256) == 72 * 16384 + 32 * 256 + 64 == 1187904
    assert W.algorithmic_bytes(65536, 1024) == 4751424 and W.algorithmic_bytes(1 << 20, 4096) == 75628608
    # registrations per GPU: the headline on one GPU, BASELINE config 4 (64 per GPU) on several, --batch wins
    assert bench.default_batch(1, 0) == 1 and bench.default_batch(8, 0) == 64 and bench.default_batch(8, 3) == 3
    assert bench.default_batch(1, 64) == 64


def test_gpus_flag_is_what_the_launch_uses():
    """bench.py --gpus N: under a launcher the rank count must equal N; a plain invocation with N > 1 drives N devices in-process;
    a mismatch or a missing device ends the run without a JSON line."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    assert bench.resolve_launch(1, {}) == ("single", 1)
    assert bench.resolve_launch(1, {"WORLD_SIZE": "1"}) == ("single", 1)
    assert bench.resolve_launch(8, {"WORLD_SIZE": "8"}) == ("ranks", 8)
    assert bench.resolve_launch(4, {}) == ("inprocess", 4)
    for gpus, env in ((8, {"WORLD_SIZE": "2"}), (1, {"WORLD_SIZE": "8"}), (0, {})):
        with pytest.raises(SystemExit):
            bench.resolve_launch(gpus, env)
    # no GPU in this container: the in-process path must refuse (there is no CPU fallback), loudly and without a line
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    import icp_amd
    if icp_amd.device_count() < 2:
        assert out.returncode != 0 and "device(s) are visible" in out.stderr
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
