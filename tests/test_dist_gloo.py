"""World-size-2 CPU test (gloo) of the multi-GPU path's only communication: the
throughput reduction, plus the stream->rank sharding."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from handbrake_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames, secs = shard.reduce_throughput(100.0 * (rank + 1), 1.0 + rank)
    streams = shard.stream_for_rank(rank, world, 5)
    host = shard.reduce_host_path({"value": 1000.0 * (rank + 1), "n_out": 512, "seconds": 0.25 * (rank + 1)})
    q.put((rank, frames, secs, streams, host))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_throughput_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, frames, secs, streams, host in res:
        assert frames == 300.0          # SUM over ranks
        assert secs == 2.0              # MAX over ranks
        # the PCIe-inclusive pass of every rank at once: all ranks' frames over the slowest rank's time
        assert host["value"] == 2048.0 and host["ranks"] == 2 and host["this_rank_value"] == 1000.0 * (rank + 1)
    assert res[0][3] == [0, 2, 4] and res[1][3] == [1, 3]


def test_single_process_identity():
    sys.path.insert(0, ROOT)
    from handbrake_amd import shard
    assert shard.reduce_throughput(10, 2.5) == (10.0, 2.5)
    one = shard.reduce_host_path({"value": 2900.0, "n_out": 512, "seconds": 0.18})
    assert one["value"] == 2900.0 and "ranks" not in one


def _launch_bench(world, extra=()):
    """bench.py under torch.distributed.run exactly as the driver launches it (one rank per GPU), with --dry-run:
    no GPU, gloo instead of RCCL, the steps credit frames and sleep."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "5", "--warmup", "1", "--dry-run", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                 # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_launch_contract_world2_and_4():
    for world in (2, 4):
        line = _launch_bench(world)
        assert line["n_gpus"] == world and line["steps"] == 5 and line["warmup"] == 1
        assert line["scaling"] == "weak" and line["higher_is_better"] is True
        assert line["metric"].startswith("filtered frames/sec, 1080p YUV420p NLMeans+decomb chain")
        # every rank put out 5 steps x 2 x 16 frames of its own stream: SUM over ranks; time = the slowest rank's
        assert line["frames_total"] == world * 5 * 32
        assert line["seconds_max"] >= 5 * 0.002 * world
        assert abs(line["value"] - line["frames_total"] / line["seconds_max"]) < 0.01 * line["value"]
        assert line["pcie_inclusive"]["ranks"] == world
        assert line["data"].startswith("dry-run")


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=dict(os.environ, WORLD_SIZE="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "needs torch.distributed.run" in (r.stderr + r.stdout)
