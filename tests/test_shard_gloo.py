"""N > 1 path on CPU: two gloo ranks shard frames the way bench.py does (frame f -> rank f mod N), with the same
barrier + max-over-ranks timing reduction; no data-path collective is involved."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.frames_for_rank(total, rank, world)
    shard.barrier(dist)
    wall = 1.0 + rank * 0.5 + len(mine) * 0.01          # stand-in for the timed region of this rank
    mx = shard.max_over_ranks(wall, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, mine, mx, gathered))
    dist.destroy_process_group()


def test_two_ranks_partition_frames_and_reduce_max():
    world, total = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(60); assert p.exitcode == 0
    frames = sorted(f for _, mine, _, _ in res for f in mine)
    assert frames == list(range(total))                               # every frame rendered exactly once
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    expect = max(1.0 + r * 0.5 + len(shard.frames_for_rank(total, r, world)) * 0.01 for r in range(world))
    assert all(abs(mx - expect) < 1e-12 for _, _, mx, _ in res)       # both ranks agree on the max
    assert all(g == [res[0][1], res[1][1]] for _, _, _, g in res)


def test_single_process_helpers_without_dist():
    assert shard.frames_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert shard.max_over_ranks(3.5) == 3.5
    shard.barrier(None)


def _strip_worker(rank, world, port, q):
    """Single-frame mode: each rank asks the library (host logic only, no GPU) for its tile-column strip, fills its columns of a
    stand-in output with its rank id, and rank 0 gathers the column ranges - the writer-side reassembly of SURVEY 8e."""
    import importlib
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("waifu2x-tensorrt_amd")
    first, cnt, x0, x1 = pkg.strip_plan(1920, 1080, 7680, 4320, 256, 960, 4, (0.0625, 0.0625), rank, world)
    cols = torch.zeros(7680, dtype=torch.int32)
    cols[x0:x1] = rank + 1
    dist.all_reduce(cols)                      # test-only check that the ranges are disjoint and complete (not a data-path collective)
    gathered = [None] * world
    dist.all_gather_object(gathered, (first, cnt, x0, x1))
    q.put((rank, gathered, bool(((cols >= 1) & (cols <= world)).all()), int((cols == rank + 1).sum())))
    dist.destroy_process_group()


def test_two_ranks_split_one_frame_into_tile_column_strips():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path: sys.path.insert(0, root)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs: p.join(60); assert p.exitcode == 0
    plans = res[0][1]
    assert all(r[1] == plans for r in res) and all(r[2] for r in res)
    # config 3: 9x5 tiles; rank 0 owns tile columns 0..4, rank 1 columns 5..8 plus column 4 again for the blend band: 25 tiles each
    assert plans[0] == (0, 25, 0, 5 * 896) and plans[1] == (20, 25, 5 * 896, 7680)
    assert res[0][3] == 5 * 896 and res[1][3] == 7680 - 5 * 896


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """bench.py must never print n_gpus != requested: under a launcher whose WORLD_SIZE disagrees with --gpus it stops before
    touching the GPU; without a launcher and with fewer devices than --gpus the self-spawn path refuses as well."""
    import subprocess, sys
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to report n_gpus" in (r.stderr + r.stdout)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "W2X_DEVICE_MAP")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to report n_gpus" in (r.stderr + r.stdout)


def _one_json_line(stdout: str) -> dict:
    import json
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                       # rank 0's stdout carries exactly one line
    return json.loads(lines[0])


@pytest.mark.parametrize("world,launcher", [(2, "self"), (8, "self"), (2, "torchrun"), (8, "torchrun")])
def test_bench_rank_protocol_at_two_and_eight_ranks_without_a_gpu(world, launcher):
    """bench.py --host-rehearsal: everything a multi-rank run does EXCEPT the engine calls - ranks spawned by the script itself or started by
    `python -m torch.distributed.run` exactly as the driver starts them (one process per GPU, rendezvous on 127.0.0.1), gloo process group, placement records
    gathered from every rank, barrier / K steps / barrier, MAX over ranks, per-rank spread, frame f -> rank f mod N, and rank 0 printing ONE JSON line on
    stdout.  The first real 8-GPU launch is then not also the first 8-process run of the script.  The line cannot be mistaken for a result: value is null,
    n_gpus counts distinct GPUs (none here) and the metric starts with REHEARSAL."""
    import subprocess, sys
    steps = 3
    args = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", "1", "--host-rehearsal"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "W2X_DEVICE_MAP")}
    env["OMP_NUM_THREADS"] = "1"
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["value"] is None and d["n_gpus"] == 0 and d["n_ranks"] == world and d["metric"].startswith(f"REHEARSAL ({world} ranks on 0 GPUs)")
    assert d["steps"] == steps and d["warmup"] == 1 and d["scaling"] == "weak"
    p = d["placement"]
    assert [rec["rank"] for rec in p["ranks"]] == list(range(world)) and [rec["local_rank"] for rec in p["ranks"]] == list(range(world))
    assert len({rec["pid"] for rec in p["ranks"]}) == world and p["distinct_gpus"] == 0 and not p["one_gpu_per_rank"]
    assert d["config"]["frames_rendered"] == list(range(steps * world))      # every frame of the weak-scaled job exactly once
    spread = d["config"]["per_rank_ms_per_step"]
    assert spread["min"] <= spread["mean"] <= spread["max"] and d["ms_per_step"] >= spread["max"] - 1e-3     # the line's time is the MAX over ranks


def test_placement_certificate_counts_distinct_gpus():
    """shard.certify: ranks that share a card (a W2X_DEVICE_MAP rehearsal) or sit on different hosts are told apart by (host, PCI bus id)."""
    rec = lambda r, host, bus: {"rank": r, "local_rank": r, "host": host, "pid": 100 + r, "pci_bus_id": bus, "cpus": None}
    real = shard.certify([rec(r, "node0", f"0000:{r:02x}:00.0") for r in range(8)])
    assert real["distinct_gpus"] == 8 and real["one_gpu_per_rank"] and real["n_ranks"] == 8 and real["device_map"] is None
    shared = shard.certify([rec(r, "node0", "0000:05:00.0") for r in range(4)], "0,0,0,0")
    assert shared["distinct_gpus"] == 1 and not shared["one_gpu_per_rank"] and shared["device_map"] == "0,0,0,0"
    two_hosts = shard.certify([rec(0, "a", "0000:05:00.0"), rec(1, "b", "0000:05:00.0")])
    assert two_hosts["distinct_gpus"] == 2 and two_hosts["one_gpu_per_rank"]
    assert shard.spread([1.0, None, 3.0]) == {"min": 1.0, "max": 3.0, "mean": 2.0} and shard.spread([None]) is None


def _fake_sysfs(root, gpus):
    """A KFD topology tree as the driver lays it out: CPU nodes (simd_count 0) first, then GPUs with a DRM render node each."""
    nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    os.makedirs(os.path.join(nodes, "0")); os.makedirs(os.path.join(nodes, "1"))
    for n in (0, 1):
        open(os.path.join(nodes, str(n), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
    for k, (numa, cpus) in enumerate(gpus):
        d = os.path.join(nodes, str(2 + k)); os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + k}\nlocation_id {k}\n")
        dev = os.path.join(root, "class", "drm", f"renderD{128 + k}", "device"); os.makedirs(dev)
        open(os.path.join(dev, "numa_node"), "w").write(f"{numa}\n")
        open(os.path.join(dev, "local_cpulist"), "w").write(cpus + "\n")


def test_gpu_count_and_numa_pinning_come_from_sysfs_without_touching_the_gpu(tmp_path):
    """bench.py's launcher parent counts GPUs from the KFD topology (no HIP / torch.cuda call before the ranks are spawned) and
    every rank runs on the CPUs local to its GPU: eight GPUs on two NUMA nodes, as on an MI355X node."""
    root = str(tmp_path)
    _fake_sysfs(root, [(0, "0-47,96-143")] * 4 + [(1, "48-95,144-191")] * 4)
    nodes = shard.gpu_nodes(root)
    assert len(nodes) == 8 and [n["node"] for n in nodes] == list(range(2, 10))
    assert [n["numa_node"] for n in nodes] == [0] * 4 + [1] * 4
    assert shard.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    allowed = os.sched_getaffinity(0)
    want = shard.parse_cpulist(nodes[5]["cpulist"]) & allowed
    assert shard.pin_to_gpu_numa(5, root, apply=False) == want        # intersected with the cpuset this process may use
    assert shard.pin_to_gpu_numa(8, root, apply=False) == set()       # no such GPU: nothing changes
    assert shard.gpu_nodes(str(tmp_path / "nothing")) == []           # no KFD (this container): zero GPUs, no exception
    # by PCI address (what bench.py's ranks use: w2x_device_pci_bus_id of their own HIP device): the ordinal plays no part - a runtime that enumerates
    # the GPUs in another order than the KFD nodes still lands on its own GPU's CPUs; an address sysfs does not know falls back to the KFD order
    pci = os.path.join(root, "bus", "pci", "devices", "0000:c1:00.0"); os.makedirs(pci)
    cpus = sorted(allowed)
    open(os.path.join(pci, "local_cpulist"), "w").write(f"{cpus[0]}\n")
    assert shard.pin_to_gpu_numa(5, root, apply=False, pci_bus_id="0000:C1:00.0") == {cpus[0]}
    assert shard.pin_to_gpu_numa(5, root, apply=False, pci_bus_id="0000:ff:00.0") == want
    # really applying it keeps the process runnable (restore afterwards)
    _fake_sysfs(str(tmp_path / "b"), [(0, ",".join(str(c) for c in sorted(allowed)))])
    assert shard.pin_to_gpu_numa(0, str(tmp_path / "b")) == allowed
    os.sched_setaffinity(0, allowed)
