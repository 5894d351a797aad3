#!/usr/bin/env python3
"""Headline benchmark: output MPix/s (and frames/s) of 1080p -> 4K upscaling, swin_unet/art scale4 noise3 fp16,
batch 4, tile 256, blend 1/16 (BASELINE.json configs[2]) on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path (gather -> network on every tile batch -> blend/compose) over one synthetic
1920x1080 frame that is already resident in HBM; frames are independent, so ranks shard frames with no data-path
collective (weak scaling: every rank renders K frames).  Rank 0 prints ONE JSON line.

`value` is the resident figure (inputs in HBM when the timed region starts: the measurement contract of this build).  The same K
frames are then rendered by every rank through the reference's whole render() contract - host frame in, host frame out
(img2img_render.cpp:226,344; caller loop main.cpp:263-269) - as a renderSequence over page-locked buffers with the copies on
side streams; that PCIe-inclusive figure is the top-level `host_to_host` object of the same line (max over ranks, same barriers,
per-rank spread), never `value`.  At N > 1 it is the number that can fail to scale (N x 100 MB per frame into host memory);
the resident one touches no shared resource.

Every line says where it ran: `placement` holds, per rank, the PCI bus id of the GPU the rank's HIP runtime rendered on, the
number of DISTINCT GPUs behind the ranks and W2X_DEVICE_MAP.  `n_gpus` is that distinct count: ranks sharing a card (a
rehearsal) give `n_gpus` < `n_ranks` and a metric that starts with "REHEARSAL".  --host-rehearsal runs the whole rank protocol
(spawn / rendezvous / barriers / reductions / the one JSON line) with the engine calls replaced by sleeps: no GPU, no value.

Ranks never touch torch.cuda: libw2x.so links /opt/rocm's HIP runtime and PyTorch bundles its own, so the timing barrier
and the max-reduction run over gloo on CPU tensors (there is no data-path collective to put on RCCL).  With --gpus N > 1
and no launcher environment the script starts the N ranks itself (child processes, spawned before any GPU call).
--mode strips times the other multi-GPU mode: ONE frame split into tile-column strips, rank r renders strip r of N
(renderStrip), strong scaling.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

MODEL, SCALE, NOISE, BATCH, TILE, BLEND = "swin_unet/art", 4, 3, 4, 256, 0.0625
FRAME_W, FRAME_H = 1920, 1080
TTA = False
# --config N (1-based index into BASELINE.json configs; 3 = the headline the metric is quoted on; configs[0] is the CPU plumbing case of tests/)
CONFIGS = {
    2: dict(name="configs[1]", model="cunet/art", scale=2, noise=1, batch=4, tile=256, w=1920, h=1080, tta=False),
    3: dict(name="configs[2]", model="swin_unet/art", scale=4, noise=3, batch=4, tile=256, w=1920, h=1080, tta=False),
    4: dict(name="configs[3]", model="swin_unet/photo", scale=4, noise=3, batch=8, tile=400, w=1920, h=1080, tta=True),
    5: dict(name="configs[4]", model="swin_unet/art_scan", scale=4, noise=3, batch=16, tile=640, w=3840, h=2160, tta=False),
    # not a BASELINE configuration: the size the multi-process tests run this script at (tests/test_gpu_pipeline_bytes.py), 48 tiles of 64
    0: dict(name="test size (not a BASELINE config)", model="swin_unet/art", scale=4, noise=3, batch=2, tile=64, w=380, h=290, tta=False),
}
CONFIG_NAME = "configs[2]"
MFMA_F16_PEAK_TFLOPS = 2500.0        # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                # spec; ~6300 achievable (same guide)
OUT_MPIX = FRAME_W * SCALE * FRAME_H * SCALE / 1e6


def select_config(n: int) -> None:
    global MODEL, SCALE, NOISE, BATCH, TILE, FRAME_W, FRAME_H, TTA, OUT_MPIX, CONFIG_NAME
    c = CONFIGS[n]
    MODEL, SCALE, NOISE, BATCH, TILE, FRAME_W, FRAME_H, TTA, CONFIG_NAME = c["model"], c["scale"], c["noise"], c["batch"], c["tile"], c["w"], c["h"], c["tta"], c["name"]
    OUT_MPIX = FRAME_W * SCALE * FRAME_H * SCALE / 1e6


def kernel_source_sha(symbol: str):
    """sha256 (12 hex digits) of the source file + build recipe that produce a kernel symbol: roofline.traffic comes from rocprofv3 --pmc passes
    recorded in profiles/pmc_traffic.json; it is only quoted while the kernel it was measured on is the kernel that runs now."""
    import hashlib
    files = {"swin_attn96_kernel": "k_swinattn96.hip", "swin_attn192u_kernel": "k_swinattn192u.hip", "mlp96q_kernel": "k_mlp96q.hip",
             "mlp2q_kernel": "k_mlp2.hip", "mlp2_kernel": "k_mlp2.hip", "conv48_kernel": "k_conv48.hip", "compose_kernel": "k_prepost.hip", "gather_kernel": "k_prepost.hip",
             "pixgemm_kernel": "k_pixgemm.hip", "merge_kernel": "k_pixgemm.hip", "toimage_kernel": "k_pixgemm.hip", "conv3_kernel": "k_conv3.hip", "conv3h_kernel": "k_conv3h.hip",
             "stem_kernel": "k_stem.hip", "gemm_kernel": "k_gemm.hip"}
    f = files.get(symbol.split("<")[0])
    if not f:
        return None
    h = hashlib.sha256()
    # (kernels.h holds what the kernel files share: the output stores' cache policy, the gate arithmetic, the parameter blocks)
    for path in (os.path.join(ROOT, "waifu2x-tensorrt_amd", "csrc", f), os.path.join(ROOT, "waifu2x-tensorrt_amd", "csrc", "kernels.h"), os.path.join(ROOT, "waifu2x-tensorrt_amd", "Makefile")):
        try:
            h.update(open(path, "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:12]


def synthetic_frame(seed: int) -> np.ndarray:
    """u8 BGR 1080p frame: seeded low-frequency sinusoids + +-4 LSB noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:FRAME_H, 0:FRAME_W].astype(np.float32)
    img = np.full((FRAME_H, FRAME_W, 3), 127.0, np.float32)
    for k in range(8):
        fx, fy, ph = rng.uniform(0.002, 0.03), rng.uniform(0.002, 0.03), rng.uniform(0, 6.28, 3)
        img += 14.0 * np.sin(xx[..., None] * fx + yy[..., None] * fy + ph)
    img += rng.integers(-4, 5, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def cpu_baseline(work: str, tile_out: int) -> dict:
    """The oracle (CPU port of the path: fp32 ONNX executor on torch-CPU kernels + numpy tile pipeline) timed on this box's host cores on a
    bounded sample of the same workload: tiles of the same graph (exported at batch 1) through the network plus their pre / post work, extrapolated
    to the tiles of a frame.  Thread count: 32 and 64 threads (or all usable cores if fewer) are both tried on one tile, the faster setting is used and
    both timings are reported.  Why not all cores of a 256-CPU box: the graph is ~640 small operators per tile, which stop scaling at a few dozen
    threads, and with one thread per CPU the pool oversubscribes - measured once with every core (profiles/r4_final/bench_cpu_all_cores_probe.json):
    0.48 s per tile on 32 threads, 81.4 s on 256.  Three samples, the median is quoted."""
    import torch
    import synth_models as sm
    from oracle import onnx_exec, pipeline
    path = sm.model_path(os.path.join(work, "cpu_b1"), MODEL, SCALE, NOISE)
    if not os.path.exists(path):
        sm.export_onnx(sm.make_model(MODEL, SCALE, seed=1234 + NOISE), path, 1, TILE, dynamic=True)
    ex = onnx_exec.Executor(path)
    frame = synthetic_frame(0)
    n, ins, outs = pipeline.calculate_tiles(FRAME_W, FRAME_H, FRAME_W * SCALE, FRAME_H * SCALE, (TILE, TILE), (tile_out, tile_out), SCALE, (BLEND, BLEND))
    ov = int(round(tile_out * BLEND))
    w = pipeline.create_tile_weights((ov, ov), (tile_out, tile_out))

    def one_tile(k):
        tile = np.ascontiguousarray(pipeline.pad_roi(frame[..., ::-1], ins[k]))
        y = ex.run(pipeline.blob_from_tiles([tile]))
        o = pipeline.apply_weights(np.ascontiguousarray(y[0].transpose(1, 2, 0)), outs[k], FRAME_W * SCALE, FRAME_H * SCALE, w)
        pipeline.to_u8(o)

    host = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host
    tried = {}
    for th in sorted({min(32, usable), min(64, usable)}):
        torch.set_num_threads(th)
        one_tile(0)                                          # warm-up (thread pool, allocator)
        t0 = time.perf_counter(); one_tile(n // 2); tried[th] = time.perf_counter() - t0
    threads = min(tried, key=tried.get)
    torch.set_num_threads(threads)
    per = max(2, min(6, int(6.0 / max(tried[threads], 1e-3))))    # tiles per sample: about 6 s each
    picks = list(range(0, n, max(1, n // (3 * per))))[:3 * per]
    samples = []
    for sidx in range(3):
        mine = picks[sidx::3]
        t0 = time.perf_counter()
        for k in mine:
            one_tile(k)
        samples.append((time.perf_counter() - t0) / len(mine) * n)
    frame_s = sorted(samples)[1]
    # the second restatement of the network: C++ loops under OpenMP (oracle/cnet/onnx_net.cpp, SURVEY 8d's "C++/OpenMP fp32 CPU oracle"), same tiles, same
    # pre / post work; its loops are rows of convolutions and matrix products, which keep scaling where the operator-at-a-time executor does not
    cpp = None
    try:
        from oracle import cnet
        cex = cnet.Executor(path)
        ex_torch = ex
        cth = min(32, usable)                 # (its thread sweep is on record: 1.73 s per tile on 32 threads, 3.1 on 64, 7.2 on 128 - profiles/r4_final/bench.json)
        cex.threads = cth
        ex = cex
        one_tile(0)
        t0 = time.perf_counter(); one_tile(n // 2); cframe = (time.perf_counter() - t0) * n
        ex = ex_torch
        cpp = {"value": round(OUT_MPIX / cframe, 4), "unit": "MPix/s", "cores": cth,
               "sample": f"oracle/cnet (C++ loops, OpenMP) + numpy pipeline, 1 tile T={TILE} on {cth} threads after one warm-up tile, extrapolated to {n} tiles/frame ({cframe:.0f} s/frame); a second oracle, not the baseline"}
        cex.close()
    except Exception as e:                                       # the figure is an extra: the torch-operator leg above is the baseline on record
        cpp = {"error": str(e)[:200]}
    return {"value": round(OUT_MPIX / frame_s, 4), "unit": "MPix/s", "cores": threads, "host_cores": host, "usable_cores": usable, "kind": "port",
            "samples_mpix_per_s": [round(OUT_MPIX / v, 4) for v in samples], "cpp_loops": cpp,
            "seconds_per_tile_by_threads": {str(k): round(v, 3) for k, v in tried.items()},
            "sample": f"oracle (torch-CPU fp32 ONNX executor + numpy pipeline), 3 samples of {len(picks) // 3} tiles T={TILE} each on {threads} threads "
                      f"(host shows {host} CPUs, {usable} usable; one tile took " + ", ".join(f"{v:.2f} s on {k}" for k, v in tried.items()) + f" threads; 81.4 s on all 256 threads of such a box, profiles/r4_final/bench_cpu_all_cores_probe.json), "
                      f"median extrapolated to {n} tiles/frame ({frame_s:.0f} s/frame)"}


def rank_record(rank: int, local_rank: int, pci_bus_id, pinned) -> dict:
    """What one rank reports about where it runs (shard.certify turns the ranks' records into the line's `placement`)."""
    import socket
    return {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "pid": os.getpid(), "pci_bus_id": pci_bus_id, "cpus": len(pinned) or None}


def rehearsal_prefix(placement: dict) -> str:
    """'' for a run with one GPU per rank; otherwise the words that keep the line from reading like an N-GPU result."""
    if placement["one_gpu_per_rank"]:
        return ""
    return f"REHEARSAL ({placement['n_ranks']} ranks on {placement['distinct_gpus']} GPU{'' if placement['distinct_gpus'] == 1 else 's'}): "


def host_rehearsal(a, dist, rank: int, world: int, json_out, placement: dict) -> int:
    """--host-rehearsal: the protocol of the frames mode - barrier, K timed steps, barrier, max over ranks, per-rank spread, rank 0 prints the line - with a
    sleep where a rank would render.  Nothing is measured; the line carries value = null."""
    import shard
    shard.barrier(dist)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.002 * (1 + rank % 3))
    shard.barrier(dist)
    wall = time.perf_counter() - t0
    wall_max = shard.max_over_ranks(wall, dist)
    walls = shard.gather_objects(wall, dist)
    frames = shard.gather_objects(shard.frames_for_rank(a.steps * world, rank, world), dist)
    if rank == 0:
        line = {"metric": rehearsal_prefix(placement) + "host rehearsal of the rank protocol: no GPU work, nothing measured", "value": None, "unit": "MPix/s",
                "n_gpus": placement["distinct_gpus"], "n_ranks": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall_max * 1e3 / a.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": None, "data": "none",
                "config": {"workload": "none (sleeps)", "mode": a.mode, "frames_rendered": sorted(f for fs in frames for f in fs),
                           "per_rank_ms_per_step": shard.spread([w * 1e3 / a.steps for w in walls])},
                "placement": placement}
        print(json.dumps(line), file=json_out, flush=True)
    json_out.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


def bench_shards_one_process(a) -> int:
    """--mode shards1p: ONE frame over --gpus N engines of THIS process with every tile computed once (w2x_render_sharded: contiguous tile ranges,
    seam bands copied device to device, each engine composes and downloads its own canvas cells).  Host frame in -> host frame out per step,
    strong scaling.  No child ranks: the seam exchange is a device-to-device copy between engines of one process."""
    import __graft_entry__ as g
    import synth_models as sm
    import shard
    have = len(shard.gpu_nodes())
    if have < a.gpus and not os.environ.get("W2X_DEVICE_MAP"):
        raise SystemExit(f"--gpus {a.gpus} but this node shows {have} GPU(s); refusing to report n_gpus != requested")
    pkg = g.package()
    work = os.path.join(a.work, "shards")
    path = sm.model_path(work, MODEL, SCALE, NOISE)
    if not os.path.exists(path):
        sm.export_onnx(sm.make_model(MODEL, SCALE, seed=1234 + NOISE), path, BATCH, TILE, dynamic=True)
    engs = []
    for d in range(a.gpus):
        e = pkg.Img2Img()
        if not e.build(path, pkg.BuildConfig.fixed(BATCH, TILE, device=d)):
            raise SystemExit("build failed: " + e.last_error())
        if not e.load(path, pkg.RenderConfig(deviceId=d, batchSize=BATCH, height=TILE, width=TILE, scaling=SCALE, overlap=(BLEND, BLEND), tta=TTA)):
            raise SystemExit("load failed: " + e.last_error())
        engs.append(e)
    frame = synthetic_frame(0)
    out = engs[0].alloc_host((FRAME_H * SCALE, FRAME_W * SCALE, 3))
    pf = engs[0].alloc_host(frame.shape); pf[...] = frame
    whole = engs[0].render(frame)
    for _ in range(max(a.warmup, 3)):
        pkg.render_sharded(engs, pf, out)
    if not np.array_equal(out, whole):
        raise SystemExit("renderSharded and render disagree")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        pkg.render_sharded(engs, pf, out)
    wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        engs[0].render(pf, out)
    wall1 = time.perf_counter() - t0
    n = pkg.calculate_tiles(FRAME_W, FRAME_H, FRAME_W * SCALE, FRAME_H * SCALE, TILE, engs[0].output_tile_size, SCALE, (BLEND, BLEND))[0]
    parts = [pkg.shard_plan(FRAME_W, FRAME_H, FRAME_W * SCALE, FRAME_H * SCALE, TILE, engs[0].output_tile_size, SCALE, (BLEND, BLEND), p, a.gpus) for p in range(a.gpus)]
    fps = a.steps / wall
    import socket
    placement = shard.certify([rank_record(d, d, pkg.device_pci_bus_id(d), set()) for d in range(a.gpus)], os.environ.get("W2X_DEVICE_MAP"))      # (one record per ENGINE here)
    line = {"metric": rehearsal_prefix(placement) + f"upscaled MPix/s, one {FRAME_W}x{FRAME_H} frame over N engines ({MODEL} fp16)", "value": round(fps * OUT_MPIX, 2), "unit": "MPix/s",
            "n_gpus": placement["distinct_gpus"], "n_ranks": 1, "n_engines": a.gpus, "placement": placement,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall * 1e3 / a.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{CONFIG_NAME}: {MODEL} scale{SCALE} noise{NOISE} batch{BATCH} tile{TILE} fp16{' +TTA' if TTA else ''}, {FRAME_W}x{FRAME_H} frame, blend={BLEND} ({n} tiles); host frame in -> host frame out",
                       "mode": "shards1p", "parallelism": f"one frame in {a.gpus} contiguous tile ranges (renderSharded), every tile computed once, seam bands copied device to device, no collectives; one process drives all engines",
                       "tiles_per_engine": [c for _, c, _, _ in parts], "speedup_bound": round(n / max(c for _, c, _, _ in parts), 2),
                       "one_engine_render_ms": round(wall1 * 1e3 / a.steps, 3), "speedup_measured": round(wall1 / wall, 3)}}
    print(json.dumps(line), flush=True)
    for e in engs:
        e.close()
    return 0


def bench_shards_ranks(a, pkg, eng, dist, rank, local_rank, world, json_out, pinned, placement) -> int:
    """--mode shards: ONE frame over N ranks, one process per GPU, every tile computed once.  Per step: rank r computes its contiguous range of the tile
    order into its slab (w2x_shard_compute), the ranks exchange the IPC handles of their slabs (a gloo all_gather of 64 bytes each - also the barrier that says
    every slab is complete), rank r opens the slabs of the parts in front of it, copies their seam bands device to device, composes its canvas cells and
    writes them into the frame all ranks share (a /dev/shm mapping), and a closing barrier keeps a rank from starting the next frame while a neighbour still
    reads its slab.  No collective on the data path."""
    import shard
    frame = synthetic_frame(0)
    oh, ow = FRAME_H * SCALE, FRAME_W * SCALE
    # the frame all ranks write their cells into: named after rank 0's process, so that two launches at once never share (or unlink) one file
    shm = f"/dev/shm/w2x_shards_{placement['ranks'][0]['pid']}_{os.environ.get('MASTER_PORT', '0')}.u8"
    # where rank q's slab lives AS THIS PROCESS SEES IT: the local ordinal whose PCI address is rank q's GPU, -1 when that card is not a device of this
    # process (per-rank *_VISIBLE_DEVICES isolation): the seam bands then travel as whole slots by hipMemcpyDefault instead of peer copies
    mine = {}
    for d in range(64):
        bus = pkg.device_pci_bus_id(d)
        if bus is None:
            break
        mine.setdefault(bus, d)
    peer_device = [mine.get(r.get("pci_bus_id"), -1) for r in placement["ranks"]]
    if rank == 0:
        with open(shm, "wb") as f:
            f.truncate(oh * ow * 3)
    shard.barrier(dist)
    out = np.memmap(shm, np.uint8, "r+", shape=(oh, ow, 3))
    opened = {}                                   # handle bytes -> device pointer in this process

    def step():
        if not eng.shard_compute(frame, rank, world):
            raise SystemExit("shard_compute failed: " + eng.last_error())
        ptr, handle = eng.shard_slab_handle()
        if dist is not None:
            handles = [None] * world
            dist.all_gather_object(handles, handle)
        else:
            handles = [handle]
        slabs = [0] * world
        for q in range(rank):                     # only parts in front of this one can hold tiles it needs
            if handles[q] not in opened:
                opened[handles[q]] = pkg.ipc_open(handles[q], local_rank)
                if not opened[handles[q]]:
                    raise SystemExit(f"rank {rank}: hipIpcOpenMemHandle of rank {q}'s slab failed")
            slabs[q] = opened[handles[q]]
        if not eng.shard_finish(out, rank, world, slabs, devices=peer_device):
            raise SystemExit("shard_finish failed: " + eng.last_error())
        shard.barrier(dist)

    for _ in range(max(a.warmup, 2)):
        step()
    shard.barrier(dist)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    wall = shard.max_over_ranks(time.perf_counter() - t0, dist)
    rc = 0
    if rank == 0:
        whole = eng.render(frame)
        same = bool(np.array_equal(np.asarray(out), whole))
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.render(frame, whole)
        wall1 = time.perf_counter() - t0
        n = pkg.calculate_tiles(FRAME_W, FRAME_H, ow, oh, TILE, eng.output_tile_size, SCALE, (BLEND, BLEND))[0]
        parts = [pkg.shard_plan(FRAME_W, FRAME_H, ow, oh, TILE, eng.output_tile_size, SCALE, (BLEND, BLEND), p, world) for p in range(world)]
        fps = a.steps / wall
        line = {"metric": rehearsal_prefix(placement) + f"upscaled MPix/s, one {FRAME_W}x{FRAME_H} frame over N ranks ({MODEL} fp16)", "value": round(fps * OUT_MPIX, 2), "unit": "MPix/s",
                "n_gpus": placement["distinct_gpus"], "n_ranks": world, "placement": placement,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall * 1e3 / a.steps, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f16", "data": "synthetic",
                "config": {"workload": f"{CONFIG_NAME}: {MODEL} scale{SCALE} noise{NOISE} batch{BATCH} tile{TILE} fp16{' +TTA' if TTA else ''}, {FRAME_W}x{FRAME_H} frame, blend={BLEND} ({n} tiles); host frame in -> host frame out (a /dev/shm mapping shared by the ranks)",
                           "mode": "shards", "parallelism": f"one frame in {world} contiguous tile ranges, one process per GPU, every tile computed once; seam bands copied device to device out of the neighbours' slabs (hipIpc handles exchanged over gloo), no data-path collective",
                           "tiles_per_rank": [c for _, c, _, _ in parts], "speedup_bound": round(n / max(c for _, c, _, _ in parts), 2),
                           "one_rank_render_ms": round(wall1 * 1e3 / a.steps, 3), "speedup_measured": round(wall1 / wall, 3), "bytes_equal_render": same,
                           "rank0_cpus": len(pinned) or None, "peer_device_ordinals_seen_by_rank0": peer_device}}
        print(json.dumps(line), file=json_out, flush=True)
        if not same:
            print("[w2x] the sharded frame differs from render()", file=sys.stderr)
            rc = 1
    shard.barrier(dist)
    for p_ in opened.values():
        pkg.ipc_close(p_)
    del out
    if rank == 0:
        try:
            os.unlink(shm)
        except OSError:
            pass
    json_out.close()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()
    return rc


def spawn_ranks(a) -> int:
    """--gpus N without a launcher: start the N ranks as child processes (this parent never touches the GPU) and pass rank 0's
    JSON line through.  Children rendezvous over gloo on 127.0.0.1."""
    import socket
    import subprocess
    import shard
    have = len(shard.gpu_nodes())                    # KFD topology in sysfs: this parent makes no HIP / torch.cuda call at all
    if have < a.gpus and not os.environ.get("W2X_DEVICE_MAP") and not a.host_rehearsal:
        raise SystemExit(f"--gpus {a.gpus} but this node shows {have} GPU(s); refusing to report n_gpus != requested")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=7, help="the K-step timed region is run this many times back to back; the median region is quoted, all are listed")
    ap.add_argument("--mode", choices=["frames", "strips", "shards", "shards1p"], default="frames",
                    help="frames: every rank renders K whole frames (weak scaling, the headline); strips: rank r renders strip r of N of the same frame K times (strong scaling); "
                         "shards: one frame in N tile ranges with every tile computed once, one process per GPU, the seam bands copied out of the neighbours' slabs through IPC handles "
                         "(strong scaling); shards1p: the same with one process driving N engines")
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=3,
                    help="BASELINE.json configuration (1-based; 3 = configs[2], the one the metric is quoted on); the others emit the same JSON record for their workload; 0 = a test-sized frame")
    ap.add_argument("--cpu-baseline", action="store_true", help="time the CPU oracle for --config other than 3 as well (minutes for the 400 / 640 tiles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frame", choices=["synthetic", "flat", "noise"], default="synthetic",
                    help="diagnostic: frame content (flat = one grey level, noise = uniform random bytes); `value` is quoted on synthetic")
    ap.add_argument("--op-times", action="store_true", help="print HIP-event time per plan op (one frame) to stderr")
    ap.add_argument("--work", default=os.environ.get("W2X_BENCH_WORK", "/tmp/w2x_bench"))
    ap.add_argument("--host-rehearsal", action="store_true",
                    help="the rank protocol only (spawn or launcher environment, gloo rendezvous, barriers, max-reduction, placement records, the one JSON line) with every engine "
                         "call replaced by a sleep: runs without a GPU, prints value = null and n_gpus = 0.  What tests/test_shard_gloo.py runs at 2 and 8 ranks")
    a = ap.parse_args()
    select_config(a.config)

    if a.mode == "shards1p":
        if int(os.environ.get("RANK", "0")) != 0:       # under a launcher: rank 0 drives every engine, the other ranks have nothing to do
            return
        raise SystemExit(bench_shards_one_process(a))
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: refusing to report n_gpus != requested")

    # Rank 0's stdout carries exactly one line, the JSON record: native libraries print there too (gloo announces its peers on
    # std::cout), so file descriptor 1 points at stderr until that line is written.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import shard
    # one rank per GPU: run on the CPUs of that GPU's NUMA node (page-locked frame buffers are allocated after this and land there)
    dmap = os.environ.get("W2X_DEVICE_MAP")
    phys = int(dmap.split(",")[local_rank]) if dmap and local_rank < len(dmap.split(",")) else local_rank
    import torch
    pkg = None
    if not a.host_rehearsal:
        import __graft_entry__ as g
        import synth_models as sm
        pkg = g.package()
    pinned = set()
    if pkg and (world > 1 or os.environ.get("W2X_PIN_NUMA")):
        # the CPUs of THIS rank's GPU by its PCI address as the HIP runtime reports it (no assumption about ordinal orders); nothing is allocated yet
        pinned = shard.pin_to_gpu_numa(phys, pci_bus_id=pkg.device_pci_bus_id(local_rank))

    dist = None
    if world > 1:
        # a rank uses torch for the ONNX export and gloo only (the CPU baseline runs at N = 1): a small fixed team - the default, one thread per CPU of the
        # NUMA node, spends a minute spinning in that export when ranks share a node's CPUs (two ranks on one GPU: 95 s -> see tests/test_gpu_pipeline_bytes.py)
        torch.set_num_threads(8)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")              # CPU tensors only: PyTorch's HIP runtime is never initialised beside /opt/rocm's

    # where every rank runs: gathered once, printed with the line (shard.certify)
    placement = shard.certify(shard.gather_objects(rank_record(rank, local_rank, pkg.device_pci_bus_id(local_rank) if pkg else None, pinned), dist), dmap)
    if a.host_rehearsal:
        raise SystemExit(host_rehearsal(a, dist, rank, world, json_out, placement))

    work = os.path.join(a.work, f"rank{rank}")
    path = sm.model_path(work, MODEL, SCALE, NOISE)
    if not os.path.exists(path):
        sm.export_onnx(sm.make_model(MODEL, SCALE, seed=1234 + NOISE), path, BATCH, TILE, dynamic=True)
    eng = pkg.Img2Img()
    bc = pkg.BuildConfig.fixed(BATCH, TILE, device=local_rank)
    if not eng.build(path, bc):
        raise SystemExit("build failed: " + eng.last_error())
    rc = pkg.RenderConfig(deviceId=local_rank, batchSize=BATCH, height=TILE, width=TILE, scaling=SCALE, overlap=(BLEND, BLEND), tta=TTA)
    if not eng.load(path, rc):
        raise SystemExit("load failed: " + eng.last_error())
    if rank == 0:
        for _, m in eng.messages[-2:]:
            print("[w2x] " + m, file=sys.stderr)

    if a.mode == "shards":
        raise SystemExit(bench_shards_ranks(a, pkg, eng, dist, rank, local_rank, world, json_out, pinned, placement))
    strips = a.mode == "strips"
    my_frames = [0] if strips else shard.frames_for_rank(a.steps * world, rank, world)   # frame f -> rank f mod N; each rank renders K frames
    frame = synthetic_frame(my_frames[0])
    if a.frame == "flat":
        frame = np.full_like(frame, 127)
    elif a.frame == "noise":
        frame = np.random.default_rng(my_frames[0]).integers(0, 256, frame.shape, dtype=np.uint8)
    out = np.empty((FRAME_H * SCALE, FRAME_W * SCALE, 3), np.uint8)
    part, parts = (rank, world) if strips else (0, 1)

    def render_once():                       # uploads the frame (it stays resident for the timed steps) and sets the strip
        ok = eng.render_strip(frame, out, part, parts) if strips else eng.render(frame, out)
        if not ok:
            raise SystemExit("render failed: " + eng.last_error())
    for _ in range(3):                       # the second call of a frame size captures its hipGraph; time a call in steady state
        render_once()
    t0 = time.perf_counter(); render_once(); pcie_ms_one = (time.perf_counter() - t0) * 1e3

    def sync_all():                          # every engine call returns after its stream has drained, so the device is idle here
        shard.barrier(dist)

    # ---- the timed region of `value`: K resident steps on every rank, between barriers, max over ranks - run R times back to back in this one
    #      invocation (--repeats, default 7); the line quotes the MEDIAN region and lists them all (ms_per_step_samples).  Why: one region of 20 frames is
    #      0.15 s on a box whose clocks differ by +-3 % from the next one's and drift within a run; a round's kernel work is worth less than that.
    eng.bench_resident(max(a.warmup, 1))
    region_walls, region_ms, region_ranks = [], [], []
    for _ in range(max(a.repeats, 1)):
        sync_all()
        t0 = time.perf_counter()
        ms_r = eng.bench_resident(a.steps)    # K frames, HIP events on the compute stream + stream sync inside
        sync_all()
        wall_r = time.perf_counter() - t0
        if ms_r <= 0:
            raise SystemExit("bench failed: " + eng.last_error())
        region_walls.append(shard.max_over_ranks(wall_r, dist)); region_ms.append(ms_r); region_ranks.append(shard.gather_objects(wall_r, dist))
    mid = sorted(range(len(region_walls)), key=lambda k: region_walls[k])[len(region_walls) // 2]     # the median region (upper median for even R)
    wall_max, ms, walls = region_walls[mid], region_ms[mid], region_ranks[mid]

    # ---- the same K frames through the whole render() contract (host buffer in, host buffer out), copies overlapped on side
    # streams (renderSequence over page-locked buffers, a ring of 3 output frames); every rank, same barriers, max over ranks
    full_wall_max, full_wall, full_regions = None, None, []
    if not strips:
        try:
            pf = eng.alloc_host(frame.shape); pf[...] = frame                    # page-locked frame buffers owned by the engine
            ring = [eng.alloc_host(out.shape) for _ in range(3)]
            eng.render_sequence([pf] * max(a.warmup, 3), outs=[ring[k % 3] for k in range(max(a.warmup, 3))])
            full_rank_walls = []
            for _ in range(max(a.repeats, 1)):               # the same R regions, the median quoted
                sync_all()
                t0 = time.perf_counter()
                eng.render_sequence([pf] * a.steps, outs=[ring[k % 3] for k in range(a.steps)])
                sync_all()
                fw = time.perf_counter() - t0
                full_regions.append(shard.max_over_ranks(fw, dist)); full_rank_walls.append(fw)
            fmid = sorted(range(len(full_regions)), key=lambda k: full_regions[k])[len(full_regions) // 2]
            full_wall, full_wall_max = full_rank_walls[fmid], full_regions[fmid]
            if not np.array_equal(ring[(a.steps - 1) % 3], out):
                raise RuntimeError("renderSequence and render disagree")
            for hb in [pf] + ring:
                eng.free_host(hb)
        except Exception as e:
            if world > 1:
                raise
            print(f"[w2x] full-path measurement skipped: {e}", file=sys.stderr)
        render_once()                        # restore the single-frame state profile_frame() works on
    full_walls = shard.gather_objects(full_wall, dist)

    prof = eng.profile_frame()
    desc = pkg.describe_plan(path, eng.pass_tiles, TILE).splitlines()[2:]
    op_ms = eng.op_times()
    # an op the engine folded into a neighbour's launch has no launch of its own and reports 0 ms: the image head riding on the last C = 96 MLP
    # (engine.cpp fuse_head; its time is inside the previous op's) and the stems / transposed convolutions computed by the convolution behind them (fuse_stem, fuse_up; inside the next op's)
    folded = [i for i, (line, t) in enumerate(zip(desc, op_ms)) if t == 0.0 and i > 0 and " gemm " in f" {line} " and " mlp " in f" {desc[i - 1]} "]
    folded_fwd = [i for i, (line, t) in enumerate(zip(desc, op_ms)) if t == 0.0 and i + 1 < len(desc) and " gemm " in f" {line} " and " gemm " in f" {desc[i + 1]} " and i not in folded]
    if a.op_times and rank == 0:
        for i, (line, t) in enumerate(zip(desc, op_ms)):
            note = "  (no launch of its own: folded into the previous op's launch, whose time includes it)" if i in folded else \
                   "  (no launch of its own: computed inside the next op's launch, whose time includes it)" if i in folded_fwd else ""
            print(f"{t:8.3f} ms  {line[:150]}{note}", file=sys.stderr)
    if rank == 0:
        import re
        fps = a.steps * (1 if strips else world) / wall_max       # strips: the N ranks together produce each frame
        n_tiles = pkg.calculate_tiles(FRAME_W, FRAME_H, FRAME_W * SCALE, FRAME_H * SCALE, TILE, eng.output_tile_size, SCALE, (BLEND, BLEND))[0]
        frame_tiles = n_tiles
        if strips:                               # the kernel figures below are rank 0's: its strip's tiles
            n_tiles = pkg.strip_plan(FRAME_W, FRAME_H, FRAME_W * SCALE, FRAME_H * SCALE, TILE, eng.output_tile_size, SCALE, (BLEND, BLEND), 0, world)[1]
        n_tiles *= 8 if TTA else 1               # network steps: a tile goes through the network once per augmentation
        live = n_tiles / eng.pass_tiles          # the zero-pad slots of the last batch are not computed (plan FLOPs are per pass of pass_tiles)
        # dominant kernel of the frame: plan ops grouped by the kernel that serves them, by summed HIP-event time
        # (measured on the compute stream by w2x_profile_frame / w2x_op_times)
        symbols = {("swinattn", 96): "swin_attn96_kernel", ("swinattn", 192): "swin_attn192u_kernel",
                   ("mlp", 96): "mlp96q_kernel", ("mlp", 192): "mlp2_kernel<192,2>"}     # (names as tools/pmc_traffic.py keys them)
        groups = {}
        for line, t in zip(desc, op_ms):
            m = re.match(r"\s*\d+ (\w+) (.*?)flops=(\d+)", line)
            if not m: continue
            kind, rest, flops = m.group(1), m.group(2), int(m.group(3)) * live
            cm = re.search(r"\bC=(\d+)", rest)
            key = (kind, int(cm.group(1))) if kind in ("swinattn", "mlp") and cm else (kind, 0)
            am = re.search(r"\ba=(\w+) M=(\d+) K=(\d+) N=(\d+)", rest)
            nbytes = 0.0
            if kind == "gemm" and am and CONFIG_NAME != "configs[2]":
                # other configs (cunet's convolutions): one group per operand shape, so that the dominant KERNEL is named - 3x3 convolutions over
                # 32-channel chunks run on conv3_kernel; algorithmic bytes = read the input map once + write the output map once
                mode, M, K, N = am.group(1), int(am.group(2)), int(am.group(3)), int(am.group(4))
                taps = 9 if "3x3" in mode else 4 if "2x2" in mode else 1
                key = (kind, f"{mode} {K // taps}->{N}")
                nbytes = float(M) * live * (K // taps + N) * 2
            if key[0] == "swinattn":    # read x + write y: tiles * windows * 36 tokens * C * 2 B, each way
                nbytes = 2.0 * n_tiles * int(re.search(r"nwin=(\d+)", rest).group(1)) * 36 * key[1] * 2
            elif key[0] == "mlp":
                nbytes = 2.0 * int(re.search(r"M=(\d+)", rest).group(1)) * live * key[1] * 2
            g = groups.setdefault(key, [0.0, 0, 0.0, 0.0]); g[0] += t; g[1] += 1; g[2] += flops; g[3] += nbytes
        def kernel_roof(key):                    # the roofline figures of one kernel group
            k_ms, k_n, k_flop, k_bytes = groups[key]
            tflops = k_flop / (k_ms * 1e-3) / 1e12
            gbs = k_bytes / (k_ms * 1e-3) / 1e9
            t_mfma, t_hbm = k_flop / (MFMA_F16_PEAK_TFLOPS * 1e12), k_bytes / (HBM_PEAK_GBS * 1e9)
            hbm_bound = t_hbm > t_mfma           # the roofline that binds this kernel = the larger of the two floor times
            return {"bound": "hbm" if hbm_bound else "mfma",
                    "achieved": round(gbs if hbm_bound else tflops, 2), "peak": HBM_PEAK_GBS if hbm_bound else MFMA_F16_PEAK_TFLOPS,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tflops / MFMA_F16_PEAK_TFLOPS), 5), "traffic": None,
                    "kernel": symbols.get(key, ("conv3_kernel" if isinstance(key[1], str) and key[1].startswith("conv3x3s1") and int(key[1].split()[1].split("->")[0]) % 32 == 0 else
                                                "gemm_kernel / pixgemm kernels") if key[0] == "gemm" else key[0]),
                    "plan_ops": key[1] if isinstance(key[1], str) else None,
                    "launches_per_frame": k_n, "avg_launch_us": round(k_ms * 1e3 / k_n, 2),
                    "algorithmic_gflop_per_launch": round(k_flop / k_n / 1e9, 3),
                    "algorithmic_mbyte_per_launch": round(k_bytes / k_n / 1e6, 3),
                    "other_roof": {"unit": "TFLOP/s" if hbm_bound else "GB/s", "achieved": round(tflops if hbm_bound else gbs, 2),
                                   "frac": round((tflops / MFMA_F16_PEAK_TFLOPS) if hbm_bound else (gbs / HBM_PEAK_GBS), 5)}}
        ranked = sorted(groups, key=lambda k: -groups[k][0])
        dom = ranked[0]
        roof = kernel_roof(dom)
        roof["launch_note"] = ("HIP events around each launch with every pass in one piece on one stream (the engine's profiling pass, = W2X_GROUPS=1: a launch covers all live tiles); "
                               "the timed region of `value` runs each pass as two tile groups on two streams")
        roof["kernels_ms_per_frame"] = {symbols.get(k, k[0] if not isinstance(k[1], str) else k[0] + " " + k[1]): round(v[0], 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0])}
        # config 3's two attention kernels take 2.0 ms per frame each and swap places from box to box: a kernel within a tenth of the dominant one is shown beside it
        if len(ranked) > 1 and groups[ranked[1]][0] >= 0.9 * groups[dom][0]:
            roof["runner_up"] = kernel_roof(ranked[1])
        notes = []
        if folded:
            notes.append("mlp96q_kernel's figure includes the image head (Linear 96 -> 64, Clip, DepthToSpace), which the engine folds into the last C = 96 MLP launch; that plan op has no launch and reports 0 ms")
        if folded_fwd:
            names = [re.sub(r"^.*\[(.*)\]\s*$", r"\1", desc[i]) for i in folded_fwd]
            notes.append("computed inside the NEXT op's launch and counted with it (the plan op has no launch and reports 0 ms; the launch's roofline figures price its own op only): " + ", ".join(names) +
                         " - a stem convolution in the halo stage of the convolution behind it (conv48_kernel<true> / conv3_kernel<false, 1>), cunet's transposed convolutions in the halo stage of "
                         "the 64 -> 64 convolutions behind them (conv3_kernel<false, 2>)")
        if notes:
            roof["kernels_note"] = "; ".join(notes)
        tr = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/profile_round.sh)
        if os.path.exists(tr) and CONFIG_NAME == "configs[2]":
            try:
                table = json.load(open(tr))
                for r in [roof] + ([roof["runner_up"]] if "runner_up" in roof else []):
                    t = table.get(r["kernel"])
                    now = kernel_source_sha(r["kernel"])
                    if t and t.get("source_sha") and t["source_sha"] == now:
                        r["traffic"] = t["bytes_per_launch"]; r["traffic_source"] = t["source"]; r["traffic_kernel_source_sha"] = now
                        for k_ in ("issue", "mfma_busy", "coexec"):      # SQ counters of the same round's passes (tools/pmc_traffic.py), same source-hash rule
                            if k_ in t: r[k_] = t[k_]
                    elif t:          # the counters were taken on another revision of this kernel: not quoted
                        r["traffic_note"] = f"profiles/pmc_traffic.json holds {t['bytes_per_launch']} bytes per launch measured on kernel source {t.get('source_sha', 'unrecorded')}; the kernel now built is {now}: not quoted"
            except Exception:
                pass
        line = {
            "metric": rehearsal_prefix(placement) + ("upscaled MPix/s, 1080p->4K swin_unet/art fp16" if CONFIG_NAME == "configs[2]" else f"upscaled MPix/s, {FRAME_W}x{FRAME_H} x{SCALE} {MODEL} fp16{' +TTA' if TTA else ''}"),
            "value": round(fps * OUT_MPIX, 2), "unit": "MPix/s", "n_gpus": placement["distinct_gpus"], "n_ranks": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(wall_max * 1e3 / a.steps, 3), "higher_is_better": True, "scaling": "strong" if strips else "weak", "vs_baseline": None,
            # every timed region of this invocation (K steps each, max over ranks); `value` / `ms_per_step` are the median one
            "ms_per_step_samples": [round(w * 1e3 / a.steps, 3) for w in region_walls], "repeats": len(region_walls),
            "dtype": "f16", "data": "synthetic" if a.frame == "synthetic" else "synthetic (" + a.frame + " frame: diagnostic)",
            "config": {"workload": f"{CONFIG_NAME}: {MODEL} scale{SCALE} noise{NOISE} batch{BATCH} tile{TILE} fp16{' +TTA' if TTA else ''}, {FRAME_W}x{FRAME_H} frame, blend={BLEND} "
                                   f"({frame_tiles} tiles, {-(-(frame_tiles * (8 if TTA else 1)) // BATCH)} batches); synthetic-weight graph of that architecture, frame resident in HBM",
                       "frames_per_s": round(fps, 3), "device_ms_per_frame": round(ms, 3), "frames_per_rank": a.steps,
                       "parallelism": (f"one frame in {world} tile-column strips (renderStrip), no collectives" if strips else f"frame-sharded x{world}, no collectives") + "; timing barrier over gloo",
                       "mode": a.mode, "rank0_cpus": len(pinned) or None,
                       "per_rank_resident_ms_per_step": shard.spread([w * 1e3 / a.steps for w in walls]),
                       "render_call_ms_pageable": round(pcie_ms_one, 2),
                       "full_path_ms_per_frame": None if full_wall_max is None else round(full_wall_max * 1e3 / a.steps, 3),
                       "full_path_frames_per_s": None if full_wall_max is None else round(a.steps * world / full_wall_max, 3),
                       "full_path_mpix_per_s": None if full_wall_max is None else round(a.steps * world / full_wall_max * OUT_MPIX, 2),
                       "full_path_note": "host frame in -> host frame out for all K frames on every rank (renderSequence over engine-allocated page-locked buffers, H2D/D2H on side streams); max over ranks",
                       "tiles_per_network_pass": eng.pass_tiles,
                       "tile_groups_per_pass": int(os.environ.get("W2X_GROUPS", "2")),
                       "algorithmic_tflop_per_frame": round(eng.plan_flops * live / 1e12, 4),
                       "families_ms_per_frame": {k: round(v[0], 3) for k, v in prof.items() if k != "frame_ms"}},
            "roofline": roof,
            "placement": placement,
            # the path the reference's render() is (img2img_render.cpp:226,344): host frame in -> host frame out, PCIe included.  Not `value` (docstring).
            "host_to_host": None if full_wall_max is None else {
                "ms_per_frame": round(full_wall_max * 1e3 / a.steps, 3), "frames_per_s": round(a.steps * world / full_wall_max, 3),
                "mpix_per_s": round(a.steps * world / full_wall_max * OUT_MPIX, 2), "unit": "MPix/s",
                "per_rank_ms_per_frame": shard.spread([None if w is None else w * 1e3 / a.steps for w in full_walls]),
                "ms_per_frame_samples": [round(w * 1e3 / a.steps, 3) for w in full_regions],
                "vs_resident": round(wall_max / full_wall_max, 4),
                "one_synchronous_render_call_ms": round(pcie_ms_one, 2),
                "how": "all K frames of every rank through renderSequence over engine-allocated page-locked buffers, H2D / D2H on side streams; same barriers, max over ranks"},
        }
        if not a.no_cpu_baseline and world == 1 and (a.config == 3 or a.cpu_baseline):
            line["cpu_baseline"] = cpu_baseline(a.work, eng.output_tile_size)
        print(json.dumps(line), file=json_out, flush=True)
    json_out.close()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
