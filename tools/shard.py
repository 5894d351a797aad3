"""Multi-GPU sharding helpers for the hot path: frames are independent units (tiles never communicate), so rank r owns
frames r, r+N, r+2N, ... and runs the full single-GPU pipeline; the only collectives are the timing barrier and the
max-reduction of the per-rank wall time (no data-path collective)."""
from __future__ import annotations


def frames_for_rank(total_frames: int, rank: int, world: int) -> list[int]:
    """frame f -> rank f mod N."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, total_frames, world))


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX all-reduce of a python float (returns value itself when not distributed)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.cpu()[0])


def barrier(dist=None):
    if dist is not None and dist.is_initialized():
        dist.barrier()


def gather_objects(obj, dist=None) -> list:
    """Every rank's `obj`, in rank order, on every rank (a list of one when not distributed).  Host-side bookkeeping only."""
    if dist is None or not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def certify(records: list, device_map=None) -> dict:
    """What a bench line says about WHERE its ranks ran, from the records the ranks gathered (rank, local_rank, host, pci_bus_id as the HIP runtime of
    that rank reports it, pid): the number of distinct GPUs behind the ranks and whether that is one per rank.  A run whose ranks share a card
    (W2X_DEVICE_MAP rehearsals) or touch no card at all (--host-rehearsal) cannot then print a line that reads like an N-GPU result."""
    gpus = sorted({(r.get("host"), r["pci_bus_id"]) for r in records if r.get("pci_bus_id")})
    return {"ranks": records, "n_ranks": len(records), "distinct_gpus": len(gpus), "one_gpu_per_rank": len(gpus) == len(records),
            "device_map": device_map or None}


def spread(values: list) -> dict:
    """min / max / mean of the per-rank values of one quantity (None entries left out)."""
    v = [float(x) for x in values if x is not None]
    return {"min": round(min(v), 4), "max": round(max(v), 4), "mean": round(sum(v) / len(v), 4)} if v else None


# ---- node topology without touching the GPU (the launcher parent of bench.py must stay GPU-free: it spawns the ranks)
def gpu_nodes(sysfs: str = "/sys") -> list[dict]:
    """GPUs of this node from the KFD topology (/sys/class/kfd/kfd/topology/nodes/*/properties: a node with simd_count > 0 is a
    GPU), in node order = HIP ordinal order when no *_VISIBLE_DEVICES filter is set.  Each entry: {"node", "render_minor",
    "numa_node", "cpulist"}; the last two come from the device's DRM render node (its PCIe root's NUMA node and local CPUs)."""
    import os
    root = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    out = []
    try:
        ids = sorted((int(n) for n in os.listdir(root) if n.isdigit()))
    except OSError:
        return out
    for n in ids:
        props = {}
        try:
            for line in open(os.path.join(root, str(n), "properties")):
                k, _, v = line.strip().partition(" ")
                props[k] = v.strip()
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1") or -1)
        numa, cpus = -1, ""
        dev = os.path.join(sysfs, "class", "drm", f"renderD{minor}", "device")
        try:
            numa = int(open(os.path.join(dev, "numa_node")).read().strip())
            cpus = open(os.path.join(dev, "local_cpulist")).read().strip()
        except (OSError, ValueError):
            pass
        out.append({"node": n, "render_minor": minor, "numa_node": numa, "cpulist": cpus})
    return out


def parse_cpulist(text: str) -> set[int]:
    """'0-3,8,10-11' -> {0,1,2,3,8,10,11} (the kernel's cpulist format)."""
    cpus: set[int] = set()
    for part in text.replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def pin_to_gpu_numa(device: int, sysfs: str = "/sys", apply: bool = True, pci_bus_id: str | None = None) -> set[int]:
    """Restrict this process to the CPUs local to GPU `device` (its PCIe root's NUMA node), intersected with what the process may
    use already (cgroup cpusets).  Page-locked frame buffers allocated afterwards then sit on the GPU's own NUMA node: with one
    rank per GPU every rank moves 100 MB per 4K frame over PCIe.  Returns the CPU set applied (empty: nothing known, nothing
    changed).  pci_bus_id: the device's PCI address as the HIP runtime of THIS process reports it (w2x_device_pci_bus_id) - then the
    CPU list comes from /sys/bus/pci/devices/<id>/local_cpulist and does not rest on HIP ordinals following the KFD node order
    (they do not under *_VISIBLE_DEVICES filters or a reordering runtime); without it the KFD order is assumed."""
    import os
    cpulist = ""
    if pci_bus_id:
        try:
            cpulist = open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "local_cpulist")).read().strip()
        except OSError:
            cpulist = ""
    if not cpulist:
        nodes = gpu_nodes(sysfs)
        if not (0 <= device < len(nodes)) or not nodes[device]["cpulist"]:
            return set()
        cpulist = nodes[device]["cpulist"]
    want = parse_cpulist(cpulist)
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return set()
    cpus = want & allowed
    if not cpus:
        return set()
    if apply:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            return set()
    return cpus
