"""Slide-sharded data parallelism: one process per GPU, slides are independent units
(the slide reduce never crosses slides, ``threshold.py:191-192``), so the only
communication is ONE all-gather of the per-slide results at the end (RCCL over xGMI on
the GPU box -- backend "nccl" is RCCL on ROCm; gloo in CPU tests).  The reference is
single-process and has no counterpart; nothing here is translated from it.
"""
import datetime
import os

import numpy as np
import torch
import torch.distributed as dist


def partition_slides(tile_counts, world_size):
    """Deterministic longest-processing-time partition of slides over ranks.

    Returns a list (per rank) of slide indices in ascending order.  Every rank computes the
    same partition from the same counts, so no communication is needed to agree on it.

    LPT decides how MANY slides of every tile count a rank gets; slides of one count are interchangeable, so
    they are handed out in index order -- rank 0 the first of them, rank 1 the next ... -- instead of round-robin.
    Loads are exactly LPT's; a rank's slides (and with them its global tile indices) come out as contiguous as the
    counts allow: for BASELINE config 3's 1 600 equal slides, blocks of 200, so a batch that
    spans two of a rank's slides is one run of consecutive tile indices and needs one head launch."""
    counts = np.asarray(tile_counts, dtype=np.int64)
    order = np.lexsort((np.arange(len(counts)), -counts))     # by -count, then index
    load = np.zeros(world_size, dtype=np.int64)
    quota = [dict() for _ in range(world_size)]               # rank -> {tile count: number of slides}
    for s in order:
        r = int(np.argmin(load))                              # ties -> lowest rank
        c = int(counts[s])
        quota[r][c] = quota[r].get(c, 0) + 1
        load[r] += c
    parts = [[] for _ in range(world_size)]
    for c in sorted({int(x) for x in counts}, reverse=True):
        ids = [int(i) for i in np.flatnonzero(counts == c)]    # ascending
        k = 0
        for r in range(world_size):
            n = quota[r].get(c, 0)
            parts[r] += ids[k:k + n]
            k += n
    return [sorted(p) for p in parts]


def global_tile_offsets(tile_counts):
    """First global tile index of every slide (dataset order): the Philox tile counter, so
    dropout masks do not depend on how slides are sharded or batched."""
    counts = np.asarray(tile_counts, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(counts)[:-1]]) if len(counts) else counts


def _cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> sorted list of ints."""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out += list(range(int(a), int(b or a) + 1))
    return sorted(set(out))


def gpu_numa_node(device_index):
    """NUMA node of HIP device ``device_index`` from sysfs (``/sys/bus/pci/devices/<domain:bus:dev.fn>/numa_node``), or None when
    the platform does not say (a single-socket box, a container without that file, -1)."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        node = int(open(f'/sys/bus/pci/devices/{bdf}/numa_node').read())
        return node if node >= 0 else None
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def rank_cores(local_rank, local_world, allowed=None, numa_cpus=None, numa_peers=None):
    """The CPU cores one of ``local_world`` ranks on a node should confine itself to: its share of ``allowed`` (default: the
    process's affinity mask) -- of the cores of its GPU's NUMA node when those are known (``numa_cpus``; ``numa_peers`` = (index
    among, number of) the local ranks whose GPUs sit on the same node), of everything otherwise -- as a contiguous slice, never
    empty.  Pure function of its arguments (tested on CPU); ``pin_rank`` applies it."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    pool, k, n = allowed, int(local_rank), max(1, int(local_world))
    if numa_cpus:
        near = [c for c in allowed if c in set(numa_cpus)]
        if near:
            pool = near
            k, n = numa_peers if numa_peers else (k, n)
    n = max(1, min(n, len(pool)))
    k = k % n
    lo, hi = k * len(pool) // n, (k + 1) * len(pool) // n
    return pool[lo:hi] or pool[:1]


def pin_rank(local_rank, local_world, device_index=None):
    """Confine this rank to its share of the node's cores, next to its GPU where the platform tells (call BEFORE any thread pool
    exists: torch's intra-op pool and libbiscuit_io's decoder threads inherit the mask of the thread that creates them).  Eight
    ranks on one node otherwise each start min(cores, 16) decoder threads on the same cores.  Returns the cores."""
    if not hasattr(os, 'sched_setaffinity') or local_world <= 1:
        return sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    numa_cpus = peers = None
    if device_index is not None and torch.cuda.is_available():
        node = gpu_numa_node(device_index)
        if node is not None:
            try:
                numa_cpus = _cpulist(open(f'/sys/devices/system/node/node{node}/cpulist').read())
                ndev = torch.cuda.device_count()
                same = [d for d in range(min(ndev, local_world)) if gpu_numa_node(d) == node]
                if device_index in same:
                    peers = (same.index(device_index), len(same))
            except (OSError, ValueError):
                numa_cpus = None
    cores = rank_cores(local_rank, local_world, numa_cpus=numa_cpus, numa_peers=peers)
    try:
        os.sched_setaffinity(0, cores)
    except OSError:
        pass
    return cores


# Rendezvous and every collective give up after this long instead of the backend's default (NCCL / RCCL: ten minutes and more): a
# rank that died before the rendezvous must not leave the others waiting into the launcher's time limit
DIST_TIMEOUT_S = 120


def init_from_env(device_type='cuda', backend=None, local_device=None, single_rank_group=False, timeout_s=DIST_TIMEOUT_S):
    """Initialise torch.distributed from the launcher's rendezvous variables (torchrun's contract: RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT -- the only environment this package reads).

    backend: None = "nccl" (RCCL) for cuda, "gloo" for cpu; local_device: None = LOCAL_RANK.  Both are explicit
    arguments so that a caller -- a test running two ranks on one GPU -- states them itself.  single_rank_group: create the
    process group even for a world of one (a one-GPU box can then run the collectives through a real RCCL communicator;
    RCCL refuses two ranks on one device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0')) if local_device is None else int(local_device)
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = backend or ('nccl' if device_type == 'cuda' else 'gloo')
        if device_type == 'cuda':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(timeout_s)))
    elif device_type == 'cuda' and torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def gather_slide_results(local_ids, mean_pred, mean_unc, count, n_slides, cap, device=None):
    """All-gather every rank's per-slide (mean y_pred, mean uncertainty, tile count).

    local_ids: this rank's global slide indices; the three arrays are aligned with it.
    cap: max slides held by any rank -- every rank derives it from the same
    ``partition_slides`` result, so the buffer size needs no negotiation.
    Returns float64 arrays of length n_slides (NaN / 0 for slides nobody reported).
    Exactly ONE collective, on a fixed-size padded buffer [cap, 4] per rank
    (2.4-6.4 KB at 200 slides/rank: latency-bound, not link-bound)."""
    local_ids = np.asarray(local_ids, dtype=np.int64)
    world = dist.get_world_size() if dist.is_initialized() else 1
    out_pred = np.full(n_slides, np.nan)
    out_unc = np.full(n_slides, np.nan)
    out_cnt = np.zeros(n_slides, dtype=np.int64)
    if not dist.is_initialized():             # (a process group of one rank still takes the collective below)
        out_pred[local_ids] = np.asarray(mean_pred, dtype=np.float64)
        out_unc[local_ids] = np.asarray(mean_unc, dtype=np.float64)
        out_cnt[local_ids] = np.asarray(count, dtype=np.int64)
        return out_pred, out_unc, out_cnt
    k = len(local_ids)
    if k > cap:
        raise ValueError(f'rank holds {k} slides but cap is {cap}')
    dev = device if device is not None else (
        torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl'
        else torch.device('cpu'))
    buf = torch.full((cap, 4), -1.0, dtype=torch.float64)
    if k:
        buf[:k, 0] = torch.from_numpy(local_ids.astype(np.float64))
        buf[:k, 1] = torch.as_tensor(np.asarray(mean_pred, dtype=np.float64))
        buf[:k, 2] = torch.as_tensor(np.asarray(mean_unc, dtype=np.float64))
        buf[:k, 3] = torch.as_tensor(np.asarray(count, dtype=np.float64))
    buf = buf.to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    allb = torch.cat(parts, 0).cpu().numpy()
    valid = allb[:, 0] >= 0
    ids = allb[valid, 0].astype(np.int64)
    out_pred[ids] = allb[valid, 1]
    out_unc[ids] = allb[valid, 2]
    out_cnt[ids] = allb[valid, 3].astype(np.int64)
    return out_pred, out_unc, out_cnt
