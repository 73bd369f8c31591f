"""Slide-sharded data parallelism: one process per GPU, slides are independent units
(the slide reduce never crosses slides, ``threshold.py:191-192``), so the only
communication is ONE all-gather of the per-slide results at the end (RCCL over xGMI on
the GPU box -- backend "nccl" is RCCL on ROCm; gloo in CPU tests).  The reference is
single-process and has no counterpart; nothing here is translated from it.
"""
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


def init_from_env(device_type='cuda', backend=None, local_device=None, single_rank_group=False):
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
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
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
