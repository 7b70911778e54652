"""Closed-GOP sharding across the GPUs of one node (parity definition P1).

Every closed GOP is an independent unit (IDR resets the reference list; CQP removes the
rate-control coupling), so rank r of `world` owns GOPs r, r+world, ...; there is no data-path
collective.  What is exchanged is the per-GOP result: a byte payload per GOP (the pass-1 records +
flip maps here -- the stand-in for the NAL bytes until an entropy coder exists on this side of the
boundary), gathered to rank 0 with tensor collectives only (sizes: all_gather of int64; bytes:
gather of padded uint8), so it runs on RCCL over xGMI without pickling through the host, and on
gloo in the CPU tests.
"""
import numpy as np


def gop_assignment(n_gops, world, rank):
    """GOP indices owned by `rank`: round-robin, so the per-rank load differs by at most one GOP."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_gops, world))


def pack_gop_payload(mbs, flip):
    """bytes of one GOP-frame result: the records (pcamv_mb_t array) followed by the flip map"""
    return np.ascontiguousarray(mbs).view(np.uint8).tobytes() + np.ascontiguousarray(flip, dtype=np.int8).tobytes()


def gather_payloads(dist, local, n_gops, world, rank, device=None):
    """local: {gop_index: bytes} of the GOPs this rank owns.  Returns the n_gops payloads in GOP order on rank 0
    (None elsewhere).  Collectives: one all_gather of the per-GOP sizes, one gather of the padded byte buffers."""
    import torch
    mine = gop_assignment(n_gops, world, rank)
    if sorted(local) != mine:
        raise ValueError(f"rank {rank} owns GOPs {mine}, got {sorted(local)}")
    if dist is None or world == 1:
        return [local[g] for g in range(n_gops)]
    dev = device if device is not None else torch.device("cpu")
    per_rank = (n_gops + world - 1) // world
    # operands are assembled on the host and moved once each: the device only sees the collectives themselves
    sizes = torch.tensor([len(local[g]) for g in mine] + [0] * (per_rank - len(mine)), dtype=torch.int64).to(dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [s.cpu() for s in all_sizes]
    cap = max(1, max(int(s.sum()) for s in all_sizes))
    mine_bytes = b"".join(local[g] for g in mine)
    buf = torch.frombuffer(bytearray(mine_bytes + bytes(cap - len(mine_bytes))), dtype=torch.uint8).to(dev)
    bucket = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, bucket, dst=0)
    if rank != 0:
        return None
    out = [None] * n_gops
    for r in range(world):
        data = bucket[r].cpu().numpy().tobytes()
        off = 0
        for k, g in enumerate(gop_assignment(n_gops, world, r)):
            n = int(all_sizes[r][k])
            out[g] = data[off:off + n]
            off += n
    missing = [g for g in range(n_gops) if out[g] is None]
    if missing:
        raise RuntimeError(f"GOPs {missing} were not produced by any rank")
    return out
