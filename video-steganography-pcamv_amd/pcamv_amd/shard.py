"""Closed-GOP sharding across the GPUs of one node (parity definition P1).

Every closed GOP is an independent unit (IDR resets the reference list; CQP removes the
rate-control coupling), so rank r of `world` simply owns GOPs r, r+world, ...; there is no
data-path collective.  What is exchanged is the per-GOP result (the record summary here, the NAL
bytes in a full encoder), gathered to rank 0 over torch.distributed (RCCL on the GPU box, gloo in
the CPU tests)."""


def gop_assignment(n_gops, world, rank):
    """GOP indices owned by `rank`: round-robin, so the per-rank load differs by at most one GOP."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_gops, world))


def gather_results(dist, local_results, n_gops, world, rank):
    """Gather {gop_index: payload} dicts to rank 0 and return them in GOP order (None elsewhere)."""
    if dist is None or world == 1:
        return [local_results[g] for g in range(n_gops)]
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(local_results, bucket, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    missing = [g for g in range(n_gops) if g not in merged]
    if missing:
        raise RuntimeError(f"GOPs {missing} were not produced by any rank")
    return [merged[g] for g in range(n_gops)]
