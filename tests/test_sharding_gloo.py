"""N > 1 host path on CPU: two gloo ranks shard closed GOPs, each analyses its own GOPs (the CPU
oracle stands in for the GPU compute here), rank 0 gathers the per-GOP payloads (records + flip maps: variable
size) in GOP order with the tensor collectives bench.py uses over RCCL, and they must be byte-identical to the
single-process run."""
import hashlib
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GOPS = 5


def _gop_payload(g):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "video-steganography-pcamv_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import orc
    from pcamv_amd.synth import make_clip
    clip = make_clip(64, 48, 2, seed=100 + g, static_cols=16 * (g % 3))     # skipped macroblocks: the carrier count differs from GOP to GOP
    o = orc.Oracle(orc.make_params(64, 48, me="hex", subme=3, mv_range=64, tscale=0))
    o.set_ref(*clip[0]); o.set_fenc(*clip[1])
    mbs, _ = o.analyse_pframe(28, 1)
    emb = o.embed_pframe(mbs, 0.5)
    o.close()
    from pcamv_amd.shard import pack_gop_payload
    return pack_gop_payload(mbs, emb["flip"])


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "video-steganography-pcamv_amd"))
    from pcamv_amd.shard import gop_assignment, gather_payloads
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mine = gop_assignment(N_GOPS, world, rank)
    local = {g: _gop_payload(g) for g in mine}
    out = gather_payloads(dist, local, N_GOPS, world, rank)
    dist.barrier()
    if rank == 0:
        q.put((mine, out))
    dist.destroy_process_group()


def test_two_ranks_shard_gops_and_gather_in_order():
    sys.path.insert(0, os.path.join(ROOT, "video-steganography-pcamv_amd"))
    from pcamv_amd.shard import gop_assignment
    assert gop_assignment(5, 2, 0) == [0, 2, 4] and gop_assignment(5, 2, 1) == [1, 3]
    assert sorted(sum((gop_assignment(16, 8, r) for r in range(8)), [])) == list(range(16))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mine, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert mine == [0, 2, 4]
    single = [_gop_payload(g) for g in range(N_GOPS)]
    assert len({len(b) for b in single}) > 1, "the payloads should differ in size (flip maps follow the carrier count)"
    assert gathered == single
