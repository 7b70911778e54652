#!/usr/bin/env python3
"""bench.py -- 1080p macroblocks/s of the P-frame hot path (embedding on) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one P frame through the whole hot path (half-pel plane production, motion search +
partition decision, RCA replacement-MV costs, pass-1 reconstruction, cover/cost assembly,
syndrome-trellis embedding, then the reference's second pass: final MVs, reconstruction and loop
filter, whose output is the next step's reference -- `--open-loop` stops before the second pass and
searches against the previous source frame) for each of --gops independent closed GOPs resident on the GPU
(closed GOPs are the reference's natural sharding unit; inside a frame the raster
dependency leaves most of the chip idle, so one GPU advances many GOP pipelines together, each
kernel launch carrying the same dependency step of all of them).  Inputs (synthetic 1080p I420, SURVEY 8(d) generator) are resident in HBM before the
timed region.  N > 1: one process per GPU, GOPs sharded across ranks, no data-path collective
(weak scaling); a summary all_gather over RCCL runs after the timed region.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, plus `roofline` for the dominant
kernel (k_analyse_flow; k_search_diag under PCAMV_SCHED=diag) and `cpu_baseline` (the oracle port on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-steganography-pcamv_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gops", type=int, default=256, help="closed GOPs in flight per GPU (batched into every launch)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1088, help="coded height (1080 rounded up to 16)")
    ap.add_argument("--me", default="umh")
    ap.add_argument("--subme", type=int, default=7)
    ap.add_argument("--qp", type=int, default=26)
    ap.add_argument("--emrate", type=float, default=0.5)
    ap.add_argument("--open-loop", action="store_true", help="pass 1 only: no pass 2 / loop filter, the reference is the previous source frame")
    ap.add_argument("--closed-loop", action="store_true", help="(default) pass 2 + loop filter on the GPU, the deblocked reconstruction is the next reference")
    ap.add_argument("--host-io-steps", type=int, default=3, help="extra untimed-for-`value` steps that also move each frame's source pictures host->device (pinned) and its records + embedding vectors device->host, reported as `pcie_inclusive` (0 = skip; rank 0, N=1 only)")
    ap.add_argument("--cpu-frames", type=int, default=6, help="P frames timed for the CPU baseline (0 = skip)")
    args = ap.parse_args()
    args.closed_loop = not args.open_loop

    import numpy as np
    import torch
    import pcamv_amd
    from pcamv_amd.synth import make_clip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal on a one-GPU box only: PCAMV_BENCH_REHEARSE=1 puts every rank on GPU 0 and runs the (untimed)
    # collectives over gloo, since RCCL refuses two ranks on one device; the driver's runs use RCCL, one GPU per rank
    rehearse = os.environ.get("PCAMV_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if rehearse else dev          # where collective operands live

    W, H = args.width, args.height
    n_mb = (W // 16) * (H // 16)
    p = pcamv_amd.param_default(W, H)
    pcamv_amd.param_parse(p, "me", args.me)
    pcamv_amd.param_parse(p, "subme", args.subme)

    # synthetic clip: a few distinct frames per GOP, cycled; each GOP gets its own phase of the clip
    nfr = 6
    clip = make_clip(W, H, nfr, seed=13 + rank)
    dframes = [[torch.from_numpy(pl).to(dev) for pl in fr] for fr in clip]
    encs = [pcamv_amd.Encoder(p, device=local) for _ in range(args.gops)]
    batch = pcamv_amd.Batch(encs)          # all GOPs advance together: one launch per dependency step
    if args.closed_loop:
        batch.set_closed_loop(True)
    recon = [enc.recon_device() for enc in encs]
    started = [False]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def step(t):
        for g, enc in enumerate(encs):
            a = dframes[(t + g) % nfr]
            b = dframes[(t + g + 1) % nfr]
            # reference = previous source frame (open loop: the deblocked pass-2 reconstruction that
            # closes the loop in the encoder is produced by the host), chained MV field
            if args.closed_loop and started[0]:     # reference = this GOP's own deblocked reconstruction of the previous step
                enc.set_ref_device(recon[g][0], recon[g][1], recon[g][2], enc.PREV_INTERNAL, enc.PREV_INTERNAL)
            else:
                enc.set_ref_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), enc.PREV_INTERNAL, enc.PREV_INTERNAL)
            enc.set_fenc_device(b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())
        batch.step(args.qp, args.emrate, stream.cuda_stream)
        started[0] = True

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(args.warmup):
        step(t)
    barrier()
    batch.kernel_time(reset=True)
    prof = None
    if os.environ.get("PCAMV_PROF_DUMP") == "1":      # diagnostics build of the library (-DPCAMV_PROF): wave cycles per phase
        import ctypes
        prof = (ctypes.c_ulonglong * 32)()
        pcamv_amd.load_library().pcamv_gpu_prof_fetch(prof, 1)
    t0 = time.perf_counter()
    for t in range(args.steps):
        step(args.warmup + t)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # the same steps with the boundary's host traffic inside the timed region: source frame up (pinned, async on the step's
    # stream), per-macroblock records + embedding vectors down (the C ABI's blocking fetch); nothing is overlapped
    hio = None
    if rank == 0 and world == 1 and args.host_io_steps > 0:
        hsrc = [[pl.cpu().pin_memory() for pl in fr] for fr in dframes]
        dstage = [[torch.empty_like(pl) for pl in dframes[0]] for _ in encs]
        down = 0
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in range(args.host_io_steps):
            tt = args.warmup + args.steps + t
            with torch.cuda.stream(stream):
                for g in range(len(encs)):
                    for dst, src in zip(dstage[g], hsrc[(tt + g + 1) % nfr]):
                        dst.copy_(src, non_blocking=True)
            for g, enc in enumerate(encs):
                if args.closed_loop:
                    enc.set_ref_device(recon[g][0], recon[g][1], recon[g][2], enc.PREV_INTERNAL, enc.PREV_INTERNAL)
                else:
                    a = dframes[(tt + g) % nfr]
                    enc.set_ref_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), enc.PREV_INTERNAL, enc.PREV_INTERNAL)
                enc.set_fenc_device(dstage[g][0].data_ptr(), dstage[g][1].data_ptr(), dstage[g][2].data_ptr())
            batch.step(args.qp, args.emrate, stream.cuda_stream)
            for enc in encs:
                m_h, e_h = enc.fetch_results(want_embed=True)
                down += m_h.nbytes + sum(np.asarray(v).nbytes for v in e_h.values() if hasattr(v, "nbytes"))
        torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        up = sum(pl.numel() for pl in hsrc[0]) * len(encs) * args.host_io_steps
        hio = {"value": args.gops * n_mb * args.host_io_steps / dth, "unit": "MB/s", "ms_per_step": dth / args.host_io_steps * 1e3,
               "h2d_bytes_per_frame": up / (len(encs) * args.host_io_steps), "d2h_bytes_per_frame": down / (len(encs) * args.host_io_steps),
               "note": "source frames host->device from pinned memory + records and embedding vectors device->host through the C ABI, serialised with the compute (no overlap)"}
    if prof is not None:
        pcamv_amd.load_library().pcamv_gpu_prof_fetch(prof, 0)
        nmb = args.gops * n_mb * args.steps
        names = ["pop+wait", "search", "publish", "reconstruct+RCA", "whole iteration", "-", "16x16 (+skip probe)", "8x8", "sub8x8 + 16x8 + 8x16",
                 "final qpel refine", "reconstruction", "neighbour load", "record store",
                 "pop: ticket (or pass 2: pop+wait)", "pop: queue entry wait (or pass 2: work)", "pop: descriptor load (or pass 2: publish)",
                 "rd: intra SATD analysis", "rd: x264_mb_analyse_p_rd", "rd trial: predict + transform", "rd trial: ssd + psy", "rd trial: cabac header",
                 "rd trial: cabac residual", "rd: final encode + entropy commit", "-",
                 "residual: per-block data", "residual: coded_block_flag chains", "residual: maps + levels", "COUNT residual walks", "COUNT blocks with levels",
                 "COUNT coded blocks", "COUNT non-zero levels", "-"]
        print("wave cycles per macroblock:", {names[i]: round(prof[i] / nmb) for i in range(len(names)) if names[i] != "-"}, file=sys.stderr)
    # dominant kernel: average duration of one launch, HIP events on its own stream
    dom = batch.dominant_kernel()
    avg_ms, n_launch = batch.kernel_time(reset=False)
    n_diag = (W // 16) + 2 * (H // 16 - 1)
    mbs, emb = encs[0].fetch_results(want_embed=True)
    ber = None
    if emb["stc_ok"] == 1 and emb["m"] >= 10:
        final = encs[0].final_mvs(mbs)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import carrier_lsbs
        try:
            ext = pcamv_amd.stc_extract(carrier_lsbs(final), emb["m"])
            ber = float((ext != emb["message"]).mean())
        except pcamv_amd.PcamvError:        # sub-matrix widths outside 2..20: the columns come from the embedder's LCG history,
            ber = None                       # which the stand-alone extractor does not have (payloads below 1/20 bit per MV)
    summary = torch.tensor([emb["n"], emb["m"], emb["num_flip"]], device=cdev, dtype=torch.int64)
    if dist is not None:
        gathered = [torch.zeros_like(summary) for _ in range(world)]
        dist.all_gather(gathered, summary)      # per-GOP result summary to every rank over RCCL (not timed)

    units = args.gops * n_mb * args.steps * world
    value = units / dt
    B_SEARCH = 1920.0       # algorithmic bytes per MB of one analysis/encode pass
    # k_analyse_flow: one launch = the whole analysis pass of every GOP in flight;
    # k_search_diag:  one launch = one anti-diagonal of every GOP in flight
    mbs_per_launch = args.gops * n_mb / (1 if dom == "k_analyse_flow" else n_diag)
    achieved = B_SEARCH * mbs_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM-side traffic of the dominant kernel: PMC FETCH_SIZE + WRITE_SIZE of this same command, collected in
    # separate rocprofv3 passes (tools/dbg/pmc_traffic.sh) and committed under profiles/; scaled to one launch
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if dom == "k_analyse_flow" and os.path.exists(tfile):
        with open(tfile) as fh:
            tj = json.load(fh)["k_analyse_flow_bytes_per_mb"]
        traffic = (tj["fetch_raw"] + tj["write"]) * mbs_per_launch
    # what actually bounds the kernel (committed SQ counter summary of the same command): share of the SIMDs' issue
    # slots in use = waves/SIMD x ACTIVE_INST_ANY / WAVE_CYCLES; the VALU alone = waves/SIMD x ACTIVE_INST_VALU / WAVE_CYCLES
    issue = None
    sfile = os.path.join(ROOT, "profiles", "r01_pmc_sq_summary.json")
    if dom == "k_analyse_flow" and os.path.exists(sfile):
        with open(sfile) as fh:
            sq = json.load(fh)["k_analyse_flow_totals"]
        issue = {"waves_per_simd": 4, "issue_slots_used": 4 * sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                 "valu_busy": 4 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"], "wave_waiting": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                 "source": "profiles/r01_pmc_sq_summary.json"}
    out = {
        "metric": "1080p macroblocks/s (embed on)", "value": value, "unit": "MB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{W}x{H} synthetic I420, --me {args.me} --subme {args.subme} --qp {args.qp} "
                               f"--emrate {args.emrate}, {args.gops} closed GOPs in flight per GPU, " + ("closed loop (pass 2 + loop filter on the GPU)" if args.closed_loop else "open-loop reference"),
                   "mb_per_frame": n_mb, "frames_per_step_per_gpu": args.gops,
                   "note": "BASELINE config 3 asks --subme 7; subme>=6 needs CABAC-size RDO (SURVEY 8f rank 3), not on the GPU path yet"},
        "extracted_payload_BER": ber,
        "carriers_per_frame": int(emb["n"]), "bits_per_frame": int(emb["m"]),
        # 2 x (1920 + 1024) B per macroblock with both passes, 1920 + 1024 for the first pass alone
        "hbm_algorithmic_GBps_whole_path": (5888.0 if args.closed_loop else 2944.0) * value / 1e9,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": traffic,
                     "avg_launch_ms": avg_ms, "mbs_per_launch": mbs_per_launch, "bytes_per_mb": B_SEARCH,
                     "issue_bound": issue},
    }

    if hio is not None:
        out["pcie_inclusive"] = hio
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc
        op = orc.make_params(W, H, me=args.me, subme=args.subme, mv_range=p.i_mv_range, tscale=256, inter=p.inter, cabac=p.b_cabac)
        assert (op.i_psy_rd, op.i_chroma_qp_offset) == (p.i_psy_rd, p.i_chroma_qp_offset)
        o = orc.Oracle(op)
        tcpu = 0.0
        prev = (None, None)
        ref = clip[0]
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers
        for t in range(args.cpu_frames):
            if not args.closed_loop:
                ref = clip[t % nfr]
            o.set_fenc(*clip[(t + 1) % nfr])
            c0 = time.perf_counter()
            o.set_ref(*ref, *prev)                      # plane production is part of the path
            m_o, _ = o.analyse_pframe(args.qp, 1)
            e_o = o.embed_pframe(m_o, args.emrate)
            if args.closed_loop:                        # second pass: final MVs, reconstruction, loop filter
                fo, _, _, ref, _ = o.pass2_pframe(args.qp, m_o, (np.asarray(e_o["flip"]) == 1).astype(np.uint8))
            tcpu += time.perf_counter() - c0
            if args.closed_loop:
                prev = helpers.mv_field(fo["mv"], W // 16, H // 16)
        o.close()
        out["cpu_baseline"] = {"value": args.cpu_frames * n_mb / tcpu, "unit": "MB/s", "cores": 1, "kind": "port",
                               "sample": f"{args.cpu_frames} P frames of the same {W}x{H} workload ({'both passes' if args.closed_loop else 'first pass only'}), oracle/pcamv_oracle.c (scalar C, 1 thread)"}
    if rank == 0:
        print(json.dumps(out))
    batch.close()
    for enc in encs:
        enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
