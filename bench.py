#!/usr/bin/env python3
"""bench.py -- 1080p macroblocks/s of the P-frame hot path (embedding on) on MI355X.

    python bench.py --gpus N --steps K --warmup W

The workload is BASELINE.json's configuration 3 as written: 1920x1088 synthetic I420, --me umh --subme 7 (RD mode decision,
CABAC, psy-RD: the reference's defaults at that level) --qp 26 --emrate 0.5, closed loop.

A step = one P frame through the whole hot path (half-pel plane production, motion search + partition decision with the RD
mode decision, RCA replacement-MV costs, pass-1 reconstruction, cover/cost assembly, syndrome-trellis embedding, then the
reference's second pass: final MVs, reconstruction and loop filter, whose output is the next step's reference) for each of
--gops independent closed GOPs resident on the GPU.  Closed GOPs are the reference's natural sharding unit; with
CABAC a frame is one serial chain of macroblocks (the context states), so a GPU advances many GOP chains together, each launch
carrying the same step of all of them.  Inputs are resident in HBM before the timed region.

N > 1: one process per GPU (the driver launches them with torch.distributed.run; run by hand with --gpus N and no WORLD_SIZE in
the environment, bench.py launches its N ranks itself).  Weak scaling (default): every rank runs --gops GOPs, no data-path
collective.  --strong: --gops is the size of ONE fixed set of GOPs, sharded round-robin over the ranks (pcamv_amd.shard), and
after the timed region payloads are gathered to rank 0 in GOP order with tensor collectives over RCCL.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, `roofline` for the dominant kernel, `cpu_baseline` (the
oracle port on a bounded 1080p sample + the reference itself, oracle/_ref, on CIF with the port beside it), `parity_at_scale`
(after the timed loop: the first GOP of every XCD queue against the port run over the same chained frames, every GOP of a content
class against that class's first GOP), and -- labelled extras, never `value` -- `pcie_inclusive` (the same steps with every
frame's pictures uploaded and its results downloaded, overlapped with the compute), `g_sweep` (throughput against the number of
GOPs in flight), `clip_600` (BASELINE config 3's 600 frames as closed GOPs of --keyint frames: wall time of the whole clip) and
`config4_literal` (8 closed GOPs).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-steganography-pcamv_amd"))

B_SEARCH = 1920.0       # algorithmic bytes per macroblock of one analysis pass (source 384 + reference window 1152 + record/motion 384)
B_WHOLE = 5888.0        # ... of both passes of the closed loop: 2 x (1920 + 1024)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md
MB_RECORD = 236         # sizeof(pcamv_mb_t)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gops", type=int, default=4096, help="closed GOPs in flight per GPU (with --strong: in total); 35 MB of HBM each at 1080p")
    ap.add_argument("--strong", action="store_true", help="one fixed set of --gops GOPs sharded over the ranks; payload gather after the timed region")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1088, help="coded height (1080 rounded up to 16)")
    ap.add_argument("--me", default="umh")
    ap.add_argument("--subme", type=int, default=7)
    ap.add_argument("--no-cabac", action="store_true")
    ap.add_argument("--qp", type=int, default=26)
    ap.add_argument("--emrate", type=float, default=0.5)
    ap.add_argument("--open-loop", action="store_true", help="pass 1 only: no pass 2 / loop filter, the reference is the previous source frame")
    ap.add_argument("--host-io-steps", type=int, default=6, help="steps of the PCIe-inclusive pipeline (0 = skip; rank 0, N=1 only)")
    ap.add_argument("--classes", type=int, default=64, help="distinct content classes: GOP g starts at frame g mod classes of a synthetic clip that long")
    ap.add_argument("--parity-gops", type=int, default=8, help="GOPs compared with the CPU port after the timed loop (the first of every XCD queue; 0 = skip)")
    ap.add_argument("--clip-keyints", default="30,8", help="keyint values of the 600-frame clip block ('' = skip; rank 0, N=1 only)")
    ap.add_argument("--clip-frames", type=int, default=600)
    ap.add_argument("--g-sweep", default="1,8,20,64,256,1024,2048", help="GOP counts of the low-G sweep ('' = skip; rank 0, N=1 only)")
    ap.add_argument("--cpu-frames", type=int, default=24, help="1080p P frames timed for the CPU baseline (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-cif-frames", type=int, default=200, help="CIF P frames timed through the reference itself (oracle/_ref) and through the port (0 = skip)")
    return ap.parse_args()


def self_launch(args):
    """--gpus N by hand: start the N ranks the way the driver does and pass rank 0's line through (nothing here has touched the GPU)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def tri(i, n):
    """frame i of a clip of n frames played forwards, then backwards, then forwards ... (no cut anywhere: a cyclic clip jumps back by
    its whole pan at the wrap, a scene change no encoder would code as a P frame)"""
    period = 2 * n - 2
    i %= period
    return i if i < n else period - i


class Gops:
    """closed-GOP pipelines on one device: contexts + batch + the step function.  phases[k] = frame of the (cyclic) clip GOP k starts at:
    the reference of its step 0 is that frame (standing in for the GOP's I picture), the source of its step t is frame phases[k] + t + 1"""

    def __init__(self, pcamv_amd, params, dframes, phases, device, closed_loop):
        self.dframes, self.closed, self.phases = dframes, closed_loop, [int(ph) % (2 * len(dframes) - 2) for ph in phases]
        self.encs = [pcamv_amd.Encoder(params, device=device) for _ in self.phases]
        self.batch = pcamv_amd.Batch(self.encs)
        if closed_loop:
            self.batch.set_closed_loop(True)
        self.recon = [e.recon_device() for e in self.encs]
        self.started = False

    def step(self, t, qp, emrate, stream, fenc_of=None):
        nfr = len(self.dframes)
        for k, enc in enumerate(self.encs):
            ph = self.phases[k]
            if self.closed and self.started:     # reference = this GOP's own deblocked reconstruction of the previous step, chained MV field
                enc.set_ref_device(self.recon[k][0], self.recon[k][1], self.recon[k][2], enc.PREV_INTERNAL, enc.PREV_INTERNAL)
            elif self.closed:                    # a GOP's first P frame: the reference is an I picture, no motion field behind it
                a = self.dframes[tri(t + ph, nfr)]
                enc.set_ref_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), 0, 0)
            else:
                a = self.dframes[tri(t + ph, nfr)]
                enc.set_ref_device(a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), enc.PREV_INTERNAL, enc.PREV_INTERNAL)
            b = fenc_of(k) if fenc_of else [pl.data_ptr() for pl in self.dframes[tri(t + ph + 1, nfr)]]
            enc.set_fenc_device(b[0], b[1], b[2])
        self.batch.step(qp, emrate, stream)
        self.started = True

    def close(self):
        self.batch.close()
        for e in self.encs:
            e.close()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    closed_loop = not args.open_loop

    import numpy as np
    import torch
    import pcamv_amd
    from pcamv_amd.synth import make_clip
    from pcamv_amd.shard import gop_assignment, gather_payloads, pack_gop_payload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    # rehearsal on a one-GPU box only: PCAMV_BENCH_REHEARSE=1 puts every rank on GPU 0 and runs the (untimed)
    # collectives over gloo, since RCCL refuses two ranks on one device; the driver's runs use RCCL, one GPU per rank
    rehearse = os.environ.get("PCAMV_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if rehearse else dev          # where collective operands live

    W, H = args.width, args.height
    n_mb = (W // 16) * (H // 16)
    p = pcamv_amd.param_default(W, H)
    pcamv_amd.param_parse(p, "me", args.me)
    pcamv_amd.param_parse(p, "subme", args.subme)
    if args.no_cabac:
        pcamv_amd.param_parse(p, "no-cabac", 1)

    # the GOPs of this rank: weak = --gops each; strong = this rank's share of one fixed set.  GOP g starts at phase g of the clip.
    if args.strong:
        mine = gop_assignment(args.gops, world, rank)
        total_gops = args.gops
    else:
        mine = list(range(rank * args.gops, (rank + 1) * args.gops))
        total_gops = args.gops * world
    G = len(mine)
    if G < 1:
        sys.exit(f"rank {rank}: no GOP to run (--strong --gops {args.gops} over {world} ranks)")

    # synthetic clip (SURVEY 8(d) generator) played forwards and backwards (tri); GOP g starts at position g of that cycle, so
    # --classes content classes with different source pictures and motion histories are in flight (round 2 cycled 6 frames: 683
    # GOPs per class, and every sixth frame of a GOP jumped back by the clip's whole pan)
    nfr = max(2, args.classes // 2 + 1)
    clip = make_clip(W, H, nfr, seed=13)
    dframes = [[torch.from_numpy(pl).to(dev) for pl in fr] for fr in clip]
    run = Gops(pcamv_amd, p, dframes, mine, local, closed_loop)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for t in range(args.warmup):
        run.step(t, args.qp, args.emrate, stream.cuda_stream)
    barrier()
    run.batch.kernel_time(reset=True)
    prof = None
    if os.environ.get("PCAMV_PROF_DUMP") == "1":      # diagnostics build of the library (-DPCAMV_PROF): wave cycles per phase
        import ctypes
        prof = (ctypes.c_ulonglong * 48)()
        pcamv_amd.load_library().pcamv_gpu_prof_fetch(prof, 1)
    t0 = time.perf_counter()
    for t in range(args.steps):
        run.step(args.warmup + t, args.qp, args.emrate, stream.cuda_stream)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    t_next = args.warmup + args.steps

    if prof is not None:
        pcamv_amd.load_library().pcamv_gpu_prof_fetch(prof, 0)
        nmb = G * n_mb * args.steps
        names = ["pop+wait", "search", "publish", "reconstruct+RCA", "whole iteration", "-", "16x16 (+skip probe)", "8x8", "sub8x8 + 16x8 + 8x16",
                 "final qpel refine", "reconstruction", "neighbour load", "record store",
                 "pop: ticket (or pass 2: pop+wait)", "pop: queue entry wait (or pass 2: work)", "pop: descriptor load (or pass 2: publish)",
                 "rd: intra SATD analysis", "rd: x264_mb_analyse_p_rd", "rd trial: predict + transform", "rd trial: ssd + psy", "rd trial: cabac header",
                 "rd trial: cabac residual", "rd: final encode + entropy commit", "-",
                 "residual: per-block data", "residual: coded_block_flag chains", "residual: maps + levels", "COUNT residual walks", "COUNT blocks with levels",
                 "COUNT coded blocks", "COUNT non-zero levels", "-",
                 "COUNT list evaluations", "COUNT candidates", "rca: copy + window load", "rca: first nine-point list", "rca: 4 predictions + transforms", "rca: 36-candidate list", "list evaluation (cycles)",
                 "-", "COUNT rca groups of 4", "COUNT rca single re-encodes", "COUNT carriers"]
        print("wave cycles per macroblock:", {names[i]: round(prof[i] / nmb) for i in range(len(names)) if names[i] != "-"}, file=sys.stderr)

    # dominant kernel: average duration of one launch, HIP events on its own stream, over the timed steps
    dom = run.batch.dominant_kernel()
    avg_ms, n_launch = run.batch.kernel_time(reset=False)
    n_diag = (W // 16) + 2 * (H // 16 - 1)
    flow = dom.startswith("k_analyse_flow")

    # the embedded payload comes back out of the final motion vectors (the first GOPs of this rank: one per XCD queue)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from helpers import carrier_lsbs
    T_done = args.warmup + args.steps
    n_chk = max(1, min(args.parity_gops if args.parity_gops > 0 else 1, G, 8))
    ber, ber_bits, chk = None, 0, []
    for k in range(n_chk):
        mbs_k, emb_k = run.encs[k].fetch_results(want_embed=True)
        rec_k = run.encs[k].fetch_recon() if closed_loop else None
        chk.append((mbs_k, emb_k, rec_k))
        if emb_k["m"] > 0:
            if emb_k["stc_ok"] != 1:
                sys.exit(f"bench.py: the syndrome-trellis embedding of GOP {k} failed")
            ext = pcamv_amd.stc_extract(carrier_lsbs(run.encs[k].final_mvs(mbs_k)), emb_k["m"])
            ber = (0.0 if ber is None else ber) + float((ext != emb_k["message"]).sum())
            ber_bits += int(emb_k["m"])
    if ber is not None:
        ber /= ber_bits
    mbs, emb = chk[0][0], chk[0][1]

    # ---- parity at scale: what the timed loop computed (4 waves per SIMD, eight queues, work stealing, second-pass tasks of 8
    # macroblocks) against the CPU port on the same chained frames.  The port runs in threads (ctypes releases the GIL) while the
    # extras below keep the GPU busy; joined before the CPU baseline is timed.
    parity, par_threads, par_res = None, [], {}
    if rank == 0 and world == 1 and args.parity_gops > 0 and closed_loop and T_done > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc
        op_par = orc.make_params(W, H, me=args.me, subme=args.subme, mv_range=p.i_mv_range, tscale=256, inter=p.inter, cabac=p.b_cabac)
        assert (op_par.i_psy_rd, op_par.i_chroma_qp_offset) == (p.i_psy_rd, p.i_chroma_qp_offset)

        phases_par = list(run.phases)

        def port_gop(k):
            t_c = time.perf_counter()
            o = orc.Oracle(op_par)
            ph, prev, ref = phases_par[k], (None, None), clip[tri(phases_par[k], nfr)]
            for t in range(T_done):
                o.set_fenc(*clip[tri(t + ph + 1, nfr)])
                o.set_ref(*ref, *prev)
                m_o, _ = o.analyse_pframe(args.qp, 1)
                e_o = o.embed_pframe(m_o, args.emrate)
                fo, _, _, ref, _ = o.pass2_pframe(args.qp, m_o, (np.asarray(e_o["flip"]) == 1).astype(np.uint8))
                prev = helpers.mv_field(fo["mv"], W // 16, H // 16)
            o.close()
            par_res[k] = (m_o, e_o, ref, time.perf_counter() - t_c)

        par_threads = [threading.Thread(target=port_gop, args=(k,)) for k in range(n_chk)]
        for th in par_threads:
            th.start()
        # every GOP of a content class has seen the same pictures and the same message stream: its records and flip map must be
        # those of the class's first GOP (compared on the device, all GOPs)
        mb_bytes_, flip_bytes_ = n_mb * MB_RECORD, 16 * n_mb
        d_all = torch.empty((G, mb_bytes_ + flip_bytes_), dtype=torch.uint8, device=dev)
        run.batch.copy_results_async(d_all.data_ptr(), mb_bytes_ + flip_bytes_, d_all.data_ptr() + mb_bytes_, mb_bytes_ + flip_bytes_, stream.cuda_stream)
        torch.cuda.synchronize()
        first_of = {}
        idx_first = torch.tensor([first_of.setdefault(ph, k) for k, ph in enumerate(run.phases)], device=dev)
        n_car = torch.tensor([0], device=dev)
        same = True
        for lo in range(0, G, 256):
            hi = min(G, lo + 256)
            same = same and bool((d_all[lo:hi, :mb_bytes_] == d_all[idx_first[lo:hi], :mb_bytes_]).all().item())
        # (flip maps: the first `n carriers` bytes are meaningful, the rest is zeroed by the embedding stage)
        for lo in range(0, G, 256):
            hi = min(G, lo + 256)
            same = same and bool((d_all[lo:hi, mb_bytes_:] == d_all[idx_first[lo:hi], mb_bytes_:]).all().item())
        del d_all, n_car
        torch.cuda.empty_cache()
        if not same:
            sys.exit("bench.py: GOPs of the same content class ended the timed loop with different records / flip maps")
        parity = {"gops_identical_within_class": True, "content_classes": len(first_of), "gops": G}

    # N > 1: per-rank summary to every rank, and (--strong) the payloads of the set's first GOPs to rank 0 in GOP order;
    # tensor collectives over RCCL, after the timed region
    summary = torch.tensor([emb["n"], emb["m"], emb["num_flip"]], device=cdev, dtype=torch.int64)
    gathered = None
    if dist is not None:
        allsum = [torch.zeros_like(summary) for _ in range(world)]
        dist.all_gather(allsum, summary)
    if args.strong:
        n_gather = min(args.gops, 64)
        local_payload = {}
        for k, g in enumerate(gop_assignment(n_gather, world, rank)):       # rank r's k-th GOP is r + k * world
            m_k, e_k = run.encs[k].fetch_results(want_embed=True)
            local_payload[g] = pack_gop_payload(m_k, np.asarray(e_k["flip"])[:e_k["n"]])
        out_p = gather_payloads(dist, local_payload, n_gather, world, rank, device=cdev)
        if rank == 0:
            h = hashlib.sha1()
            for b in out_p:
                h.update(b)
            gathered = {"gops": n_gather, "bytes": sum(len(b) for b in out_p), "sha1": h.hexdigest(),
                        "note": "records + flip map of the last step of the set's first GOPs, in GOP order (all_gather of sizes + gather of padded "
                                "bytes over RCCL); with the same --gops/--steps the hash does not depend on the number of ranks"}

    units = total_gops * n_mb * args.steps
    value = units / dt
    # one launch of the flow kernels = the analysis pass of every GOP of the rank; of k_search_diag = one anti-diagonal of them
    mbs_per_launch = G * n_mb / (1 if flow else n_diag)
    achieved = B_SEARCH * mbs_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM-side traffic and SIMD issue counters of the dominant kernel: from the committed PMC summaries of this same command
    # (tools/dbg/pmc.sh + pmc_profiles.py; separate rocprofv3 passes), not measured in this run -- labelled with their source
    traffic, traffic_src, issue, valu_issue = None, None, None, None
    tag = "rd" if dom == "k_analyse_flow_rd" else "base"

    def profile_json(kind):
        for rnd in ("r03", "r02"):
            f = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{kind}_{tag}.json")
            if os.path.exists(f):
                with open(f) as fh:
                    return json.load(fh), os.path.relpath(f, ROOT)
        return None, None

    tj, tname = profile_json("traffic")
    if tj:
        traffic = tj["bytes_per_mb"] * mbs_per_launch
        traffic_src = f"{tname} ({tj['gops']} GOPs in flight; {tj['bytes_per_mb']:.0f} B per macroblock scaled to this launch)"
    sj, sname = profile_json("sq_summary")
    if sj:
        issue = dict(sj["summary"], source=f"{sname} (committed profile of this command, not this run)")
        # what bounds the kernel: VALU instructions per macroblock (committed counters) x the macroblocks of this launch, against what
        # the chip's 1024 SIMDs issue in the launch's measured duration (one wave64 VALU instruction per 4 cycles at 2.4 GHz)
        if avg_ms > 0 and flow:
            valu_issue = {"frac": sj["summary"]["instructions_per_mb"]["VALU"] * mbs_per_launch / (1024 * 2.4e9 / 4 * avg_ms * 1e-3),
                          "valu_per_mb": sj["summary"]["instructions_per_mb"]["VALU"], "lanes_active_of_64": sj["summary"].get("valu_active_lanes_of_64"),
                          "peak": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction", "source": sname}
    out = {
        "metric": "1080p macroblocks/s (embed on)", "value": value, "unit": "MB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE config 3: {W}x{H} synthetic I420, --me {args.me} --subme {args.subme} --qp {args.qp} --emrate {args.emrate}"
                               f"{' --no-cabac' if args.no_cabac else ''}, " + ("closed loop (pass 2 + loop filter on the GPU)" if closed_loop else "open-loop reference")
                               + (f", {args.gops} closed GOPs in total sharded over the ranks" if args.strong else f", {args.gops} closed GOPs in flight per GPU"),
                   "mb_per_frame": n_mb, "gops_per_gpu": G, "frames_per_step": total_gops, "content_classes": min(2 * nfr - 2, total_gops),
                   "value_is": "inputs resident in HBM when the timed region starts; pcie_inclusive is the host-fed pipeline"},
        "extracted_payload_BER": ber, "BER_checked": {"gops": n_chk, "bits": ber_bits},
        "carriers_per_frame": int(emb["n"]), "bits_per_frame": int(emb["m"]),
        "hbm_algorithmic_GBps_whole_path": (B_WHOLE if closed_loop else B_WHOLE / 2) * value / 1e9,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_ms": avg_ms, "launches_timed": n_launch, "mbs_per_launch": mbs_per_launch, "bytes_per_mb": B_SEARCH,
                     "bound_note": "byte/integer work priced against HBM as the contract asks; what limits this kernel is not bandwidth but instruction "
                                   "issue (at 4096 chains the VALU is ~83 % busy; with few chains, the dependent chain inside a macroblock): issue_counters, DESIGN.md 4a",
                     "issue_counters": issue, "valu_issue_frac": valu_issue},
    }
    if gathered is not None:
        out["gathered_payloads"] = gathered

    solo = rank == 0 and world == 1
    # ---- PCIe-inclusive pipeline (extra): every step's source pictures go host -> device and its records + flip maps device -> host,
    # double-buffered and overlapped with the compute: uploads of step k+1 and downloads of step k run on a copy stream during the compute
    if solo and args.host_io_steps > 0:
        mb_bytes, flip_bytes = n_mb * MB_RECORD, 16 * n_mb
        row_out = mb_bytes + flip_bytes
        fbytes = [pl.numel() for pl in dframes[0]]
        row_in = sum(fbytes)
        offs = [0, fbytes[0], fbytes[0] + fbytes[1]]
        hsrc = [torch.cat([pl.reshape(-1) for pl in fr]).cpu().pin_memory() for fr in dframes]          # pinned source pictures (Y|U|V per frame)
        dstage = [torch.empty((G, row_in), dtype=torch.uint8, device=dev) for _ in range(2)]            # device staging of the uploads, two steps deep
        d_out = torch.empty((G, row_out), dtype=torch.uint8, device=dev)                                 # device staging of the results
        h_out = torch.empty((G, row_out), dtype=torch.uint8).pin_memory()
        copy_st = torch.cuda.Stream(device=dev)
        up_done = [torch.cuda.Event() for _ in range(2)]
        step_done = [torch.cuda.Event() for _ in range(2)]
        down_done = torch.cuda.Event()
        down_t = [torch.cuda.Event(enable_timing=True) for _ in range(args.host_io_steps)]     # step k's results are in host memory

        def upload(t, buf):
            with torch.cuda.stream(copy_st):
                for k in range(G):
                    dstage[buf][k].copy_(hsrc[tri(t + run.phases[k] + 1, nfr)], non_blocking=True)
                up_done[buf].record(copy_st)

        def fenc_of(buf):
            base = dstage[buf].data_ptr()
            return lambda k: [base + k * row_in + o for o in offs]

        torch.cuda.synchronize()
        t1 = time.perf_counter()
        upload(t_next, 0)
        for k in range(args.host_io_steps):
            cur = k & 1
            stream.wait_event(up_done[cur])
            run.step(t_next + k, args.qp, args.emrate, stream.cuda_stream, fenc_of(cur))
            if k > 0:
                stream.wait_event(down_done)           # the result staging is free again
            run.batch.copy_results_async(d_out.data_ptr(), row_out, d_out.data_ptr() + mb_bytes, row_out, stream.cuda_stream)
            step_done[cur].record(stream)
            if k + 1 < args.host_io_steps:             # the next step's pictures up (its staging was last read by step k-1)
                if k > 0:
                    copy_st.wait_event(step_done[cur ^ 1])
                upload(t_next + k + 1, cur ^ 1)
            with torch.cuda.stream(copy_st):           # this step's results down as soon as it is done (overlaps the next step)
                copy_st.wait_event(step_done[cur])
                h_out.copy_(d_out, non_blocking=True)
                down_done.record(copy_st)
                down_t[k].record(copy_st)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t1
        nio = args.host_io_steps
        steady_ms = down_t[0].elapsed_time(down_t[nio - 1]) / (nio - 1) if nio > 1 else dth * 1e3
        t_next += args.host_io_steps
        got = np.frombuffer(h_out[0, :mb_bytes].numpy().tobytes(), dtype=np.uint8)
        blocking = np.asarray(run.encs[0].fetch_results(want_embed=False)[0]).view(np.uint8).reshape(-1)
        if not np.array_equal(got, blocking):
            sys.exit("bench.py: the records downloaded by the overlapped pipeline differ from the blocking fetch")
        out["pcie_inclusive"] = {"value": G * n_mb / (steady_ms * 1e-3), "unit": "MB/s", "ms_per_step": steady_ms,
                                 "steps": nio, "whole_pipeline_ms": dth * 1e3, "fill_and_drain_ms": dth * 1e3 - steady_ms * (nio - 1),
                                 "value_incl_fill_and_drain": G * n_mb * nio / dth,
                                 "h2d_bytes_per_frame": row_in, "d2h_bytes_per_frame": row_out,
                                 "note": "extra, never `value`: source pictures host->device from pinned memory and records + flip maps device->host "
                                         "(pcamv_gpu_batch_copy_results_async) on a copy stream, double-buffered, overlapped with the compute; "
                                         "value / ms_per_step = the steady state (HIP events: arrival of one step's results in host memory to the "
                                         "next one's, averaged over the pipeline's steps); fill_and_drain_ms = the first step's upload + compute + "
                                         "download that nothing overlaps"}
        del dstage, d_out, h_out, hsrc

    # the main set of GOPs is done with (its results were fetched above): its HBM goes to the runs below
    if solo:
        run.close()
        run = None
        torch.cuda.empty_cache()

    # ---- throughput against the number of GOPs in flight (extra): a frame's macroblocks form one chain, so few GOPs = few busy waves
    if solo and args.g_sweep:
        sweep = []
        for g_n in [int(x) for x in args.g_sweep.split(",") if x]:
            if g_n >= G:
                continue
            sub = Gops(pcamv_amd, p, dframes, range(g_n), local, closed_loop)
            sub.step(0, args.qp, args.emrate, stream.cuda_stream)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for t in range(2):
                sub.step(1 + t, args.qp, args.emrate, stream.cuda_stream)
            torch.cuda.synchronize()
            d = (time.perf_counter() - ts) / 2
            sweep.append({"gops": g_n, "value": g_n * n_mb / d, "unit": "MB/s", "ms_per_step": d * 1e3})
            sub.close()
        sweep.append({"gops": G, "value": value, "unit": "MB/s", "ms_per_step": dt / args.steps * 1e3})
        out["g_sweep"] = sweep
        # round 1's workload (--subme 5: no RD decision, a frame is a wavefront of macroblocks instead of one chain) for comparison
        if args.subme >= 6 and not args.no_cabac:
            p5 = pcamv_amd.param_default(W, H)
            pcamv_amd.param_parse(p5, "me", args.me)
            pcamv_amd.param_parse(p5, "subme", 5)
            sub = Gops(pcamv_amd, p5, dframes, range(256), local, closed_loop)
            for t in range(2):
                sub.step(t, args.qp, args.emrate, stream.cuda_stream)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for t in range(5):
                sub.step(2 + t, args.qp, args.emrate, stream.cuda_stream)
            torch.cuda.synchronize()
            d = (time.perf_counter() - ts) / 5
            out["other_workloads"] = [{"workload": f"{W}x{H} --me {args.me} --subme 5, 256 GOPs in flight, same closed loop", "value": 256 * n_mb / d, "unit": "MB/s", "ms_per_step": d * 1e3}]
            sub.close()

    # ---- BASELINE config 3 as a clip (extra): --clip-frames frames cut into closed GOPs of keyint frames (1 I + keyint - 1 P), all GOPs in
    # flight on this GPU; wall time until the clip's last P frame is done (every step's ramp included).  The I pictures are not part of
    # the path (the source picture stands in for their reconstruction) and of no time here.
    if solo and args.clip_keyints and closed_loop:
        blocks = []
        for K in [int(x) for x in args.clip_keyints.split(",") if x]:
            n_g = max(1, args.clip_frames // K)
            if K < 2:
                continue
            sub = Gops(pcamv_amd, p, dframes, [g * K for g in range(n_g)], local, closed_loop)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for t in range(K - 1):
                sub.step(t, args.qp, args.emrate, stream.cuda_stream)
            torch.cuda.synchronize()
            d = time.perf_counter() - ts
            blocks.append({"keyint": K, "gops": n_g, "p_frames": n_g * (K - 1), "wall_s": d, "value": n_g * (K - 1) * n_mb / d, "unit": "MB/s"})
            sub.close()
        out["clip_600"] = {"frames": args.clip_frames, "runs": blocks,
                           "note": "extra: P-frame macroblocks / wall time of the whole clip on one GPU, no warm-up (context creation outside); "
                                   "keyint 30 = 20 chains for 29 steps: a CABAC frame is one serial chain, so few GOPs = few busy waves"}
    # ---- BASELINE config 4 read literally (extra): 8 closed GOPs.  On one GPU that is 8 chains in flight; on 8 GPUs one chain each --
    # the g_sweep's 1-GOP time predicts what 8 GPUs would take for the same 8 GOPs (nothing multi-GPU is measured here)
    if solo and closed_loop and "g_sweep" in out:
        by_g = {e["gops"]: e for e in out["g_sweep"]}
        if 1 in by_g and 8 in by_g:
            out["config4_literal"] = {"gops": 8, "one_gpu_ms_per_step": by_g[8]["ms_per_step"], "one_gpu_value": by_g[8]["value"], "unit": "MB/s",
                                      "eight_gpus_predicted_ms_per_step": by_g[1]["ms_per_step"],
                                      "predicted_scaling_1_to_8": by_g[8]["ms_per_step"] / by_g[1]["ms_per_step"],
                                      "note": "prediction from this run's g_sweep, not a measurement: with 8 GOPs there are 8 chains whatever the number "
                                              "of GPUs; the >= 6x scaling target needs >= 8 x the GOPs one GPU saturates with (weak scaling, the default of --gpus N)"}

    # ---- parity at scale, second half: the port's threads have been running beside the extras above
    if par_threads:
        for th in par_threads:
            th.join()
        for k in range(n_chk):
            if k not in par_res:
                sys.exit(f"bench.py: the CPU port of GOP {k} did not finish")
            m_o, e_o, rec_o, _ = par_res[k]
            mbs_k, emb_k, rec_k = chk[k]
            for f in mbs_k.dtype.names:
                if not np.array_equal(mbs_k[f], m_o[f]):
                    sys.exit(f"bench.py: parity at scale: GOP {k} field {f} differs from the CPU port after {T_done} chained steps")
            if not np.array_equal(np.asarray(emb_k["flip"]), np.asarray(e_o["flip"])) or not np.array_equal(np.asarray(emb_k["message"]), np.asarray(e_o["message"])):
                sys.exit(f"bench.py: parity at scale: GOP {k}: flip map / message differ from the CPU port")
            for a, b in zip(rec_k, rec_o):
                if not np.array_equal(a, b):
                    sys.exit(f"bench.py: parity at scale: GOP {k}: deblocked picture differs from the CPU port")
        parity.update(gops_vs_port=list(range(n_chk)), chained_steps=T_done, ok=True,
                      compared="pass-1 records, flip map, message and deblocked planes of the timed loop's last step (each the product of all "
                               "chained steps before it) against oracle/pcamv_oracle.c run over the same frames; GOP k lives in XCD queue k & 7",
                      port_cpu_s=round(sum(par_res[k][3] for k in range(n_chk)), 1), port_threads=n_chk)
    if parity is not None:
        out["parity_at_scale"] = parity

    # ---- CPU baseline (rank 0, N=1): the port on a bounded sample of the same 1080p workload; the reference itself on CIF
    if solo and args.cpu_frames > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc
        import helpers

        def port_run(w, h, frames, clip_, mv_range):
            o = orc.Oracle(orc.make_params(w, h, me=args.me, subme=args.subme, mv_range=mv_range, tscale=256, inter=p.inter, cabac=p.b_cabac))
            tcpu, prev, ref = 0.0, (None, None), clip_[0]
            for t in range(frames):
                if not closed_loop:
                    ref = clip_[tri(t, len(clip_))]
                o.set_fenc(*clip_[tri(t + 1, len(clip_))])
                c0 = time.perf_counter()
                o.set_ref(*ref, *prev)                      # plane production is part of the path
                m_o, _ = o.analyse_pframe(args.qp, 1)
                e_o = o.embed_pframe(m_o, args.emrate)
                if closed_loop:                             # second pass: final MVs, reconstruction, loop filter
                    fo, _, _, ref, _ = o.pass2_pframe(args.qp, m_o, (np.asarray(e_o["flip"]) == 1).astype(np.uint8))
                tcpu += time.perf_counter() - c0
                if closed_loop:
                    prev = helpers.mv_field(fo["mv"], w // 16, h // 16)
            o.close()
            return tcpu

        op0 = orc.make_params(W, H, me=args.me, subme=args.subme, mv_range=p.i_mv_range, tscale=256, inter=p.inter, cabac=p.b_cabac)
        assert (op0.i_psy_rd, op0.i_chroma_qp_offset) == (p.i_psy_rd, p.i_chroma_qp_offset)
        tcpu = port_run(W, H, args.cpu_frames, clip, p.i_mv_range)
        cb = {"value": args.cpu_frames * n_mb / tcpu, "unit": "MB/s", "cores": 1, "kind": "port",
              "sample": f"{args.cpu_frames} P frames of the same {W}x{H} workload ({'both passes' if closed_loop else 'first pass only'}, embedding on), "
                        f"oracle/pcamv_oracle.c (scalar C, 1 thread), {tcpu:.1f} s"}
        # The reference's own code (oracle/_ref, built from its sources) keeps its per-frame record in arrays sized for CIF, so it is
        # timed on CIF: analysis + second pass with the same options (its STC call is outside the harness: the flip map comes from the
        # port, untimed); the port on the same frames gives the ratio between the two.
        import refh
        if refh.available() and args.cpu_cif_frames > 0:
            cw, ch, cn = 352, 288, 396
            cclip = make_clip(cw, ch, nfr, seed=13)
            mvr = pcamv_amd.level_mv_range(cw, ch)
            r = refh.Ref(cw, ch, qp=args.qp, me=args.me, subme=args.subme, mv_range=mvr, cabac=int(p.b_cabac), embed=1, inter_flags=int(p.inter))
            oe = orc.Oracle(orc.make_params(cw, ch, me=args.me, subme=args.subme, mv_range=mvr, tscale=256, inter=p.inter, cabac=p.b_cabac))
            tref, prev, ref = 0.0, (None, None), cclip[0]
            for t in range(args.cpu_cif_frames):
                if not closed_loop:
                    ref = cclip[tri(t, nfr)]
                r.set_fenc(*cclip[tri(t + 1, nfr)])
                c0 = time.perf_counter()
                r.set_ref(*ref, *prev)
                m_r, _ = r.analyse_pframe(args.qp)
                tref += time.perf_counter() - c0
                if closed_loop:
                    e_r = oe.embed_pframe(m_r.view(orc.MB_DTYPE), args.emrate)
                    flips = (np.asarray(e_r["flip"]) == 1).astype(np.int8)
                    c0 = time.perf_counter()
                    fm, _, _, ref, _ = r.pass2_pframe(flips, args.qp)
                    tref += time.perf_counter() - c0
                    prev = helpers.mv_field(fm["mv"], cw // 16, ch // 16)
            oe.close()
            tport = port_run(cw, ch, args.cpu_cif_frames, cclip, mvr)
            cb["ref_cif"] = {"value": args.cpu_cif_frames * cn / tref, "unit": "MB/s", "cores": 1, "kind": "reference",
                             "port_on_same_sample": args.cpu_cif_frames * cn / tport,
                             "path": "C path (cpu = 0: no x86 SIMD -- no assembler in this image), 1 thread, its stc_embed call excluded",
                             "sample": f"{args.cpu_cif_frames} CIF P frames, same options, through oracle/_ref (the reference's sources, gcc -O3): analysis"
                                       f"{' + second pass' if closed_loop else ''} ({tref:.1f} s); the port's figure beside it also contains its embedding ({tport:.1f} s)"}
            if "clip_600" in out:
                for b_ in out["clip_600"]["runs"]:
                    b_["x_ref_cif"] = b_["value"] / cb["ref_cif"]["value"]
        out["cpu_baseline"] = cb
    if rank == 0:
        print(json.dumps(out))
    if run is not None:
        run.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
