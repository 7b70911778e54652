"""Turn the per-kernel counter sums that `tools/dbg/pmc_sq.sh <gops> <tag>` and `tools/dbg/pmc_cache_ab.sh <gops> <tag>` leave under
gpurun_out/ (pmc_<tag>_{a,b,c,fetch,write,l2,l1}.json; the raw rocprofv3 CSVs are aggregated on the box, they exceed what gpurun
returns) into the committed summaries profiles/{RND}_pmc_traffic_<name>.json and profiles/{RND}_pmc_sq_summary_<name>.json, which bench.py
quotes (labelled with their source).   usage: python tools/dbg/pmc_profiles.py <tag> <gops> [name] [kernel] [waves per SIMD]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/"
tag, gops = sys.argv[1], int(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "rd"
kern = sys.argv[4] if len(sys.argv) > 4 else "k_analyse_flow_rd"
wps = int(sys.argv[5]) if len(sys.argv) > 5 else 4
RND = os.environ.get("PCAMV_ROUND", "r03")
mbs = gops * 8160


def load(group):
    f = R + f"gpurun_out/pmc_{tag}_{group}.json"
    return json.load(open(f)) if os.path.exists(f) else None


bench = f"python3 bench.py --steps S --warmup 1 --gops {gops} --cpu-frames 0 --host-io-steps 0 --g-sweep ''"
tf, tw = load("fetch"), load("write")
if tf and tw:
    per = {k: {"FETCH_SIZE_KB_per_dispatch": tf[k]['FETCH_SIZE']['sum'] / tf[k]['FETCH_SIZE']['dispatches'],
               "WRITE_SIZE_KB_per_dispatch": tw[k]['WRITE_SIZE']['sum'] / tw[k]['WRITE_SIZE']['dispatches'],
               "dispatches": tf[k]['FETCH_SIZE']['dispatches']} for k in tf if k in tw}
    fetch, write = per[kern]["FETCH_SIZE_KB_per_dispatch"] * 1024 / mbs, per[kern]["WRITE_SIZE_KB_per_dispatch"] * 1024 / mbs
    out = {"command": f"rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, tools/dbg/pmc_cache_ab.sh {gops} {tag}) -- {bench.replace(' S ', ' 2 ')}",
           "gops": gops, "kernel": kern, "mbs_per_dispatch": mbs, "fetch_raw_bytes_per_mb": fetch, "write_bytes_per_mb": write, "bytes_per_mb": fetch + write,
           "algorithmic_bytes_per_mb": 1920,
           "note": "gfx950: FETCH_SIZE tallies 128-B fabric requests at 64 B for wide coalesced streams (guide: double it); this kernel's reads are scattered "
                   "dword rows, for which the guide gives no calibration, so the raw value is kept (the doubled value is the upper bound). Memory-side counters: "
                   "Infinity-Cache hits are included.", "per_kernel": per}
    for grp in ("l2", "l1"):
        t = load(grp)
        if t: out[grp + "_per_mb"] = {c: v['sum'] / v['dispatches'] / mbs for c, v in t[kern].items()}
    json.dump(out, open(R + f'profiles/{RND}_pmc_traffic_{name}.json', 'w'), indent=1)
    print("traffic B/MB", round(fetch), "+", round(write))
d, nd = {}, 0
for p in 'abc':
    t = load(p)
    for c, v in t[kern].items():
        d[c] = v['sum'] / v['dispatches']; nd = max(nd, v['dispatches'])
per_mb = {k: v / mbs for k, v in d.items()}
summary = {"waves_per_simd": wps,
           "issue_slots_used": wps * d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES'], "valu_busy": wps * d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES'],
           "salu_busy": wps * d['SQ_ACTIVE_INST_SCA'] / d['SQ_WAVE_CYCLES'], "wave_waiting_s_waitcnt": d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'],
           "wave_waiting_issue": d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES'],
           "valu_active_lanes_of_64": d['SQ_THREAD_CYCLES_VALU'] / d['SQ_ACTIVE_INST_VALU'] if 'SQ_THREAD_CYCLES_VALU' in d else None,
           "instructions_per_mb": {k[9:]: round(v) for k, v in per_mb.items() if k.startswith('SQ_INSTS_')}}
for grp, key in (("vmemlat", "VmemLatency"), ("ldslat", "LdsLatency"), ("smemlat", "SmemLatency"), ("ifetchlat", "InstrFetchLatency")):
    t = load(grp)
    if t and kern in t: summary.setdefault("average_latency_cycles", {})[key] = t[kern][key]['sum'] / t[kern][key]['dispatches']
res = {"command": f"tools/dbg/pmc_sq.sh {gops} {tag} (three rocprofv3 --pmc passes of: {bench.replace(' S ', ' 1 ')}; per-dispatch averages)", "gops": gops, "kernel": kern,
       "macroblocks_per_dispatch": mbs, "dispatches": nd, "per_dispatch": d, "per_macroblock": per_mb, "summary": summary,
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); issue_slots_used = waves/SIMD x ACTIVE_INST_ANY / WAVE_CYCLES "
                "= average number of the SIMD's resident waves that are executing an instruction; wave_waiting_s_waitcnt = share of a wave's life parked on s_waitcnt"}
json.dump(res, open(R + f'profiles/{RND}_pmc_sq_summary_{name}.json', 'w'), indent=1)
print(json.dumps(summary))
