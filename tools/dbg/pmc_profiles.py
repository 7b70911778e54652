"""Turn the raw rocprofv3 --pmc outputs of `tools/dbg/pmc.sh <tag> <gops>` and `tools/dbg/pmc_traffic.sh <gops>` (under gpurun_out/)
into the committed summaries profiles/r02_pmc_traffic_<tag>.json and profiles/r02_pmc_sq_summary_<tag>.json, which bench.py quotes
(labelled with their source).   usage: python tools/dbg/pmc_profiles.py <tag> <gops> [kernel] [waves per SIMD]"""
import collections, csv, glob, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/"
tag, gops = sys.argv[1], int(sys.argv[2])
kern = sys.argv[3] if len(sys.argv) > 3 else "k_analyse_flow_rd"
wps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
mbs = gops * 8160


def agg(pattern):
    f = sorted(glob.glob(R + pattern))[-1]
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    return tot, n


bench = f"python3 bench.py --steps S --warmup 1 --gops {gops} --cpu-frames 0 --host-io-steps 0 --g-sweep ''"
if glob.glob(R + 'gpurun_out/pmc_fetch/*/*counter_collection.csv'):
    tf, nf = agg('gpurun_out/pmc_fetch/*/*counter_collection.csv')
    tw, nw = agg('gpurun_out/pmc_write/*/*counter_collection.csv')
    per = {k: {"FETCH_SIZE_KB_per_dispatch": tf[k]['FETCH_SIZE'] / nf[k]['FETCH_SIZE'], "WRITE_SIZE_KB_per_dispatch": tw[k]['WRITE_SIZE'] / nw[k]['WRITE_SIZE'],
               "dispatches": nf[k]['FETCH_SIZE']} for k in tf if k.startswith('k_')}
    fetch, write = per[kern]["FETCH_SIZE_KB_per_dispatch"] * 1024 / mbs, per[kern]["WRITE_SIZE_KB_per_dispatch"] * 1024 / mbs
    out = {"command": f"rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, tools/dbg/pmc_traffic.sh {gops}) -- {bench.replace(' S ', ' 2 ')}",
           "gops": gops, "kernel": kern, "mbs_per_dispatch": mbs, "fetch_raw_bytes_per_mb": fetch, "write_bytes_per_mb": write, "bytes_per_mb": fetch + write,
           "algorithmic_bytes_per_mb": 1920,
           "note": "gfx950: FETCH_SIZE tallies 128-B fabric requests at 64 B for wide coalesced streams (guide: double it); this kernel's reads are scattered "
                   "dword rows, for which the guide gives no calibration, so the raw value is kept (the doubled value is the upper bound). Memory-side counters: "
                   "Infinity-Cache hits are included.", "per_kernel": per}
    json.dump(out, open(R + f'profiles/r02_pmc_traffic_{tag}.json', 'w'), indent=1)
    print("traffic B/MB", round(fetch), "+", round(write))
d, nd = {}, 0
for p in 'abc':
    t, n = agg(f'gpurun_out/pmc_{tag}_{p}/*/*counter_collection.csv')
    d.update(t[kern]); nd = max(nd, max(n[kern].values()))
per_mb = {k: v / (nd * mbs) for k, v in d.items()}
summary = {"waves_per_simd": wps,
           "issue_slots_used": wps * d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES'], "valu_busy": wps * d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES'],
           "salu_busy": wps * d['SQ_ACTIVE_INST_SCA'] / d['SQ_WAVE_CYCLES'], "wave_waiting_s_waitcnt": d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'],
           "wave_waiting_issue": d['SQ_WAIT_INST_ANY'] / d['SQ_WAVE_CYCLES'],
           "valu_active_lanes_of_64": d['SQ_THREAD_CYCLES_VALU'] / d['SQ_ACTIVE_INST_VALU'] if 'SQ_THREAD_CYCLES_VALU' in d else None,
           "instructions_per_mb": {k[9:]: round(v) for k, v in per_mb.items() if k.startswith('SQ_INSTS_')}}
res = {"command": f"tools/dbg/pmc.sh {tag} {gops} (three rocprofv3 --pmc passes of: {bench.replace(' S ', ' 1 ')})", "gops": gops, "kernel": kern,
       "macroblocks_per_dispatch": mbs, "dispatches": nd, "totals": d, "per_macroblock": per_mb, "summary": summary,
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); issue_slots_used = waves/SIMD x ACTIVE_INST_ANY / WAVE_CYCLES "
                "= average number of the SIMD's resident waves that are executing an instruction; wave_waiting_s_waitcnt = share of a wave's life parked on s_waitcnt"}
json.dump(res, open(R + f'profiles/r02_pmc_sq_summary_{tag}.json', 'w'), indent=1)
print(json.dumps(summary))
