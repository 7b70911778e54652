"""Turn the raw rocprofv3 --pmc outputs of tools/dbg/pmc.sh <tag> 256 and tools/dbg/pmc_traffic.sh 256 (under gpurun_out/)
into the committed summaries profiles/r01_pmc_traffic.json, profiles/r01_pmc_sq_summary.json (+ the raw FETCH/WRITE CSVs).
usage: python tools/dbg/pmc_profiles.py <tag> [macroblocks per dispatch]"""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/"
tag = sys.argv[1]
mbs = int(sys.argv[2]) if len(sys.argv) > 2 else 256 * 8160


def agg(pattern):
    f = sorted(glob.glob(R + pattern))[-1]
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
    return tot, n, f


tf, nf, ff = agg('gpurun_out/pmc_fetch/*/*counter_collection.csv')
tw, nw, fw = agg('gpurun_out/pmc_write/*/*counter_collection.csv')
per = {'FETCH_SIZE': {k: v['FETCH_SIZE'] / nf[k]['FETCH_SIZE'] for k, v in tf.items()},
       'WRITE_SIZE': {k: v['WRITE_SIZE'] / nw[k]['WRITE_SIZE'] for k, v in tw.items()}}
out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes, tools/dbg/pmc_traffic.sh) -- python3 bench.py --steps 2 --warmup 1 --gops 256 --cpu-frames 0 --host-io-steps 0",
       "unit": "KB per dispatch (rocprofv3 derived FETCH_SIZE / WRITE_SIZE)", "mbs_per_dispatch": mbs, "per_dispatch_KB": per,
       "note": "gfx950: FETCH_SIZE tallies 128-B fabric requests at 64 B for wide coalesced streams (guide: double it); this kernel's reads are scattered dword rows, for which the guide gives no calibration, so the raw value is kept and the doubled value is the upper bound. Infinity-Cache hits are included (memory-side counters)."}
for kern in ('k_analyse_flow', 'k_pass2_deblock_flow'):
    out[kern + "_bytes_per_mb"] = {"fetch_raw": per['FETCH_SIZE'][kern] * 1024 / mbs, "write": per['WRITE_SIZE'][kern] * 1024 / mbs}
json.dump(out, open(R + 'profiles/r01_pmc_traffic.json', 'w'), indent=1)
shutil.copy(ff, R + 'profiles/r01_pmc_fetch_size.csv'); shutil.copy(fw, R + 'profiles/r01_pmc_write_size.csv')
ta, na, _ = agg(f'gpurun_out/pmc_{tag}_a/*/*counter_collection.csv'); tb, nb, _ = agg(f'gpurun_out/pmc_{tag}_b/*/*counter_collection.csv')
res = {"command": f"tools/dbg/pmc.sh {tag} 256 (two rocprofv3 --pmc passes of: python3 bench.py --steps 1 --warmup 1 --gops 256 --cpu-frames 0 --host-io-steps 0)", "macroblocks_per_dispatch": mbs}
for kern in ('k_analyse_flow', 'k_pass2_deblock_flow'):
    d = dict(ta[kern]); d.update(tb[kern]); nd = na[kern]['SQ_WAVE_CYCLES']
    res[kern + "_dispatches"] = nd; res[kern + "_totals"] = d; res[kern + "_per_macroblock"] = {k: v / (nd * mbs) for k, v in d.items()}
    print(kern, "issue slots", round(4 * d['SQ_ACTIVE_INST_ANY'] / d['SQ_WAVE_CYCLES'], 3), "valu", round(4 * d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES'], 3),
          "waiting", round(d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'], 3), {k: round(v) for k, v in res[kern + "_per_macroblock"].items() if k.startswith("SQ_INSTS")}, out[kern + "_bytes_per_mb"])
json.dump(res, open(R + 'profiles/r01_pmc_sq_summary.json', 'w'), indent=1)
