#!/usr/bin/env python3
"""Turn rocprofv3 CSVs (kernel trace + separate --pmc passes) into the per-launch summary kept under profiles/.
usage: summarize_profile.py <kernel_trace.csv> <out.txt> [--fetch f.csv] [--write w.csv] [--mfma m.csv] [--title ...]
                            [--traffic-json profiles/rNN/traffic.json --workload NAME --bench-json <bench line printed under rocprofv3>]
With --traffic-json the measured FETCH_SIZE + WRITE_SIZE bytes per launch of the grouped-GEMM kernels are merged into that file
under NAME, keyed by bench.workload_key() of the bench line and the hash of the kernel sources, which is what bench.py's
roofline.traffic reads back."""
import argparse
import collections
import csv
import json
import os
import re
import sys


def short(n):
    m = re.search(r'(gg[48]_kernel<[^>]*>|gg_fast_kernel<[^>]*>|gg_generic_kernel<[^>]*>)', n)
    if m:
        return m.group(1)
    m = re.search(r'N12_GLOBAL__N_1\d+([a-z_0-9]+kernel)', n)
    if m:
        return m.group(1)
    m = re.search(r'(\w+_kernel)\b', n)
    return m.group(1) if m else n[:50]


def pmc(path, name):
    a = collections.defaultdict(list)
    if not path:
        return {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == name:
            a[(short(r['Kernel_Name']), r['Grid_Size'])].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in a.items()}


def bench_name(k):
    """bench.py's profile label (ops._timed) of a grouped-GEMM kernel: operand layout for the bf16 kernels, the file for the others."""
    if re.match(r'gg4_kernel<0>|gg8_kernel<0, 0,', k):
        return "grouped_gemm_nt"
    if re.match(r'gg4_kernel<1>|gg8_kernel<0, 1,', k) or k == "gg8c_kernel":      # gg8c: fp32 master weights, K-major (ops.grouped_gemm_f32w)
        return "grouped_gemm_nn"
    return {"gg8w_kernel": "grouped_wgrad_tn", "gg8f_kernel": "grouped_gemm_mxfp8"}.get(k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("out")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--mfma")
    ap.add_argument("--title", default="")
    ap.add_argument("--traffic-json")
    ap.add_argument("--workload")
    ap.add_argument("--bench-json")
    a = ap.parse_args()
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(a.trace)):
        d[(short(r['Kernel_Name']), r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    f, w = pmc(a.fetch, 'FETCH_SIZE'), pmc(a.write, 'WRITE_SIZE')
    mb, ga = pmc(a.mfma, 'SQ_VALU_MFMA_BUSY_CYCLES'), pmc(a.mfma, 'GRBM_GUI_ACTIVE')
    with open(a.out, 'w') as fo:
        fo.write(f"# {a.title}\n")
        fo.write("# mean duration per launch by (kernel, grid threads) from rocprofv3 --kernel-trace; FETCH_SIZE / WRITE_SIZE (KiB per launch) and\n")
        fo.write("# SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE from separate --pmc passes (never combined with tracing domains).\n")
        fo.write("# gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 64 B per 128-B request on wide coalesced streams -> the copy-like kernels are\n")
        fo.write("# priced as 2*FETCH+WRITE (= their algorithmic bytes); for the GEMMs' LDS-DMA reads TCC_MISS*128 B matched FETCH un-doubled in a\n")
        fo.write("# calibration run, so both columns are given.  Infinity-Cache hits are included in these fabric-side counters.\n")
        fo.write("# mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); clk_GHz = GRBM_GUI_ACTIVE/8 / duration.\n")
        fo.write(f"{'kernel':30s} {'grid':>10s} {'calls':>5s} {'avg_ms':>8s} {'FETCH_KiB':>11s} {'WRITE_KiB':>11s} {'GB(F+W)':>8s} {'GB(2F+W)':>9s} {'mfma_util':>9s} {'clk_GHz':>8s}\n")
        for (k, gs), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            if sum(v) < 0.3:
                continue
            fk, wk = f.get((k, gs)), w.get((k, gs))
            g1 = (fk + wk) * 1024 / 1e9 if fk is not None and wk is not None else float('nan')
            g2 = (2 * fk + wk) * 1024 / 1e9 if fk is not None and wk is not None else float('nan')
            m, g = mb.get((k, gs)), ga.get((k, gs))
            util = m / (g / 8 * 1024) if m and g else float('nan')
            avg = sum(v) / len(v)
            clk = g / 8 / (avg * 1e-3) / 1e9 if g else float('nan')
            fo.write(f"{k:30s} {gs:>10s} {len(v):5d} {avg:8.3f} {fk if fk is not None else float('nan'):11.0f} "
                     f"{wk if wk is not None else float('nan'):11.0f} {g1:8.3f} {g2:9.3f} {util:9.3f} {clk:8.2f}\n")
    print(open(a.out).read())
    if a.traffic_json:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import bench
        line = [l for l in open(a.bench_json) if l.startswith("{")][-1]
        res = json.loads(line)
        tot = collections.defaultdict(lambda: [0.0, 0])
        longest = collections.defaultdict(float)          # per label: the longest mean launch
        for (k, gs), v in d.items():
            if bench_name(k):
                longest[bench_name(k)] = max(longest[bench_name(k)], sum(v) / len(v))
        for (k, gs), v in d.items():
            name, fk, wk = bench_name(k), f.get((k, gs)), w.get((k, gs))
            # the expert GEMMs only, not the gate-sized launches of the same kernels: at least a quarter of the label's longest launch
            if name and fk is not None and wk is not None and sum(v) / len(v) > 0.25 * longest[name]:
                tot[name][0] += (fk + wk) * 1024 * len(v)
                tot[name][1] += len(v)
        try:
            cur = json.load(open(a.traffic_json))
        except (OSError, ValueError):
            cur = {}
        cur["_comment"] = ("HBM/fabric-side bytes per launch (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes, KiB -> bytes; Infinity-Cache "
                           "hits included; for the GEMMs' LDS-DMA reads FETCH_SIZE is taken un-doubled -- TCC_MISS * 128 B matched it in round 1's "
                           "calibration), call-weighted mean over the launches bench.py times under one label.  Written by tools/summarize_profile.py "
                           "from the per-launch table named in each entry; bench.py quotes an entry only for the workload and kernel sources it was "
                           "measured on.")
        cur.setdefault("workloads", {})[a.workload] = {
            "key": bench.workload_key(res["config"], res["dtype"]),
            "kernel_sources_sha1_16": bench.kernel_sources_hash(),
            "note": f"FETCH_SIZE + WRITE_SIZE per launch from {os.path.basename(a.out)}",
            "bytes_per_launch": {k: int(round(b / n)) for k, (b, n) in sorted(tot.items())},
        }
        with open(a.traffic_json, "w") as fo:
            json.dump(cur, fo, indent=1)
            fo.write("\n")


if __name__ == "__main__":
    main()
