#!/usr/bin/env python3
"""Turn rocprofv3 CSVs (kernel trace + separate --pmc passes) into the per-launch summary kept under profiles/.
usage: summarize_profile.py <kernel_trace.csv> <out.txt> [--fetch f.csv] [--write w.csv] [--mfma m.csv] [--title ...]"""
import argparse
import collections
import csv
import re


def short(n):
    m = re.search(r'(gg8_kernel<[^>]*>|gg_fast_kernel<[^>]*>|gg_generic_kernel<[^>]*>)', n)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("out")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--mfma")
    ap.add_argument("--title", default="")
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


if __name__ == "__main__":
    main()
