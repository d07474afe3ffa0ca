#!/usr/bin/env python
"""Build profiles/r01_pmc_hbm_traffic.json from two rocprofv3 PMC passes (one counter each, as
MI355X_MICROARCH.md prescribes) of the same command:

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o pmc -- python bench.py --steps 1 --warmup 1 \
            --no-cpu-baseline --no-sampling --no-roofline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o pmc -- (same)
  python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r01_pmc_hbm_traffic.json

Per kernel: launches, average FETCH_SIZE / WRITE_SIZE (KiB) per launch and HBM bytes per launch
= (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts half of a wide coalesced read)."""
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            a = acc.setdefault(re.sub(r"\(.*", "", r["Kernel_Name"]), [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    f, w = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, (0, 0.0))
        nw, sw = w.get(k, (0, 0.0))
        fa, wa = (sf / nf if nf else 0.0), (sw / nw if nw else 0.0)
        kernels[k] = {"launches": max(nf, nw), "fetch_size_kib_avg": round(fa, 1), "write_size_kib_avg": round(wa, 1),
                      "hbm_bytes_per_launch": int((2 * fa + wa) * 1024)}
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes over `python bench.py --steps 1 --warmup 1 "
            "--no-cpu-baseline --no-sampling --no-roofline`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE "
            "counts half of a wide coalesced read, MI355X_MICROARCH.md section HBM); per-launch averages")
    json.dump({"note": note, "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote %s (%d kernels)" % (out, len(kernels)))


if __name__ == "__main__":
    main()
