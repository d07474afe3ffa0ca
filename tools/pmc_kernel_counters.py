#!/usr/bin/env python
"""Per-kernel sums of the counters of one or more rocprofv3 --pmc passes (csv output), averaged per launch.
   python tools/pmc_kernel_counters.py <dir> [<dir> ...] [--match substring]"""
import collections
import csv
import glob
import os
import re
import sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
acc = collections.OrderedDict()
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])
            if match and match not in k:
                continue
            a = acc.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, s) in cs.items():
        print("   %-28s %14.0f per launch (%d launches)" % (c, s / n, n))
