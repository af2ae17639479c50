#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc pass per GEMM kernel form (tall / 8phase / 8phase_t).  python tools/debug/pmc_sum_tall.py <counter_collection.csv>"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    form = "tall" if "tall_kernel" in k else ("8phase_t" if "8phase_t_kernel" in k else ("8phase" if "8phase_kernel" in k else None))
    if form is None:
        continue
    acc[form][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, d in sorted(acc.items()):
    print(kn, "launches", len(next(iter(d.values()))), {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
