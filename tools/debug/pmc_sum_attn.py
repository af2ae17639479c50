import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "attn_" not in k: continue
    acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, d in acc.items():
    print(kn, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
