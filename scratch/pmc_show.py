import csv, glob, sys
pat = sys.argv[1]
for d in sorted(glob.glob("gpurun_out/pmcg_*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no file"); continue
    acc = {}
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(d.split("pmcg_")[1], {k: round(sum(v) / len(v)) for k, v in acc.items()})
