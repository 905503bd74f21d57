"""sum a rocprofv3 --pmc pass per kernel (average per dispatch of the 8 longest dispatches of each kernel)"""
import csv, glob, re, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d)
    for k in sorted(acc):
        print("  %-40s" % k[:40], "  ".join("%s=%.4g" % (c, sum(sorted(v)[-8:]) / min(8, len(v))) for c, v in sorted(acc[k].items())))
