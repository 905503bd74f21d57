"""print a rocprofv3 --stats kernel_stats.csv found under a directory (name, calls, total ms, average us, share)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print("%-70s %6s %10.3f ms %10.1f us %6s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
