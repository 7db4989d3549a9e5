import csv, sys, glob, collections, re
def load(d):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.split(r"[<(]", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""), 1)[0]
            agg[n][0] += int(r["Calls"]); agg[n][1] += float(r["TotalDurationNs"]) / 1e3
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in set(a) | set(b):
    rows.append((b[k][1] - a[k][1], k, a[k][0], a[k][1], b[k][0], b[k][1]))
rows.sort(key=lambda r: -abs(r[0]))
print("%-44s %8s %12s %8s %12s %10s" % ("kernel", "calls A", "us A", "calls B", "us B", "B - A us"))
for d, k, ca, ta, cb, tb in rows[:25]:
    print("%-44s %8d %12.1f %8d %12.1f %10.1f" % (k[:44], ca, ta, cb, tb, d))
print("total A %.1f us, B %.1f us" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
