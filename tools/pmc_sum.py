import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"][:60]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    print(k, {c: (round(v / n[(k, c)], 1)) for c, v in acc[k].items()})
