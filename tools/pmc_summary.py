"""Mean per-launch PMC counter values per kernel from a rocprofv3 counter_collection.csv."""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:56]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
