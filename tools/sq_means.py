import csv,glob,sys
from collections import defaultdict
acc=defaultdict(list)
for path in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "p16" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:32s} {len(v):3d} {sum(v)/len(v):18.1f}")
