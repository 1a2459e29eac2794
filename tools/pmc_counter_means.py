import csv,glob,sys
from collections import defaultdict
acc=defaultdict(list)
for path in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"]==sys.argv[2]: acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
for k,v in acc.items(): print(k, len(v), "mean KiB", sum(v)/len(v), "= GB", sum(v)/len(v)*1024/1e9)
