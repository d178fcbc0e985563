"""Per-kernel averages of every counter found in rocprofv3 counter_collection.csv files under a directory."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
    seen = defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
        if pat and pat not in k:
            continue
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        seen[(k, row['Counter_Name'])].add(row['Dispatch_Id'])
    for (k, c), ids in seen.items():
        cnt[k][c] += len(ids)
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print('   %-44s %16.1f per launch (%d launches)' % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
