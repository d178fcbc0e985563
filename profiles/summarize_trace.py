"""Per-kernel, per-grid summary of a `rocprofv3 --kernel-trace --output-format csv` trace:
python profiles/summarize_trace.py <kernel_trace.csv> [min_share_percent]"""
import csv
import sys
from collections import defaultdict


def main(path, min_share=0.05):
    groups = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
            short = name.split('(')[0]
            key = (short[:70], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])),
                   int(r['VGPR_Count']), int(r['Accum_VGPR_Count']), int(r['LDS_Block_Size']))
            groups[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    total = sum(sum(v) for v in groups.values())
    print('%-72s %9s %6s %5s %5s %7s %12s %8s' % ('kernel', 'blocks', 'calls', 'vgpr', 'agpr', 'lds', 'avg_us', 'share%'))
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        share = 100.0 * sum(v) / total
        if share < min_share:
            continue
        print('%-72s %9d %6d %5d %5d %7d %12.1f %8.2f' % (key[0], key[1], len(v), key[2], key[3], key[4],
                                                         sum(v) / len(v) / 1e3, share))


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.05)
