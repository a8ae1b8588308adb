"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel dispatch.
python tools/pmc_summary.py <dir with pass*/ *_counter_collection.csv> [out.md]"""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(root + "/pass*/*counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                name = name.split("(")[0].split("<")[0].strip()
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = []
    for k, cs in acc.items():
        lines.append("### `%s`" % k)
        lines.append("| counter | mean per dispatch | dispatches |")
        lines.append("|---|---|---|")
        for c, vals in sorted(cs.items()):
            lines.append("| %s | %.6g | %d |" % (c, sum(vals) / len(vals), len(vals)))
        lines.append("")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
