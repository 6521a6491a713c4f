"""Per-kernel means of every counter in rocprofv3 --pmc passes (multi-counter CSVs): python tools/pmc_sq.py DIR [DIR ...]
Dispatches of one kernel whose durations differ by more than 4x are split into a long and a short class."""
import csv, glob, os, sys, collections, re


def main():
    rows = collections.defaultdict(lambda: collections.defaultdict(list))      # (kernel, class) -> counter -> values
    dur = collections.defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = re.sub(r"^void orl::|\(.*$", "", r["Kernel_Name"])
                per[k].append((r["Counter_Name"], float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
            for k, v in per.items():
                lo, hi = min(u for _, _, u in v), max(u for _, _, u in v)
                cut = (lo * hi) ** 0.5 if hi > 4 * lo else None
                for name, c, u in v:
                    kk = (k, ("long" if u > cut else "short") if cut else "")
                    rows[kk][name].append(c)
                    dur[kk].append(u)
    order = sorted(rows, key=lambda kk: -sum(dur[kk]))[:8]
    for kk in order:
        n = len(dur[kk])
        print(f"## {kk[0][:100]} {kk[1]}  avg {sum(dur[kk]) / n:.1f} us")
        for name, vals in sorted(rows[kk].items()):
            print(f"   {name:32s} {sum(vals) / len(vals):16.0f}")


if __name__ == "__main__":
    main()
