"""Summarise rocprofv3 --pmc passes per kernel: python tools/pmc_summary.py FETCH_DIR WRITE_DIR [MFMA_DIR]

Each DIR holds one `*_counter_collection.csv` of a separate `rocprofv3 --kernel-trace --pmc <COUNTER>` pass of the same command.
Corrections per MI355X_MICROARCH.md (HBM section): counters in KiB, FETCH_SIZE doubled on gfx950, WRITE_SIZE exact.
Dispatches of one kernel are split by grid size and, when their durations differ by more than 4x, into a long and a short class
(the many-row and the 256-row launches of the same kernel)."""
import csv, glob, os, sys, collections, re


def load(d):
    f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)     # newest pass
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = (re.sub(r"^void orl::|\(.*$", "", r["Kernel_Name"]), int(r["Grid_Size"]))
        per[k].append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    acc = {}
    for k, v in per.items():
        lo, hi = min(u for _, u in v), max(u for _, u in v)
        cut = (lo * hi) ** 0.5 if hi > 4 * lo else None        # the same kernel on many-row and on 256-row batches: two duration classes
        for c, u in v:
            kk = k + (("long" if u > cut else "short") if cut else "",)
            a = acc.setdefault(kk, [0, 0.0, 0.0])
            a[0] += 1; a[1] += c; a[2] += u
    return acc


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    mfma = load(sys.argv[3]) if len(sys.argv) > 3 else {}
    rows = []
    for k, (n, v, us) in fetch.items():
        w = write.get(k, [1, 0.0, 0.0]); m = mfma.get(k)
        busy = (m[1] / m[0]) / ((m[2] / m[0]) * 1e-6 * 2.4e9 * 1024) if m else None
        rows.append((us, k, n, us / n, 2 * v / n * 1024 / 1e6, w[1] / w[0] * 1024 / 1e6, busy))
    print("| kernel (grid) | dispatches | avg us | read MB/launch (FETCH x2) | written MB/launch | MFMA-pipe busy |\n|---|---|---|---|---|---|")
    for us, k, n, avg, rd, wr, busy in sorted(rows, reverse=True)[:14]:
        print("| `%s` (%d%s) | %d | %.1f | %.0f | %.0f | %s |" % (k[0][:90], k[1], " " + k[2] if k[2] else "", n, avg, rd, wr, "%.0f %%" % (100 * busy) if busy is not None else "-"))


if __name__ == "__main__":
    main()
