"""Per-kernel durations and launch gaps of the steady state of a `rocprofv3 --kernel-trace --output-format csv` run:
python tools/trace_summary.py TRACE.csv KERNELS_PER_STEP [STEPS]   (the last STEPS steps of the trace are summarised)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per_step = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-per_step * steps:]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
busy = sum(dur(r) for r in tail)
gaps = sorted(int(tail[i + 1]["Start_Timestamp"]) - int(tail[i]["End_Timestamp"]) for i in range(len(tail) - 1))
print(f"# {len(tail)} dispatches = {steps} steps x {per_step} kernels; per step: wall {span / steps / 1e3:.1f} us, sum of kernel durations "
      f"{busy / steps / 1e3:.1f} us; gap between consecutive dispatches: median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us")
d = collections.defaultdict(list)
for r in tail:
    d[(r["Kernel_Name"].split("(")[0][:72], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append(dur(r))
print("%-74s %-18s %9s %10s %10s" % ("kernel", "grid (threads)", "per step", "avg us", "us/step"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%-74s %-18s %9.2f %10.1f %10.1f" % (k[0], f"{k[1]}x{k[2]}x{k[3]}", len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / steps / 1e3))
