"""Scan `hipcc -S` listings for serialized load chains: runs of (one or two global loads, then a wait for all of them) inside one kernel.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o k.s csrc/<unit>.hip && python tools/scan_waits.py k.s
Prints (file, longest run, kernel) for every kernel with a run of >= 3 such groups: each group is an exposed L2 / HBM round trip."""
import re, sys
# find runs of (few loads, wait for ~all) patterns: >= 6 consecutive "load(s) then s_waitcnt vmcnt(<=1)" groups inside one kernel
for f in sys.argv[1:]:
    name = None; seq = []; best = {}
    for line in open(f):
        m = re.match(r"^(_Z\w+):", line)
        if m: name = m.group(1); seq = []; continue
        t = line.strip()
        if t.startswith("global_load") or t.startswith("buffer_load"): seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1)); seq.append("W%d" % min(n, 9))
        elif t.startswith("s_endpgm"):
            s = "".join(x if x == "L" else "w" if x in ("W0", "W1") else "x" for x in seq)
            runs = re.findall(r"(?:L{1,2}w{1,2}){3,}", s)
            if runs: best[name] = max(len(re.findall(r"L{1,2}w{1,2}", r)) for r in runs)
            seq = []
    for k, v in sorted(best.items(), key=lambda kv: -kv[1]): print(f, v, k[:110])
