"""Turn the raw rocprofv3 / HIP-event outputs of tools/collect_profiles_r04.sh (gpurun_out/prof_r04/) into the committed profiles/r04_* files:
    python tools/summarise_profiles_r04.py [COMMIT [SECTIONS]]      SECTIONS: comma list of cql, algos, few (default all) -- the raw directory may
    hold sections collected at different commits
kernel stats csv (copied), dispatch classes (each kernel's dispatches split into a long and a short duration class: the many-row and the
256-row launches of one kernel), PMC traffic summaries (tools/pmc_summary.py corrections), SQ counter tables, few-runs traces, tag tables."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r04")
DST = os.path.join(ROOT, "profiles")
SECTIONS = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else {"cql", "algos", "few", "p2"}
commit = sys.argv[1] if len(sys.argv) > 1 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()


def one(pattern):
    f = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def short(name):
    return re.sub(r"^void orl::|^orl::|\(.*$", "", name)


def classes(trace_csv, out, note):
    rows = list(csv.DictReader(open(trace_csv)))
    per = collections.defaultdict(list)
    for r in rows:
        per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out_rows = []
    for k, v in per.items():
        lo, hi = min(v), max(v)
        cut = (lo * hi) ** 0.5 if hi > 4 * lo else None
        for cls in (("long", "short") if cut else ("",)):
            d = [u for u in v if (cut is None or (u > cut) == (cls == "long"))]
            if d:
                out_rows.append((sum(d), k, cls, len(d), sum(d) / len(d), min(d), max(d)))
    with open(out, "w") as f:
        f.write(f"# {note}\n# kernel,duration class,dispatches,avg us,min us,max us,total ms\n")
        for tot, k, cls, n, avg, mn, mx in sorted(out_rows, reverse=True):
            f.write(f"\"{k}\",{cls},{n},{avg:.1f},{mn:.1f},{mx:.1f},{tot / 1e3:.2f}\n")


def copy_stats(tag, dst_name, note):
    f = one(f"{tag}/**/*kernel_stats.csv")
    if not f:
        print("missing", tag); return
    with open(os.path.join(DST, dst_name), "w") as o:
        o.write(f"# {note} (commit {commit})\n" + open(f).read())
    t = one(f"{tag}/**/*kernel_trace.csv")
    if t:
        classes(t, os.path.join(DST, dst_name.replace("kernel_stats", "dispatch_classes")), note + f" (commit {commit}): dispatches split by duration class")


def run_tool(tool, args, out, header):
    txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", tool)] + args).decode()
    with open(os.path.join(DST, out), "w") as f:
        f.write(header + txt)
    return txt


B = "python3 bench.py --steps 20 --warmup 5 --no-sides --no-cpu-baseline --profile-steps 0 --min-reps 1 --min-seconds 0"
if "cql" in SECTIONS:
  copy_stats("default_stats", "r04_cql_default_2x96_kernel_stats.csv", f"rocprofv3 --kernel-trace --stats -- {B}  (2 engines x 96 runs, split precision; with two engines wall durations of one engine's launches include waiting behind the other's)")
  copy_stats("1x128_stats", "r04_cql_1x128_kernel_stats.csv", f"rocprofv3 --kernel-trace --stats -- {B} --engines-per-gpu 1 --runs-per-gpu 128")
for a in (("iql", "td3bc", "edac", "cql_h3") if "algos" in SECTIONS else ()):
    copy_stats(f"{a}_stats", f"r04_{a}_128runs_kernel_stats.csv", f"rocprofv3 --kernel-trace --stats -- python3 tools/algo_run.py {a} 128 1 30")
    t = os.path.join(SRC, f"tags_{a}.txt")
    if os.path.exists(t):
        shutil.copy(t, os.path.join(DST, f"r04_hip_event_tags_{a}_128runs.txt"))
    if one(f"{a}_FETCH_SIZE/**/*counter_collection.csv") and one(f"{a}_WRITE_SIZE/**/*counter_collection.csv"):
        run_tool("pmc_summary.py", [os.path.join(SRC, f"{a}_FETCH_SIZE"), os.path.join(SRC, f"{a}_WRITE_SIZE")], f"r04_pmc_summary_{a}_128runs.md",
                 f"## HBM traffic per launch, {a.upper()} at 128 runs, split precision (commit {commit})\nseparate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `python3 tools/algo_run.py {a} 128 1 30`; gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x 2, counters in KiB)\n\n")
for t, name in (("tags_1x128.txt", "r04_hip_event_tags_1x128.txt"), ("tags_fp32_1x128.txt", "r04_hip_event_tags_fp32_1x128.txt")):
    if "cql" in SECTIONS and os.path.exists(os.path.join(SRC, t)):
        shutil.copy(os.path.join(SRC, t), os.path.join(DST, name))
for b, name in (("bench_1x128.json", "r04_bench_1x128.json"), ("bench_fp32_1x128.json", "r04_bench_fp32_1x128.json")):
    if "cql" in SECTIONS and os.path.exists(os.path.join(SRC, b)):
        shutil.copy(os.path.join(SRC, b), os.path.join(DST, name))
# PMC traffic of the dominant kernels, one engine x 96 runs in the one-round decomposition of the two-engine default
if "cql" in SECTIONS and one("1x96_FETCH_SIZE/**/*counter_collection.csv") and one("1x96_WRITE_SIZE/**/*counter_collection.csv"):
    txt = run_tool("pmc_summary.py", [os.path.join(SRC, "1x96_FETCH_SIZE"), os.path.join(SRC, "1x96_WRITE_SIZE")], "r04_pmc_summary_1x96_one_round.md",
                   f"## HBM traffic per launch, CQL, one engine x 96 runs, ORL_WS_ONE_ROUND=1 (the decomposition of the two-engine default), split precision (commit {commit})\nseparate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `{B} --engines-per-gpu 1 --runs-per-gpu 96`; gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x 2, counters in KiB)\n\n")
    traffic = {}
    tags = {"ws_fwd_kernel<true, true, false, false, false, true>": "critic.fwd1", "ws_wgrad_kernel<2>": "critic.bwd.wgrad1",
            "ws_dgrad_w0_kernel<true, false, false>": "critic.bwd.dgrad1"}      # (template argument lists as of the plain / discard flavours)
    for line in txt.splitlines():
        m = re.match(r"\| `([^`]+)` \((\d+)( long| short)?\) \| (\d+) \| ([\d.]+) \| (\d+) \| (\d+) \|", line)
        if m and m.group(1) in tags and (m.group(3) or " long").strip() == "long":
            rd, wr = float(m.group(6)) * 1e6, float(m.group(7)) * 1e6
            traffic[tags[m.group(1)]] = dict(bytes_per_launch=rd + wr, read_bytes=rd, written_bytes=wr, runs_per_gpu=96, precision=1, commit=commit,
                                             avg_us_under_pmc=float(m.group(5)),
                                             source="profiles/r04_pmc_summary_1x96_one_round.md: rocprofv3 --kernel-trace --pmc FETCH_SIZE (x2 gfx950 correction) and --pmc WRITE_SIZE, separate passes, one engine x 96 runs, ORL_WS_ONE_ROUND=1")
    if traffic:
        json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
        print("pmc_traffic.json:", {k: round(v["bytes_per_launch"] / 1e6) for k, v in traffic.items()})
sq = [os.path.join(SRC, d) for d in ("1x128_sq1", "1x128_sq2", "1x128_sq3") if "cql" in SECTIONS and one(f"{d}/**/*counter_collection.csv")]
if sq:
    run_tool("pmc_sq.py", sq, "r04_pmc_sq_counters_1x128.txt",
             f"# SQ counters per kernel (means per dispatch), CQL one engine x 128 runs, split precision (fp16 planes), commit {commit}: three separate rocprofv3 --kernel-trace --pmc passes\n"
             "# (units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_VALU_MFMA_COEXEC_CYCLES count cycles)\n")
# precision 2 (three fp16 planes in the critic launches, fp32 MFMA elsewhere), one engine x 128 runs
if "p2" in SECTIONS:
    copy_stats("p2_stats", "r04_cql_p2_1x128_kernel_stats.csv", f"rocprofv3 --kernel-trace --stats -- {B} --engines-per-gpu 1 --runs-per-gpu 128 --precision 2")
    for src, name in (("tags_p2_1x128.txt", "r04_hip_event_tags_p2_1x128.txt"), ("bench_p2_1x128.json", "r04_bench_p2_1x128.json")):
        if os.path.exists(os.path.join(SRC, src)):
            shutil.copy(os.path.join(SRC, src), os.path.join(DST, name))
    if one("p2_FETCH_SIZE/**/*counter_collection.csv") and one("p2_WRITE_SIZE/**/*counter_collection.csv"):
        run_tool("pmc_summary.py", [os.path.join(SRC, "p2_FETCH_SIZE"), os.path.join(SRC, "p2_WRITE_SIZE")], "r04_pmc_summary_p2_1x128.md",
                 f"## HBM traffic per launch, CQL, one engine x 128 runs, precision 2 (commit {commit})\nseparate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of `{B} --engines-per-gpu 1 --runs-per-gpu 128 --precision 2`; gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x 2, counters in KiB)\n\n")
    parts = []
    for f, title in (("few_runs_p0.txt", "precision 0 (exact fp32 MFMA)"), ("few_runs_p2.txt", "precision 2 (three fp16 planes in the critic launches, fp32 MFMA elsewhere)")):
        if os.path.exists(os.path.join(SRC, f)):
            parts.append(f"## {title}: `tools/few_runs_ab.py - --runs 1 8 16 32 128 --precision ...`\n```\n{open(os.path.join(SRC, f)).read()}```\n")
    if parts:
        open(os.path.join(DST, "r04_fp32_class_by_runs.md"), "w").write(f"# CQL gradient-steps/s by runs per engine, fp32-class arithmetic against exact fp32 (one engine, graph replay; commit {commit})\n\n" + "\n".join(parts))
# few runs
out = [f"# Kernel nodes per step in the few-runs regime (graph replay; rocprofv3 --kernel-trace, tools/trace_summary.py), commit {commit}\n"]
for r in ((1, 8) if "few" in SECTIONS else ()):
    t = one(f"few_{r}/**/*kernel_trace.csv")
    if not t:
        continue
    rows = sorted(csv.DictReader(open(t)), key=lambda x: int(x["Start_Timestamp"]))
    names = [short(x["Kernel_Name"]) for x in rows]
    idx = [i for i, n in enumerate(names) if n.startswith("k_prepare")]
    per_step = idx[-1] - idx[-2] if len(idx) > 2 else 30
    txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "trace_summary.py"), t, str(per_step), "20"]).decode()
    out.append(f"## One engine x {r} run(s) (`bench.py --engines-per-gpu 1 --runs-per-gpu {r}`): {per_step} kernel nodes per step\n```\n{txt}```\n")
for f, title in (("few_runs_ab.txt", "split precision"), ("few_runs_ab_fp32.txt", "exact fp32")):
    if "few" in SECTIONS and os.path.exists(os.path.join(SRC, f)):
        out.append(f"## Fused actor phase (ORL_FUSE_SMALL=1, the default) against the separate launches (=0), {title}: `tools/few_runs_ab.py`, engines created alternately in one process\n```\n{open(os.path.join(SRC, f)).read()}```\n")
if len(out) > 1:
    open(os.path.join(DST, "r04_kernel_trace_few_runs.md"), "w").write("\n".join(out))
print("done")
