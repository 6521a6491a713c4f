"""quick bench wrapper: python tools/bq.py [bench args] -> one short line"""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--profile-steps", "0", "--no-single"] + sys.argv[1:],
                     capture_output=True, text=True)
try:
    d = json.loads(out.stdout.strip().splitlines()[-1])
    print(os.environ.get("TAG", ""), "R", d["config"]["runs_per_gpu"], "steps/s", round(d["value"]), "ms", round(d["ms_per_step"], 4), flush=True)
except Exception as e:
    print("bench failed:", e, out.stdout[-500:], out.stderr[-1500:])
