"""GPU-box helper: hardware counters of the render kernel for one bench workload, one rocprofv3 --pmc pass per group (kernel trace
only, the program directly after `--`: MI355X_MICROARCH.md's recipe).
    python tools/experiments/prof_counters.py OUT.json "<bench args>" "CTR_A CTR_B" "CTR_C ..."
    python tools/experiments/prof_counters.py OUT.json "@<scene> <width> <spp>" "CTR_A CTR_B" ...      (tools/experiments/render_once.py instead of bench.py)
Writes {counter: sum over the production render-kernel dispatches of the ONE frame rendered — a frame is two or more dispatches
when the library launches 16 + 12 waves per CU, and three more with exact re-treeing's second and fallback launches} plus
`dispatches` = their number."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out_path, bench_args, groups = sys.argv[1], sys.argv[2].split(), [g.split() for g in sys.argv[3:]]
res = {"bench_args": bench_args, "counters": {}}
for group in groups:
    tmp = tempfile.mkdtemp(prefix="vk_pmc_", dir="/tmp")
    if bench_args[0].startswith("@"):
        prog = [os.path.join(ROOT, "tools", "experiments", "render_once.py"), bench_args[0][1:]] + bench_args[1:]
    else:
        prog = [os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu", "--no-verify", "--no-also", "--no-traffic"] + bench_args
    cmd = ["rocprofv3", "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable] + prog
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=int(os.environ.get("PASS_TIMEOUT", "150")))
    except subprocess.TimeoutExpired:
        res.setdefault("errors", []).append({"group": group, "timeout": True})
        break
    vals = {c: [] for c in group}
    for f in glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "render_kernel" in row["Kernel_Name"] and ", true>(" not in row["Kernel_Name"] and row["Counter_Name"] in vals:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
                res["kernel"] = row["Kernel_Name"]
    for c in group:
        res["counters"][c] = sum(vals[c]) if vals[c] else None       # one frame was rendered: its dispatches add up
        res["dispatches"] = len(vals[c]) if vals[c] else res.get("dispatches")
    if r.returncode != 0:
        res.setdefault("errors", []).append({"group": group, "rc": r.returncode, "stderr": r.stderr[-500:]})
    res.setdefault("stdout", []).append(r.stdout[-300:])
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            res["value"], res["ms_per_step"], res["workload"] = d["value"], d["ms_per_step"], d["config"]["workload"]
    shutil.rmtree(tmp, ignore_errors=True)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
