"""Condenses the rocprofv3 PMC passes of tools/experiments/prof_*.sh into one JSON under profiles/."""
import collections
import csv
import glob
import json
import sys

import os
src, dst, samples = sys.argv[1], sys.argv[2], float(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # frames rendered per pass (steps + warm-up); 0 = average per DISPATCH
cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "bench.py --steps 2 --warmup 1 --no-cpu"
out = {"command": f"rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 {cmd} (separate passes; see tools/experiments/prof_r03.sh)",
       "samples_per_dispatch": samples, "counters_per_dispatch": {}}
single = {}
for f in sorted(glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    is_single = "/pmc_sq2_single/" in f
    for r in csv.DictReader(open(f)):
        # the production build of the megakernel only (..., false, false>): not the probe (COST) launch that precedes it
        if "render_kernel" in r["Kernel_Name"] and ", true>(" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            # (the first dispatch of the file: a frame's first launch, not the short second launch of exact re-treeing)
            if "dispatch" not in out:
                out["dispatch"] = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        if is_single:      # the single-launch shape (VK_NO_DUAL_LAUNCH=1): kept apart
            single[k] = sum(v) / (frames if frames else len(v))
            continue
        # per frame: a frame may be more than one dispatch of the production kernel (the dual launch of sphere-only LDS scenes)
        out["counters_per_dispatch"][k] = sum(v) / (frames if frames else len(v))
        out["dispatches_per_frame"] = len(v) / frames if frames else 1
c = out["counters_per_dispatch"]
d = {}
if "FETCH_SIZE" in c:
    d["hbm_read_bytes_per_dispatch"] = {"FETCH_SIZE_KB_x1024": c["FETCH_SIZE"] * 1024, "with_gfx950_x2_correction_upper_bound": 2 * c["FETCH_SIZE"] * 1024}
if "WRITE_SIZE" in c:
    d["hbm_write_bytes_per_dispatch"] = c["WRITE_SIZE"] * 1024
if "SQ_THREAD_CYCLES_VALU" in c:
    d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
    w = c["SQ_WAVE_CYCLES"]
    d["wave_time_split"] = {"issuing": c["SQ_ACTIVE_INST_ANY"] / w, "waiting(s_waitcnt)": c["SQ_WAIT_ANY"] / w, "issue_stall": c["SQ_WAIT_INST_ANY"] / w}
if single.get("SQ_WAVE_CYCLES"):
    w = single["SQ_WAVE_CYCLES"]
    d["wave_time_split_single_launch_shape"] = {"issuing": single["SQ_ACTIVE_INST_ANY"] / w, "waiting(s_waitcnt)": single["SQ_WAIT_ANY"] / w,
        "issue_stall": single["SQ_WAIT_INST_ANY"] / w, "waves": single.get("SQ_WAVES"),
        "note": "VK_NO_DUAL_LAUNCH=1: 2 x 768-thread workgroups per CU (six waves per SIMD), the shape that runs under the counter profiler as "
                "it is timed; the 16 + 12-wave dual launch is serialised by the profiler, so `wave_time_split` above describes a 16-wave and "
                "a 12-wave run one after the other, not the timed kernel"}
    out["counters_single_launch_shape"] = single
if "TCC_ATOMIC_sum" in c:
    d["l2_write_requests_per_frame"] = {"writes": c.get("TCC_WRITE_sum"), "atomics": c["TCC_ATOMIC_sum"], "to_memory_writes": c.get("TCC_EA0_WRREQ_sum"),
        "to_memory_atomics": c.get("TCC_EA0_ATOMIC_sum"),
        "note": "the kernel's vector-memory writes: 64-bit atomic adds of the fixed-point pixel sums (unit flushes + stragglers) against plain stores "
                "(register spills, redo queue entries, per-sample debug dumps)"}
if "SQ_LDS_BANK_CONFLICT" in c:
    d["lds_bank_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
if "SQ_INSTS_VALU" in c and "SQ_BUSY_CYCLES" in c:
    # VALU issue: one wave64 VALU instruction occupies its SIMD for 2 cycles (MI355X_MICROARCH.md); 1024 SIMDs
    d["valu_wave_instr_per_sample"] = c["SQ_INSTS_VALU"] / samples
if "TCC_HIT_sum" in c:
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    d["l2_misses_per_sample"] = c["TCC_MISS_sum"] / samples
if "TCP_TCC_READ_REQ_sum" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
    d["l1_miss_rate(TCP->TCC reads / TCP accesses)"] = c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
if "TCC_EA0_RDREQ_sum" in c:
    d["memory_read_requests_per_sample"] = c["TCC_EA0_RDREQ_sum"] / samples
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_FLAT", "SQ_INSTS_VALU_TRANS"):
    if k in c:
        d[k + "_per_sample"] = c[k] / samples
out["derived"] = d
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(d, indent=1))
print(out.get("dispatch"))
