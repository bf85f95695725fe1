#!/bin/bash
# the round-5 profiles (same recipe as round 4) of the four GPU configs (GPU box, repo root):  bash tools/experiments/prof_all_r05.sh "C2 C5"
# C3 / C5 at reduced spp (per-sample figures do not depend on it; the launch is then seconds, not tens of seconds, per counter pass)
for w in ${1:-C2 C4 C3 C5}; do
  EXTRA=""
  case $w in
    C2) WLA="--workload C2"; N=2123366400; EXTRA="single";;
    C4) WLA="--workload C4"; N=4294967296;;
    C3) WLA="--workload C3 --spp 2000"; N=1280000000;;
    C5) WLA="--workload C5 --spp 32"; N=536870912;;
    C5E) WLA="--workload C5 --spp 32 --empirical-trees"; N=536870912;;
  esac
  lw=$(echo $w | tr A-Z a-z)
  OUT=gpurun_out/prof_$lw WL="$WLA" STEPS=1 WARMUP=1 TMO=280 bash tools/experiments/prof_r04.sh pmc cache $EXTRA || echo "prof $w failed rc=$?"
  python tools/experiments/summarize_pmc.py gpurun_out/prof_$lw gpurun_out/${lw}_pmc_summary.json $N 2 > gpurun_out/${lw}_pmc.log 2>&1
  cp gpurun_out/prof_$lw/trace/*/*kernel_stats.csv gpurun_out/${lw}_kernel_stats.csv 2>/dev/null
  tail -3 gpurun_out/${lw}_pmc.log
done
