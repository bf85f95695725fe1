#!/bin/bash
# GPU box: A/B of device-library builds, interleaved.  usage: ab_libs.sh "libA libB ..." WL reps rounds
libs=$1; wl=${2:-C2,C3,C4,C5}; reps=${3:-3}; rounds=${4:-2}
l=""; for r in $(seq $rounds); do for x in $libs; do l="$l,vecchio_amd/lib/exp/$x.so"; done; done
python tools/experiments/perf_quick.py --libs ${l#,} --wl $wl --reps $reps 2>&1 | grep -v "^+"
