#!/bin/bash
# GPU dev tool: effective shader clock of a kernel = GRBM_GUI_ACTIVE / wall time (the chip clocks to its power budget).
# usage: tools/prof_clock.sh <tag> <label> <command ...>     appends "label kernel avg_us counters clock" lines to gpurun_out/clock_<tag>.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; LABEL=$2; shift 2
OUT=$R/gpurun_out/prof_clock_$TAG/$LABEL
rm -rf "$OUT"; mkdir -p "$OUT/pass1"
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/pass1" -o p -- "$@" > "$OUT/pass1.log" 2>&1 )
python $R/tools/pmc_summary.py "$OUT" | awk -v l="$LABEL" '/^== /{k=$2; us=$4} /GRBM_GUI_ACTIVE/{split($2,a,"="); g=a[2]} /SQ_BUSY_CYCLES/{split($2,a,"="); b=a[2]} /SQ_INSTS_VALU/{split($2,a,"="); v=a[2]} /SQ_WAVE_CYCLES/{split($2,a,"="); w=a[2]; split(us,u,"="); if (u[2] > 50) printf "%-28s %-22s %10.1f us  GRBM_GUI_ACTIVE %.4g  SQ_BUSY %.4g  INSTS_VALU %.4g  WAVE_CYCLES %.4g  GUI/us %.1f  SQBUSY/us %.1f\n", l, k, u[2], g, b, v, w, g/u[2], b/u[2]}' >> $R/gpurun_out/clock_$TAG.txt
