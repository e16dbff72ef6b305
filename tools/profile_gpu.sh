#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline numbers are checked against (run on the GPU box):
#   tools/profile_gpu.sh <preset> <tag> [batch] ["extra bench.py arguments"]
#                                                  e.g.  tools/profile_gpu.sh high r01_v3      (batch 32: all passes)
#                                                        tools/profile_gpu.sh high r04_v1_f16 32 "--gen-precision f16"
#                                                        tools/profile_gpu.sh medium r02_v7 1  (kernel trace only)
# Pass "stats": --kernel-trace --stats (per-kernel time).  Passes "pmc*": PMC counters, each set in its own
# run (MI355X_MICROARCH.md: separate --pmc passes; FETCH_SIZE and WRITE_SIZE do not fit one pass, and
# FETCH_SIZE is reported at half the streamed bytes on gfx950).  Every pass runs under its own `timeout`:
# a counter set the hardware refuses makes rocprofv3 abort and then hang in its finaliser.
# Raw output goes to gpurun_out/prof_<tag>_<preset>/, the summaries to profiles/.
set -u
PRESET=${1:-high}
TAG=${2:-r01}
BATCH=${3:-32}
EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${PRESET}_b${BATCH}
mkdir -p "$OUT" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
# (--parts 1: one handle / stream, as in the roofline block of the bench line, so that kernels do not overlap in the trace)
BENCH="python3 $R/bench.py --preset $PRESET --steps 3 --warmup 1 --no-cpu-baseline --no-extras --parts 1 --batch $BATCH $EXTRA"
T="timeout -k 10 240"
$T rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o s -- $BENCH > "$OUT/stats.log" 2>&1
if [ "$BATCH" = 32 ]; then
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc1" -o p -- $BENCH > "$OUT/pmc1.log" 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc2" -o p -- $BENCH > "$OUT/pmc2.log" 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc3" -o p -- $BENCH > "$OUT/pmc3.log" 2>&1
fi
python3 "$R/tools/rocprof_summary.py" "$OUT" "$R/profiles/${TAG}_${PRESET}_b${BATCH}" --preset "$PRESET" --length-scale 1.95
cp "$R/profiles/${TAG}_${PRESET}_b${BATCH}_kernel_stats.csv" "$R/profiles/${TAG}_${PRESET}_b${BATCH}_pmc.json" "$R/gpurun_out/" 2>/dev/null
# the raw traces / counter databases are tens of MiB per pass and gpurun copies back at most 64 MiB of gpurun_out/: keep the
# logs and the summaries only
rm -rf "$OUT/stats" "$OUT/pmc1" "$OUT/pmc2" "$OUT/pmc3"
