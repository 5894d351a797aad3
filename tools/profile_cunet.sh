#!/bin/bash
# rocprofv3 evidence for config 2 (cunet/art x2, batch 4, tile 256, 1080p) on the GPU box:  tools/profile_cunet.sh <tag>
# Same recipe as profile_round.sh (kernel-trace stats, then separate PMC passes) over tools/op_times.py.
set -u
TAG=${1:-cunet}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
CMD="python3 tools/op_times.py cunet/art 2 1 4 256 1080 1920"
$CMD > "$OUT/op_times.txt" 2>&1
# per-kernel evidence with every pass in one piece on one stream, as in profile_round.sh
export W2X_GROUPS=1
export W2X_RENDER_PARTS=1     # render() as one part: every launch of the traced run covers all live tiles (the two-part render() of round 4 would add half-size launches to the averages)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o w2x -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o w2x -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w2x -- $CMD > "$OUT/pmc_write.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA -d "$OUT/pmc_sq" -o w2x -- $CMD > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d "$OUT/pmc_mfma" -o w2x -- $CMD > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d "$OUT/pmc_lds" -o w2x -- $CMD > "$OUT/pmc_lds.log" 2>&1
for d in pmc_fetch pmc_write pmc_sq pmc_mfma pmc_lds; do python3 tools/pmc_summary.py "$OUT/$d" > "$OUT/$d.summary.txt" 2>&1; done
find "$OUT" -name "*_kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 tools/mfma_util.py "$OUT/pmc_mfma" > "$OUT/mfma_util.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
du -sh "$OUT"; head -3 "$OUT/op_times.txt"; head -14 "$OUT/kernel_stats.csv" | cut -c1-160
