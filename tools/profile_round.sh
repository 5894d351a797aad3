#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:  tools/profile_round.sh <tag>
#   kernel-trace stats of the default bench workload, then separate PMC passes (HBM bytes, SQ cycles) as
#   MI355X_MICROARCH.md prescribes (no trace domains combined with --pmc).  Results land in gpurun_out/<tag>/.
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --op-times > "$OUT/op_times.txt" 2>&1
# Per-kernel evidence is collected with every pass in one piece on one stream (W2X_GROUPS=1): a launch is then the 45-tile launch
# that bench.py's roofline block prices (its HIP-event durations come from the engine's profiling mode, which does not split either);
# the timed region of bench.json above runs each pass as two tile groups on two streams, whose half-size launches overlap.
export W2X_GROUPS=1
export W2X_RENDER_PARTS=1     # render() as one part: every launch of the traced run covers all live tiles (the two-part render() of round 4 would add half-size launches to the averages)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o w2x -- python3 bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline > "$OUT/trace.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o w2x -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w2x -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA -d "$OUT/pmc_sq" -o w2x -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d "$OUT/pmc_mfma" -o w2x -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/pmc_mfma.log" 2>&1
# vector and matrix instructions executing in the SAME cycle (MI355X_MICROARCH.md, two waves per SIMD, item 9) and the LDS pipe
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -d "$OUT/pmc_coexec" -o w2x -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/pmc_coexec.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES -d "$OUT/pmc_lds" -o w2x -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/pmc_lds.log" 2>&1
for d in pmc_fetch pmc_write pmc_sq pmc_mfma pmc_coexec pmc_lds; do python3 tools/pmc_summary.py "$OUT/$d" > "$OUT/$d.summary.txt" 2>&1; done
find "$OUT" -name "*_kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 tools/mfma_util.py "$OUT/pmc_mfma" "$OUT/kernel_stats.csv" > "$OUT/mfma_util.txt" 2>&1
python3 tools/pmc_traffic.py "$OUT" "$OUT/pmc_traffic.json" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 2 --warmup 1, profiles/$TAG" > "$OUT/pmc_traffic.txt" 2>&1
# keep the merge-back small: raw traces are not needed once summarised
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
du -sh "$OUT"; cat "$OUT/bench.json"; head -12 "$OUT/kernel_stats.csv"
