#!/bin/bash
# SQ counter passes over tools/ab/attn_ab (both schedules of the C = 96 attention, headline size):  tools/ab/profile_attn_ab.sh <out dir>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/attn_ab}
mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I waifu2x-tensorrt_amd/csrc tools/ab/attn_ab.hip tools/ab/k_swinattn96_g2.hip waifu2x-tensorrt_amd/csrc/k_swinattn96.hip -o /tmp/attn_ab 2> "$OUT/build.log" || exit 1
/tmp/attn_ab timing > "$OUT/run.txt" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES -d "$OUT/pmc1" -o ab -- /tmp/attn_ab timing > "$OUT/pmc1.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM -d "$OUT/pmc2" -o ab -- /tmp/attn_ab timing > "$OUT/pmc2.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS -d "$OUT/pmc3" -o ab -- /tmp/attn_ab timing > "$OUT/pmc3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o ab -- /tmp/attn_ab timing > "$OUT/trace.log" 2>&1
for d in pmc1 pmc2 pmc3; do python3 tools/pmc_summary.py "$OUT/$d" > "$OUT/$d.summary.txt" 2>&1; done
find "$OUT" -name "*_kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name "*counter_collection.csv" -size +4M -delete; find "$OUT" -name "*kernel_trace.csv" -size +4M -delete
cat "$OUT/run.txt" | tail -3; cat "$OUT"/pmc1.summary.txt | head -40
