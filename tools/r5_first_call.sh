#!/bin/bash
# First GPU call of the next session (everything round 4 wrote after its GPU budget ran out, in one box):
#   gpurun --timeout 600 -- 'bash tools/r5_first_call.sh'
# 1. the full-model train scope (capdec_train_set_scope(1)) has never run on a GPU: its parity test, in its own process
# 2. CAPDEC_TRAIN_F16X2=1 (backward GEMMs of the train step on the two-fp16-plane kernels): the three validated parity cases
#    under the knob, then the A/B of the train-step bench line (default: native fp32 MFMA GEMM, 21.7 of 39.5 ms)
# 3. the per-kernel table of the train step (tools/train_profile.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
echo "== 1. full-model train scope (never run before)"
CAPDEC_TEST_UNVALIDATED=1 timeout 120 python -m pytest tests/test_hip_parity.py -q -m gpu -k "train_step_full_model" 2>&1 | tail -30 | tee "$OUT/r5_train_full.txt"
echo "== 2a. frozen-scope parity under CAPDEC_TRAIN_F16X2=1"
CAPDEC_TRAIN_F16X2=1 timeout 120 python -m pytest tests/test_hip_parity.py -q -m gpu -k "train_step_frozen" 2>&1 | tail -30 | tee "$OUT/r5_train_f16x2_parity.txt"
echo "== 2b. train-step bench A/B"
for kv in X=0 CAPDEC_TRAIN_F16X2=1; do
    echo "-- $kv"
    env $kv timeout 100 python bench.py --workload train_step --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | tee "$OUT/r5_train_bench_${kv%%=*}.json" | cut -c1-700
done
echo "== 3. per-kernel table"
bash tools/train_profile.sh r5 2>&1 | tail -45
