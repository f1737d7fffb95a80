#!/bin/bash
# Data-parallel scaling sweep: bench.py at N = 1, 2, 4, 8 GPUs of this node x the three placements of the gradient
# collectives (parallel.GradReducer.overlap), one table.  Settles the default placement on the first multi-GPU node
# ("attention_windows" was chosen by single-GPU reasoning, no node was available to the build).
#
#   tools/scale_sweep.sh [max_gpus] [steps]          (default: all GPUs rocm-smi / torch sees, 10 steps)
#
# N larger than the number of visible GPUs is skipped; bench.py itself refuses to print a line whose collectives did not run
# over RCCL on all N ranks.  At N = 1 the reducer is attached with --force-reducer (bucket bookkeeping, no collectives).
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
have=$(python -c 'import torch; print(torch.cuda.device_count())')
max=${1:-$have}
steps=${2:-10}
out=${SWEEP_OUT:-gpurun_out/scale_sweep}
mkdir -p "$out"
printf "%-4s %-18s %12s %12s %10s %8s\n" N overlap samples/s ms/step rccl_ranks x_vs_N1
declare -A base
for n in 1 2 4 8; do
  if [ "$n" -gt "$max" ] || [ "$n" -gt "$have" ]; then continue; fi
  for ov in attention_windows backward after; do
    f="$out/n${n}_${ov}.json"
    extra=""
    [ "$n" -eq 1 ] && extra="--force-reducer"
    python bench.py --gpus "$n" --steps "$steps" --warmup 3 --dp-overlap "$ov" --no-cpu-baseline --no-roofline --no-extras $extra \
      > "$f" 2> "$out/n${n}_${ov}.err" || { printf "%-4s %-18s %12s\n" "$n" "$ov" "FAILED (see $out/n${n}_${ov}.err)"; continue; }
    python - "$f" "$n" "$ov" "${base[$ov]:-}" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
b = float(sys.argv[4]) if sys.argv[4] else d["value"]
print("%-4s %-18s %12.1f %12.2f %10d %8.2f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["rccl_ranks"], d["value"] / b))
PY
    if [ "$n" -eq 1 ]; then
      base[$ov]=$(python -c "import json,sys; print(json.loads([l for l in open('$f') if l.startswith('{')][-1])['value'])")
    fi
  done
done
