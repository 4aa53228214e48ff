#!/bin/bash
# conv-tile kernel: parity tests, then same-box A/B against the two-kernel conv (MATTEN_CONV_TILE=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "conv_fused_kernel or conv_tile" > gpurun_out/r4b_pytest_tile.log 2>&1; echo "tile tests rc=$?"
tail -15 gpurun_out/r4b_pytest_tile.log
if grep -q "passed" gpurun_out/r4b_pytest_tile.log && ! grep -q "failed" gpurun_out/r4b_pytest_tile.log; then
  for i in 1 2; do
    MATTEN_CONV_TILE=0 timeout 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 > gpurun_out/r4b_bench_off_$i.json 2> gpurun_out/r4b_bench_off_$i.err
    timeout 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 > gpurun_out/r4b_bench_on_$i.json 2> gpurun_out/r4b_bench_on_$i.err
  done
  python3 - <<'PY'
import json
for tag in ("off_1","on_1","off_2","on_2"):
    try:
        d=json.load(open(f"gpurun_out/r4b_bench_{tag}.json"))
        print(tag, "ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernel_ms_per_launch"].items()}, "valu ns %.3f"%d["calibration"]["before"]["valu"]["ns_per_wave_inst_per_simd"])
    except Exception as e:
        print(tag, "failed", e)
PY
fi
timeout 900 python3 -m pytest tests/test_gpu_training.py -x -q -k "eval_after_flat or flat_adam_state" > gpurun_out/r4b_pytest_train.log 2>&1; echo "train tests rc=$?"; tail -5 gpurun_out/r4b_pytest_train.log
