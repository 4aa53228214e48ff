#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "tp_kernels or config3 or gate_and_batchnorm or n100 or dead_output or full_size or conv_fused_kernel" > gpurun_out/r4g_pytest.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4g_pytest.log
for i in 1 2; do
  MATTEN_TP_MERGE=0 timeout 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 > gpurun_out/r4g_bench_off_$i.json 2>/dev/null
  timeout 300 python3 bench.py --no-extras --no-cpu-baseline --steps 30 > gpurun_out/r4g_bench_on_$i.json 2>/dev/null
done
python3 - <<'PY'
import json
for tag in ("off_1","on_1","off_2","on_2"):
    try:
        d=json.load(open(f"gpurun_out/r4g_bench_{tag}.json"))
        print(tag, "ms/step %.3f"%d["ms_per_step"], {k.split('/')[0][:4]+k.split('/')[1][-5:]:round(v,3) for k,v in d["kernel_ms_per_launch"].items()}, "valu ns %.3f"%d["calibration"]["before"]["valu"]["ns_per_wave_inst_per_simd"])
    except Exception as e:
        print(tag, "failed", e)
PY
