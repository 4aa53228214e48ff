#!/bin/bash
# round 4, first GPU call: full GPU suite, bench line (calibration + VALU roofline), 2-rank rehearsal, PMC of the current kernels,
# and the self-check against a library built WITHOUT the species-linear correctness flag
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4a_pytest.log
tail -5 gpurun_out/r4a_pytest.log
timeout 600 python3 bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/r4a_bench.json
timeout 600 python3 bench.py --gpus 2 --backend gloo --share-gpu --no-extras --steps 10 --warmup 3 > gpurun_out/r4a_bench_2rank_rehearsal.json 2> gpurun_out/r4a_bench_2rank_rehearsal.err; echo "rehearsal rc=$?"
cut -c1-600 gpurun_out/r4a_bench_2rank_rehearsal.json; tail -3 gpurun_out/r4a_bench_2rank_rehearsal.err
bash tools/collect_valu.sh r04base
bash tools/collect_traffic.sh r04base
# a build without the correctness flag: does the self-check catch it?
(cd matten_amd/csrc && make SL_FLAGS= -B build/species_linear.o build/species_linear_rows.o > /dev/null 2>&1 && make > /dev/null 2>&1)
timeout 300 python3 - > gpurun_out/r4a_selfcheck_noflag.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
from matten_amd import selfcheck
try:
    selfcheck.check_species_linear("cuda:0")
    print("SL_FLAGS= build: self-check PASSED (this toolchain does not miscompile the current source without the flag)")
except Exception as e:
    print("SL_FLAGS= build: self-check REFUSED the library:", str(e)[:600])
PY
cat gpurun_out/r4a_selfcheck_noflag.log
(cd matten_amd/csrc && make -B build/species_linear.o build/species_linear_rows.o > /dev/null 2>&1 && make > /dev/null 2>&1)
