#!/bin/bash
# The l1 >= 2 kinds of tp_fused compiled alone: 3 workgroups per CU (168 VGPRs) with the per-coupling CG code vs
# 2 workgroups (256 VGPRs) with the shared-product code (gen_cg.py MATTEN_CG_SHARED_ROWS: 22-29 % fewer operations,
# needs the registers).  Per-kind times of the last layer.
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
run() {  # $1 = label, $2 = rows, $3 = maxP, $4.. = flags
  label=$1; rows=$2; maxp=$3; shift 3
  MATTEN_CG_SHARED_ROWS=$rows MATTEN_CG_SHARED_MAX_P=$maxp python3 gen_cg.py > cg_gen.h
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I../../include -I. -DMATTEN_LAB -DTPF_ONLY_HEAVY "$@" -c tp_fused.hip -o build/tp_fused.o 2>/dev/null || { echo "build failed"; return; }
  hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o ../libmatten_hip.so
  echo "== $label"; PK=0 python3 ../../tools/fused_kind_bench.py 2>&1 | grep "l1=[234]" | awk '{s+=$(NF-1); print} END {printf "   sum %.3f ms\n", s}'
}
make -j8 > /dev/null 2>&1
run "per-coupling code, 3 blocks/CU" 0 27 -DTPF_MIN_BLOCKS=3
run "per-coupling code, 2 blocks/CU" 0 27 -DTPF_MIN_BLOCKS=2
run "shared products 3 rows (P<=27), 3 blocks/CU" 3 27 -DTPF_MIN_BLOCKS=3
run "shared products 3 rows (P<=27), 2 blocks/CU" 3 27 -DTPF_MIN_BLOCKS=2
run "shared products 5 rows (P<=45), 2 blocks/CU" 5 45 -DTPF_MIN_BLOCKS=2
run "shared products all rows (P<=81), 2 blocks/CU" 9 81 -DTPF_MIN_BLOCKS=2
python3 gen_cg.py > cg_gen.h; touch tp_fused.hip; make -j8 > /dev/null 2>&1
