#!/bin/bash
# species_linear_wgrad_kernel: where the time goes (variants built on the GPU box, library restored afterwards)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/matten_amd/csrc
run() {  # label, extra flags
    touch backward.hip
    make EXTRA_CXXFLAGS="$2" > /dev/null 2>&1 || { echo "build failed: $1"; return; }
    echo "== $1"
    python3 $R/tools/wgrad_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids"
}
run "HEAD" ""
run "one workgroup per species" "-DMATTEN_LAB -DMATTEN_WG_FORCE_SLICES=1"
run "slices of 64 rows" "-DMATTEN_WG_SLICE_ROWS=64"
run "slices of 256 rows" "-DMATTEN_WG_SLICE_ROWS=256"
run "no MFMA" "-DMATTEN_LAB -DMATTEN_WG_NO_MFMA"
run "no loads" "-DMATTEN_LAB -DMATTEN_WG_NO_LOAD"
run "tile block 2" "-DMATTEN_WG_TB=2"
touch backward.hip; make > /dev/null 2>&1
