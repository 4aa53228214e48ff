#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/matten_amd/csrc
echo "== full"; bash $R/tools/dbg/n100_timeline.sh 2>&1 | sed -n '/==== graph/,$p' | grep "tp_fused\|kernels "
touch tp_fused.hip; make EXTRA_CXXFLAGS="-DMATTEN_LAB -DMATTEN_ABLATE_NO_LOOP" > /dev/null 2>&1
echo "== no loop (prologue + epilogue only)"; MATTEN_SELFCHECK=0 bash $R/tools/dbg/n100_timeline.sh 2>&1 | sed -n '/==== graph/,$p' | grep "tp_fused\|kernels "
for L in 8 4; do
touch tp_fused.hip; make > /dev/null 2>&1
echo "== pieces of $L"; MATTEN_HUB_SPLIT_LEN=$L bash $R/tools/dbg/n100_timeline.sh 2>&1 | sed -n '/==== graph/,$p' | grep "tp_fused\|segment_reduce\|kernels "
done
