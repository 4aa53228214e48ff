#!/bin/bash
# fixed cost of a small tp_fused launch: the n100 hipGraph timeline with the kernel built without its chunk loop, and
# without loop and epilogue (lab builds; the library is restored afterwards)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/matten_amd/csrc
echo "== full"; bash $R/tools/dbg/n100_timeline.sh 2>&1 | sed -n '/==== graph/,$p' | grep "tp_fused\|kernels "
touch tp_fused.hip; make EXTRA_CXXFLAGS="-DMATTEN_LAB -DMATTEN_ABLATE_NO_LOOP" > /dev/null 2>&1
echo "== no loop (prologue + epilogue only)"; MATTEN_SELFCHECK=0 bash $R/tools/dbg/n100_timeline.sh 2>&1 | sed -n '/==== graph/,$p' | grep "tp_fused\|kernels "
touch tp_fused.hip; make EXTRA_CXXFLAGS="-DMATTEN_LAB -DMATTEN_ABLATE_NO_LOOP -DMATTEN_ABLATE_NO_EPI" > /dev/null 2>&1
echo "== no loop, no epilogue (prologue only)"; MATTEN_SELFCHECK=0 bash $R/tools/dbg/n100_timeline.sh 2>&1 | sed -n '/==== graph/,$p' | grep "tp_fused\|kernels "
touch tp_fused.hip; make > /dev/null 2>&1
