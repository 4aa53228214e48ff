#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/matten_amd/csrc
for F in "-DMATTEN_ABLATE_NO_LOOP" "-DMATTEN_ABLATE_NO_LOOP -DMATTEN_ABLATE_NO_H2LOAD" "-DMATTEN_ABLATE_NO_LOOP -DMATTEN_ABLATE_NO_XLOAD" "-DMATTEN_ABLATE_NO_LOOP -DMATTEN_ABLATE_NO_GATHER" "-DMATTEN_ABLATE_NO_LOOP -DMATTEN_ABLATE_NO_H2LOAD -DMATTEN_ABLATE_NO_XLOAD"; do
touch tp_fused.hip; make EXTRA_CXXFLAGS="-DMATTEN_LAB $F" > /dev/null 2>&1 || { echo "build failed $F"; continue; }
echo "== $F"; MATTEN_SELFCHECK=0 bash $R/tools/dbg/n100_timeline.sh 2>&1 | sed -n '/==== graph/,$p' | grep "tp_fused" | awk '{printf "%s ", $4} END {print ""}'
done
touch tp_fused.hip; make > /dev/null 2>&1
