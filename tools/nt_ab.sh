#!/bin/bash
# non-temporal hints on the agg stream: tp_fused stores (TPF_NT_STORES) / agg_linear loads (AL_NT_LOADS); whole forward
cd "$GRAFT_REPO_ROOT/matten_amd/csrc" || exit 1
run() { python3 ../../bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_launch']; print('$1', round(d['ms_per_step'],3), {a.split('/')[0]+'/'+a.split('=')[1].split('/')[0]:round(v,3) for a,v in k.items() if 'tp_scatter' in a or 'agg_linear' in a})"; }
for fl in "" "-DTPF_NT_STORES" "-DAL_NT_LOADS" "-DTPF_NT_STORES -DAL_NT_LOADS" ""; do
  touch tp_fused.hip agg_linear.hip; make -j8 EXTRA_CXXFLAGS="$fl" > /dev/null 2>&1 || { echo "build failed $fl"; continue; }
  run "[$fl]"
done
touch tp_fused.hip agg_linear.hip; make -j8 > /dev/null 2>&1
