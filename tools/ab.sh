#!/bin/bash
# dev tool, runs on the GPU box: A/B of engine builds on generation jobs, interleaved passes on ONE box (the only comparison that means anything:
# boxes differ by 1-2 %).
#   tools/ab.sh <tag> "<builds>" "<jobs>" [samples=3000] [passes=2] [check]
#     builds: names of tools/variants/libwn_<name>.so (tools/build_variant.py), "product" (the in-tree library), or "tree:<dir>" -- a git-archive
#             of another revision with its own library and tools/rate.py (e.g. tools/variants/r04tree)
#     jobs:   "cfg streams" pairs separated by ';'            e.g. "cfg3 64"   "cfg2 1;cfg1 1;chaconne 1"
#     check:  first run tools/quick_check.py (logits vs the C oracle) of every build on every job
# Output: gpurun_out/ab_<tag>.txt
TAG=$1; LIST=$2; JOBS=${3:-cfg3 64}; N=${4:-3000}; PASSES=${5:-2}; CHECK=${6:-}
mkdir -p gpurun_out; cd /root/repo
lib_of() { case $1 in product) echo pytorch-wavenet_amd/mi355_wavenet/libwn_mi355.so ;; *) echo tools/variants/libwn_$1.so ;; esac; }
run() {  # build, tool, job...
  local v=$1 tool=$2; shift 2
  case $v in
    tree:*) ( cd ${v#tree:} && timeout 300 python tools/$tool "$@" 2>&1 ) ;;
    *) WN_DEV_LIB=$(lib_of $v) timeout 300 python tools/$tool "$@" 2>&1 ;;
  esac
}
{
  IFS=';' read -ra JOBLIST <<< "$JOBS"
  if [ -n "$CHECK" ]; then
    for v in $LIST; do for j in "${JOBLIST[@]}"; do run $v quick_check.py $j | grep quick_check | sed "s|^|$v: |"; done; done
  fi
  for pass in $(seq 1 $PASSES); do
    for v in $LIST; do for j in "${JOBLIST[@]}"; do run $v rate.py $j $N 3 | grep "samples/s" | sed "s|^|$v (pass $pass): |"; done; done
  done
} 2>&1 | tee gpurun_out/ab_$TAG.txt
