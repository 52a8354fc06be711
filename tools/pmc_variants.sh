#!/bin/bash
# HBM-side traffic (FETCH_SIZE; WRITE_SIZE + L2 hit/miss) of one 16,384-bit launch for several developer builds of the engine
# (variant names of tools/ablate_k2.py).  usage (one gpurun call): bash tools/pmc_variants.sh <tag> v1,v2,...
TAG=$1; VARS=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 tools/ablate_k2.py 16384 $VARS > $O/abl.log 2>&1; grep "M=" $O/abl.log
cd /tmp && export TMPDIR=/tmp
for v in ${VARS//,/ }; do
  so=$R/gpurun_out/abl/libfheaes_$v.so
  if [ "$v" = "parkwg" ]; then export K2_PARKING=private; else unset K2_PARKING; fi
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "grbm GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum"; do
    set -- $pass; name=$1; shift
    timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$v/$name -- python3 $R/tools/run_k2.py 16384 1 $so > $O/${v}_$name.log 2>&1 || echo "pass $v $name failed"
  done
  echo "$v done"
done
