#!/bin/bash
# Counter passes over ONE 16,384-bit blind-rotation launch (tools/run_k2.py), separate rocprofv3 runs with --kernel-trace only
# (MI355X_MICROARCH.md, HBM section).  K2_SO=<path of a developer build> profiles that library instead of the product.  usage (one gpurun call):  bash tools/pmc_k2.sh <tag> [extra passes: name=CTR1,CTR2 ...]
#   python tools/summarize_pmc.py gpurun_out/<tag>/pmc "blind_rotate16_kernel<5, 5" profiles/<name> 16384
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$name -- python3 $R/tools/run_k2.py 16384 1 $K2_SO > $O/pmc_$name.log 2>&1 || echo "pass $name failed"; tail -1 $O/pmc_$name.log; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum
pmc act SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM
for p in "$@"; do n=${p%%=*}; c=${p#*=}; pmc $n ${c//,/ }; done
echo done
