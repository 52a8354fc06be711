#!/bin/bash
# Counter passes over 16,384-bit packing key switch launches (tools/run_k3.py), separate rocprofv3 runs with --kernel-trace only
# (MI355X_MICROARCH.md, HBM section).  usage (one gpurun call):  bash tools/pmc_k3.sh <tag>
#   python tools/summarize_pmc_k3.py gpurun_out/<tag>/pmc profiles/<round>_pmc_pfpks 16384
TAG=${1:-pmc_k3}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 -c 'from tfhe_aes_amd import _build; _build.build_all()'
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$name -- python3 $R/tools/run_k3.py 16384 3 $K3_SO > $O/pmc_$name.log 2>&1 || echo "pass $name failed"; tail -1 $O/pmc_$name.log; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA
for p in "$@"; do n=${p%%=*}; c=${p#*=}; pmc $n ${c//,/ }; done
echo done
