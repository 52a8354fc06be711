#!/bin/bash
# The measurements behind profiles/rNN_*: run as ONE gpurun call from the repo root on the GPU box,
#   gpurun --timeout 1200 -- 'bash tools/final_measure.sh r02_v2'
# then summarise in the container:
#   python tools/summarize_rocprof.py gpurun_out/<tag>/prof profiles/<tag>          (kernel stats + launch table)
#   python tools/summarize_pmc.py gpurun_out/<tag>/pmc "blind_rotate16_kernel<5, 5" profiles/<round>_pmc_blind_rotate 16384
# Counter passes are separate rocprofv3 runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section); the
# program after `--` is python3 itself (no wrapper that would re-exec a GPU-initialised process).
set -e -o pipefail
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-400
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof.log 2>&1
pmc() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$name -- python3 $R/tools/run_k2.py 16384 1 > $O/pmc_$name.log 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum
echo done
