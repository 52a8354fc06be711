#!/bin/bash
# The measurements behind profiles/rNN_*: run as ONE gpurun call from the repo root on the GPU box,
#   gpurun --timeout 1200 -- 'bash tools/final_measure.sh r03_v1'
# then copy in the container:
#   python tools/summarize_rocprof.py gpurun_out/<tag>/prof profiles/<tag>          (kernel stats + launch table)
#   cp gpurun_out/<tag>/<round>_pmc_blind_rotate.json profiles/ ; cp gpurun_out/<tag>/bench.json profiles/<tag>_bench.json
# Order: counter passes first (separate rocprofv3 runs with --kernel-trace only, MI355X_MICROARCH.md HBM section; the program
# after `--` is python3 itself, no wrapper that would re-exec a GPU-initialised process), their summary is written next to the
# sources it was taken from (profiles/<round>_pmc_blind_rotate.json, stamped with the sha256 of the engine sources), THEN the
# un-profiled bench.py, whose roofline object picks traffic / valu_busy / ceiling_frac up from that file, then the kernel trace.
set -e -o pipefail
TAG=${1:-final}
ROUND=${TAG%%_*}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
# the ISA-count x issue-cost model of the blind-rotation kernel (tools/k2_dyncount.py) for roofline.ceiling_frac
# (the product's engine unit: its flags come from tfhe_aes_amd/_build.py -- post-RA scheduler off, key-switching kernels in the other unit)
hipcc $(python3 -c 'from tfhe_aes_amd import _build; print(" ".join(_build.unit_flags("engine")))') -S --cuda-device-only \
      -o /tmp/engine_final.s $R/tfhe_aes_amd/csrc/engine.hip 2> /dev/null
python3 $R/tools/k2_dyncount.py /tmp/engine_final.s blind_rotate_pair_kernelILi5ELi5ELi8ELi3ELi2 1 0 | tee $O/k2_dyncount.txt
# build (if stale) BEFORE anything runs under the profiler: hipcc must not be spawned from a process tree rocprofv3 has GPU-initialised
python3 -c 'from tfhe_aes_amd import _build; _build.build_all()'
cd /tmp && export TMPDIR=/tmp
pmc() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$name -- python3 $R/tools/run_k2.py 16384 1 > $O/pmc_$name.log 2>&1; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum
pmc act SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM
cd $R
# 16,384 bits on 2,816 workgroups (2,560 of six ciphertexts, 256 of four): 5.8182 ciphertexts per workgroup on average
python3 tools/summarize_pmc.py $O/pmc "blind_rotate_pair_kernel<5, 5" profiles/${ROUND}_pmc_blind_rotate 16384 --cus 256 --dyncount $O/k2_dyncount.txt \
        --flops-per-ct-iteration 609280 --cts-per-wg 5.8182 --waves-per-wg 8 --algorithmic-bytes 698912768 > $O/pmc_summary.txt
cp profiles/${ROUND}_pmc_blind_rotate.json $O/
# the second kernel of a step (K3, packing key switch): its own counter passes and summary (bench.py attaches them to roofline_k3)
cd /tmp
pmc3() { name=$1; shift; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_k3/$name -- python3 $R/tools/run_k3.py 16384 3 > $O/pmc_k3_$name.log 2>&1; }
pmc3 fetch FETCH_SIZE
pmc3 write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc3 sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pmc3 grbm GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum
pmc3 mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA
cd $R
python3 tools/summarize_pmc_k3.py $O/pmc_k3 profiles/${ROUND}_pmc_pfpks 16384 > $O/pmc_k3_summary.txt
cp profiles/${ROUND}_pmc_pfpks.json $O/
cd /tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
# the driver's exact command (every block of every step verified; exit code 1 on a wrong one)
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err && echo "driver-style bench: all verified" || echo "driver-style bench FAILED (see bench_driver_style.err)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/prof.log 2>&1
echo done
