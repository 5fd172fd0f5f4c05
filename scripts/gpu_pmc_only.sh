#!/bin/bash
# Just the PMC passes of gpu_round.sh (separate --pmc runs; one launch per frame through prof_frame.py).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in fast strict; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_sq_$m -o sq -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_sq_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_sq2_$m -o sq2 -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_sq2_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_grbm_$m -o grbm -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_grbm_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$m -o fetch -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_fetch_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$m -o write -- python scripts/prof_frame.py --mode $m --frames 3 > gpurun_out/pmc_write_$m.log 2>&1
done
tail -n 1 gpurun_out/pmc_sq_fast.log | cut -c1-200
