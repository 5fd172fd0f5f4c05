#!/bin/bash
# The round's EVIDENCE PASS on the GPU box: `gpurun --timeout 1500 -- 'ROUND=r06 bash scripts/gpu_round.sh'`, then, back in the container,
# `python scripts/collect_profiles.py r06` condenses gpurun_out/r06/ into the tracked profiles/r06_*.
# Order matters: every un-profiled number first, profiler passes last (a PMC session can leave the clocks in another state for a while).
set -u
R=${ROUND:-r06}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
# 1. the box: CPUs, quota, NUMA, where hipHostMalloc pages land, write(2) rates of 1..8 threads (the host side of bs_render_png_files)
{ nproc; cat /sys/fs/cgroup/cpu.max; rocminfo | grep -E "Marketing|gfx9" | head -4; } > $O/box.txt 2>&1
hipcc -O2 scripts/numa_probe.cpp -o /tmp/numa_probe -lpthread 2>/dev/null && { /tmp/numa_probe /dev/shm; /tmp/numa_probe /tmp | grep writer; } > $O/host_topology.txt 2>&1
# 2. does it run, is it right
(python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.txt 2>&1
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -4) > $O/pytest_gpu_tail.txt
# 3. the driver's own command, and the other single-GPU workloads / modes
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --mode strict --cpu-seconds 0 --traffic static --form resident --sustained-frames 0 > $O/bench_strict.json 2>> $O/bench_default.err
python bench.py --workload default --cpu-seconds 0 --traffic static --sustained-frames 0 > $O/bench_c2_default.json 2>> $O/bench_default.err
python bench.py --workload lensing-4k --cpu-seconds 0 --traffic static --sustained-frames 0 > $O/bench_c4_lensing_4k.json 2>> $O/bench_default.err
python bench.py --workload animation --cpu-seconds 0 --traffic static --sustained-frames 0 > $O/bench_c5_animation.json 2>> $O/bench_default.err
# 4. what the driver will run on the 8-GPU node, rehearsed on this one device (oversubscribed: smoke, not a result), and RCCL at world 1
python3 bench.py --gpus 8 --steps 20 --warmup 5 > $O/bench_n8_driver_form.json 2> $O/bench_n8.err
python3 bench.py --gpus 8 --steps 20 --warmup 5 --launcher torchrun > $O/bench_n8_torchrun_gloo_smoke.json 2>> $O/bench_n8.err
python3 bench.py --gpus 1 --launcher torchrun --cpu-seconds 0 --traffic static > $O/bench_rccl_world1.json 2>> $O/bench_n8.err
python scripts/png_files_probe.py 20 60 200 > $O/png_files_vs_png_batch.jsonl 2>> $O/bench_default.err
# 5. rocprofv3 kernel trace + stats of THE SAME COMMAND as the driver's (--traffic static: no profiler inside the profiler; the CPU leg is
#    not a GPU matter).  collect_profiles.py takes the timed launches (after the warm-up ones) out of the trace and sets them beside the
#    kernel_ms of the line this very run printed.
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fast -o fast -- python bench.py --traffic static --cpu-seconds 0 > $O/bench_profiled_fast.json 2> $O/prof_fast.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_strict -o strict -- python bench.py --mode strict --traffic static --cpu-seconds 0 --form resident > $O/bench_profiled_strict.json 2> $O/prof_strict.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rgb8 -o rgb8 -- python scripts/prof_rgb8.py > $O/prof_rgb8.txt 2>&1
# 6. PMC passes, one counter group per run (TCC: FETCH_SIZE and WRITE_SIZE never together; GRBM rides with WRITE_SIZE)
for m in fast strict; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq_$m -o sq -- python scripts/prof_frame.py --mode $m --frames 3 > $O/pmc_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2_$m -o sq2 -- python scripts/prof_frame.py --mode $m --frames 3 >> $O/pmc_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$m -o fetch -- python scripts/prof_frame.py --mode $m --frames 3 >> $O/pmc_$m.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_write_$m -o write -- python scripts/prof_frame.py --mode $m --frames 3 >> $O/pmc_$m.log 2>&1
done
for g in fetch write; do
  c=FETCH_SIZE; [ $g = write ] && c=WRITE_SIZE
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${g}_rgb8 -o $g -- python scripts/prof_rgb8.py >> $O/pmc_rgb8.log 2>&1
done
cat $O/smoke.txt | tail -1; cat $O/pytest_gpu_tail.txt; python -c "
import json,sys
for f in ('bench_default','bench_profiled_fast','bench_n8_driver_form'):
    try:
        d=json.load(open('$O/%s.json'%f)); print(f, d['value'], d.get('valid'), d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('frac_cycles'), d.get('leg_seconds',{}).get('total_so_far'))
    except Exception as e: print(f, 'unreadable', e)
"
