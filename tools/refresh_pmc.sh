#!/bin/bash
# The headline kernel's evidence only (kernel trace + the three PMC passes of tools/refresh_profiles.sh), into gpurun_out/refresh/: run it after a change to
# the MSM sources that leaves everything else in profiles/ valid, then `python tools/refresh_profiles_post.py` re-locks profiles/latest_pmc.json to the sources.
set -u
R=/root/repo
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 400 $B > $O/bench_n1.json 2> $O/bench_n1.err
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_s
timeout 400 rocprofv3 --kernel-trace -d /tmp/p_kt -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) > $O/msm26_kernel_stats.txt
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/pf.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/pw.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) --pmc > $O/msm26_pmc_fetch.txt
python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) --pmc > $O/msm26_pmc_write.txt
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY -d /tmp/p_s -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/ps.log 2>&1
python $R/tools/pmc_kernel.py $(find /tmp/p_s -name "*.db" | head -1) msm_accumulate_kernel > $O/msm26_accumulate_sq_pmc.txt
ls -la $O | head -5
