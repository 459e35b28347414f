#!/bin/bash
# Runs on the GPU box (via gpurun): collects everything profiles/ is built from into gpurun_out/refresh/.
#   tools/refresh_profiles.sh            then locally: python tools/refresh_profiles_post.py
set -u
R=/root/repo
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
timeout 400 $B > $O/bench_n1.json 2> $O/bench_n1.err
timeout 200 $B --log-n 20 --steps 20 --warmup 3 --no-secondary > $O/bench_2e20.json 2>/dev/null
timeout 400 rocprofv3 --kernel-trace -d /tmp/p_kt -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_kt -name "*.db" | head -1) > $O/msm26_kernel_stats.txt
timeout 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/pf.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/pw.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_f -name "*.db" | head -1) --pmc > $O/msm26_pmc_fetch.txt
python $R/tools/rocpd_summary.py $(find /tmp/p_w -name "*.db" | head -1) --pmc > $O/msm26_pmc_write.txt
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY -d /tmp/p_s -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-h2d-leg --no-secondary > $O/ps.log 2>&1
python $R/tools/pmc_kernel.py $(find /tmp/p_s -name "*.db" | head -1) msm_accumulate_kernel > $O/msm26_accumulate_sq_pmc.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_ntt -- python $R/tools/bench_ntt.py --check > $O/ntt20.json 2> $O/ntt.err
python $R/tools/rocpd_summary.py $(find /tmp/p_ntt -name "*.db" | head -1) > $O/ntt20_kernel_stats.txt
rm -rf /tmp/p_t20; timeout 300 rocprofv3 --kernel-trace -d /tmp/p_t20 -- python $R/tools/trace_one_msm.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_t20 -name "*.db" | head -1) --timeline 27 > $O/msm20_timeline.txt
timeout 300 python $R/tools/bench_g2.py > $O/g2_2e20.json 2>/dev/null
timeout 400 python $R/tools/bench_next_rows.py --log-n 20 > $O/next_rows_2e20.json 2>/dev/null
timeout 300 python $R/tools/bench_contribute.py > $O/contribute_2e20.json 2>/dev/null
timeout 400 python $R/tools/bench_host_entry.py --log-n 20 24 26 > $O/host_entry.json 2>/dev/null
timeout 400 python $R/tools/bench_shard_cells.py > $O/shard_cells_2e26.json 2>/dev/null
for ln in 16 20 24; do timeout 300 python $R/tools/bench_ntt.py --log-n $ln; done > $O/ntt_16_20_24.json 2>/dev/null
for ln in 12 14 18 19 21 22 23 25 26 28; do timeout 300 python $R/tools/bench_ntt.py --log-n $ln; done > $O/ntt_other_sizes.json 2>/dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY -d /tmp/p_ns1 -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d /tmp/p_ns2 -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
{ python $R/tools/pmc_kernel.py $(find /tmp/p_ns1 -name "*.db" | head -1) ntt_pass_; python $R/tools/pmc_kernel.py $(find /tmp/p_ns2 -name "*.db" | head -1) ntt_pass_; } > $O/ntt20_pass_sq_pmc.txt 2>&1
rm -rf /tmp/p_nf /tmp/p_nw
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_nf -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_nw -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_nf -name "*.db" | head -1) --pmc > $O/ntt20_pmc_fetch.txt
python $R/tools/rocpd_summary.py $(find /tmp/p_nw -name "*.db" | head -1) --pmc > $O/ntt20_pmc_write.txt
# round 5: per-pass durations of the four operations, transforms in batches, transforms on separate streams
LOGN=20 bash $R/tools/ntt_pass_split.sh > /dev/null 2>&1; cp $R/gpurun_out/ntt_pass_split.txt $O/ntt_pass_split.txt
for ln in 16 18 20; do timeout 300 python $R/tools/bench_ntt_batch.py --log-n $ln; done > $O/ntt_batch.json 2>/dev/null
{ timeout 300 python $R/tools/exp_ntt_streams.py --k 2; timeout 300 python $R/tools/exp_ntt_streams.py --k 3; } > $O/ntt_streams.json 2>/dev/null
{ for ln in 20 22 24; do MI355ZK_NTT_NO_FOLD=1 timeout 300 python $R/tools/bench_ntt.py --log-n $ln; done; } > $O/ntt_no_fold.json 2>/dev/null
timeout 400 python $R/tools/bench_skew.py --log-n 26 --iters 2 > $O/skew_2e26.json 2>/dev/null
timeout 400 $B --steps 5 --warmup 1 --bases tau --no-cpu-baseline --no-secondary > $O/bench_n1_tau.json 2>/dev/null
$R/tools/bin/ubench_valu > $O/ubench_valu.txt 2>&1
$R/tools/bin/ubench_gather > $O/ubench_gather.txt 2>&1
$R/tools/bin/ubench_fieldmul > $O/ubench_fieldmul.txt 2>&1
$R/tools/bin/ubench_wave_bucket > $O/ubench_wave_bucket.txt 2>&1
$R/tools/bin/ubench_gather_footprint > $O/ubench_gather_footprint.txt 2>&1
for lm in 16 20 22; do timeout 300 python $R/tools/bench_prover.py --log-m $lm; done > $O/prover.json 2>/dev/null
{ for ln in 19 20 22 23; do timeout 300 python $R/tools/bench_table.py --log-n $ln --iters 10; done; for ln in 16 20 22; do timeout 300 python $R/tools/bench_table.py --group 2 --log-n $ln --iters 8; done; } > $O/table_mode.json 2>/dev/null
for ln in 16 20; do timeout 200 python $R/tools/bench_skew.py --log-n $ln --iters 10; done > $O/skew_small.json 2>/dev/null
rm -rf /tmp/p_g; timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_g -- $R/tools/bin/ubench_gather > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_g -name "*.db" | head -1) --pmc > $O/ubench_gather_fetch_size.txt
# round 4
cd $R
bash tools/ab_quad.sh > $O/ab_quad.txt 2>/dev/null
timeout 600 python tools/bench_multi_device.py --log-n 26 --devices 1 2 4 8 > $O/multi_device_2e26.json 2>/dev/null
bash tools/ab_ntt_lds.sh > $O/ntt_configs.txt 2>/dev/null
[ -f phase2-bn254_amd/libmi355zk_mont.so ] && bash tools/ab_ntt_shoup.sh > $O/ab_ntt_shoup.txt 2>/dev/null
cd /tmp; rm -rf /tmp/p_he; timeout 600 rocprofv3 --kernel-trace -d /tmp/p_he -- python $R/tools/trace_host_entry.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_he -name "*.db" | head -1) --timeline 140 > $O/host_entry_timeline.txt
rm -rf /tmp/p_t16; TRACE_LOG_N=16 timeout 300 rocprofv3 --kernel-trace -d /tmp/p_t16 -- python $R/tools/trace_one_msm.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/p_t16 -name "*.db" | head -1) --timeline 26 > $O/msm16_timeline.txt
ls -la $O
