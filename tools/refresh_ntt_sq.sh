cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/refresh
rm -rf /tmp/p_ns1 /tmp/p_ns2
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY -d /tmp/p_ns1 -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d /tmp/p_ns2 -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
{ python $R/tools/pmc_kernel.py $(find /tmp/p_ns1 -name "*.db" | head -1) ntt_pass_; python $R/tools/pmc_kernel.py $(find /tmp/p_ns2 -name "*.db" | head -1) ntt_pass_; } > $O/ntt20_pass_sq_pmc.txt 2>&1
cat $O/ntt20_pass_sq_pmc.txt
