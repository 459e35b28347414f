#!/bin/bash
# end-of-round run on the GPU box: the GPU suite, the differential fuzz on a new seed, the pair A/Bs, then the profile refresh
cd /root/repo
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -x -q -m gpu) > gpurun_out/pytest_gpu.txt 2>&1; tail -3 gpurun_out/pytest_gpu.txt
CASES=30 SEED=41 bash tools/fuzz_msm.sh > gpurun_out/fuzz_msm_seed41.txt 2>&1; grep -c "30/30 ok" gpurun_out/fuzz_msm_seed41.txt; grep -v "30/30 ok" gpurun_out/fuzz_msm_seed41.txt | head
bash tools/ab_g1_pair.sh > gpurun_out/ab_g1_pair.txt 2>&1
bash tools/ab_g2_pair.sh > gpurun_out/ab_g2_pair.txt 2>&1
bash tools/ab_g2_waves.sh > gpurun_out/ab_g2_waves.txt 2>&1
bash tools/ab_g2_waves_prover.sh > gpurun_out/ab_g2_waves_prover.txt 2>&1
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
tail -5 gpurun_out/refresh.log
cat gpurun_out/refresh/bench_n1.json | cut -c1-600
