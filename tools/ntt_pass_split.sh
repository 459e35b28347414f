#!/bin/bash
# per-pass durations of the 2^20 transforms (one op per traced run)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
{
for op in fft ifft coset_fft icoset_fft; do
  rm -rf /tmp/p_tr
  timeout 300 rocprofv3 --kernel-trace -d /tmp/p_tr -- python $R/tools/bench_ntt.py --log-n ${LOGN:-20} --ops $op > /tmp/bn.json 2>/dev/null
  echo "== $op  $(cat /tmp/bn.json | cut -c1-160)"
  python $R/tools/ntt_pass_split.py $(find /tmp/p_tr -name "*.db" | head -1) ${PASSES:-2}
done
} > $O/ntt_pass_split.txt 2>&1
cat $O/ntt_pass_split.txt
