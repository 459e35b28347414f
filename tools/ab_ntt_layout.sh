#!/bin/bash
# same-box A/B of the NTT LDS layout: tools/bin/libmi355zk_prev.so (round 4: pitch np + 1, xor of bits 5..9) against the product library
# (round 5: pitch np, multiplier 25, row term), sizes 2^16 .. 2^26, two alternating rounds; then the 2^20 pass's SQ / HBM counters on the product
cd "$(dirname "$0")/.."
R=$PWD
PREV=$R/tools/bin/libmi355zk_prev.so
for round in 1 2; do
  for ln in 16 18 20 22 24 26; do
    for so in "$PREV" ""; do
      tag=$([ -z "$so" ] && echo "round5" || echo "round4")
      echo "$tag 2^$ln $(MI355ZK_SO=$so python tools/bench_ntt.py --log-n $ln --iters 30 --check 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k:((v["ms"], v["ntt_pass_ms_avg"]) if isinstance(v,dict) else v) for k,v in d.items() if k in ("fft","ifft","fft_matches_oracle")})')"
    done
  done
done
cd /tmp && export TMPDIR=/tmp
for so in "$PREV" ""; do
  tag=$([ -z "$so" ] && echo "round5" || echo "round4")
  rm -rf /tmp/p_ns1 /tmp/p_ns2 /tmp/p_nf /tmp/p_nw
  MI355ZK_SO=$so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAIT_ANY -d /tmp/p_ns1 -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
  MI355ZK_SO=$so timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d /tmp/p_ns2 -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
  echo "== $tag: ntt_pass_kernel, 2^20, per dispatch"
  python $R/tools/pmc_kernel.py $(find /tmp/p_ns1 -name "*.db" | head -1) ntt_pass_kernel
  python $R/tools/pmc_kernel.py $(find /tmp/p_ns2 -name "*.db" | head -1) ntt_pass_kernel
  if [ -z "$so" ]; then
    timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/p_nf -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/p_nw -- python $R/tools/bench_ntt.py --log-n 20 > /dev/null 2>&1
    echo "== $tag: FETCH_SIZE / WRITE_SIZE (KiB per dispatch, raw counters)"
    python $R/tools/rocpd_summary.py $(find /tmp/p_nf -name "*.db" | head -1) --pmc
    python $R/tools/rocpd_summary.py $(find /tmp/p_nw -name "*.db" | head -1) --pmc
  fi
done
