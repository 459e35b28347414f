# per-launch accumulate durations of a 16-equal-chunk streamed call for library variants (tools/build_variant.sh)
cd /tmp && export TMPDIR=/tmp
for so in "" e1 e2; do
  rm -rf /tmp/p_v
  [ -n "$so" ] && export MI355ZK_SO=/root/repo/tools/bin/libmi355zk_$so.so || unset MI355ZK_SO
  MI355ZK_HOST_CHUNK_TEST=4194304 timeout 600 rocprofv3 --kernel-trace -d /tmp/p_v -- python /root/repo/tools/trace_host_entry.py > /dev/null 2>&1
  echo "== variant '${so}'"
  python /root/repo/tools/rocpd_summary.py $(find /tmp/p_v -name "*.db" | head -1) --timeline 420 | grep -E "accumulate_kernel" | tail -16 | awk '{printf "%s ", $4} END {print ""}'
done
