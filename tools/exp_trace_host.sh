cd /root/repo
for cfg in ",," "20,300," "21,240,"; do
IFS=, read f g t <<< "$cfg"
echo "== first=$f growth=$g"
MI355ZK_TRACE_HOST=1 MI355ZK_HOST_CHUNK_FIRST=$f MI355ZK_HOST_CHUNK_GROWTH=$g python tools/trace_host_entry.py 2>&1 | grep "mi355zk" | tail -14
done
