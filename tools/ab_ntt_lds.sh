# same-box A/B of NTT variants (round 4): usage on the GPU box: bash tools/ab_ntt_lds.sh
# The limb-PLANE LDS tiles of rounds 1-3 against the element-major ones were measured with a second build of the library:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DZK_NTT_LDS_PLANES -c phase2-bn254_amd/csrc/ntt.hip -o build/ntt_planes.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o phase2-bn254_amd/libmi355zk_planes.so build/ntt_planes.o build/{msm_g1,msm_g2,api,field_ops,point_fft,point_fft_g2,codec}.o
#   MI355ZK_SO=$PWD/phase2-bn254_amd/libmi355zk_planes.so python tools/bench_ntt.py --log-n 20      (against the same command without MI355ZK_SO)
run() { python tools/bench_ntt.py --log-n $1 --iters 30 --warm 60 | python -c "import sys,json; d=json.load(sys.stdin); print(d['log_n'], {k:(v['ms'],v['ntt_pass_ms_avg'],v['passes']) for k,v in d.items() if isinstance(v,dict)})"; }
for rep in 1 2; do
  echo "default"; for ln in 20 23 24 25; do run $ln; done
  echo "LOGNP=12 TILE=4096"; for ln in 23 24; do MI355ZK_NTT_LOGNP=12 MI355ZK_NTT_TILE=4096 run $ln; done
  echo "LOGNP=12 (tile default)"; for ln in 24; do MI355ZK_NTT_LOGNP=12 run $ln; done
  echo "TILE=4096"; for ln in 20 24; do MI355ZK_NTT_TILE=4096 run $ln; done
done
