cd /root/repo
export ASAN_OPTIONS=log_path=/root/repo/gpurun_out/asanlog UBSAN_OPTIONS=log_path=/root/repo/gpurun_out/ubsanlog
echo "--- step 1: torch cuda under preload"
tools/run_asan.sh python -c "import torch; print(torch.cuda.is_available()); x=torch.ones(4,device='cuda'); print(x.sum().item())" 2>&1 | tail -5; echo "rc=${PIPESTATUS[0]}"
echo "--- step 2: load lib + a tiny msm"
tools/run_asan.sh python - <<'PY' 2>&1 | tail -20
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import phase2_bn254_amd as zk, inputs, oracle_lib as O
w = zk.Worker(0)
print("worker ok", flush=True)
b = inputs.bases_progression_cpu(1, 500, seed=1); s = inputs.random_scalars(500, seed=2)
got = zk.multiexp(w, (b, 0), zk.FullDensity(), s).wait()
rc, want = O.G1.multiexp(b, s)
print("msm ok", np.array_equal(O.G1.to_affine(got), O.G1.to_affine(want)), flush=True)
PY
echo "rc=${PIPESTATUS[0]}"
ls gpurun_out/ | grep -i "san" ; for f in gpurun_out/asanlog* gpurun_out/ubsanlog*; do [ -f $f ] && (echo "== $f"; head -40 $f); done
