#!/usr/bin/env python3
"""power_pairs over 2^24 G1 points: the device-resident call against mi355zk_bn254_g1_merge_pairs on (pageable) host buffers -- pieces of 2^22 points,
two host threads, one upload of the shared array per piece (round 4: 36 ms against 73 ms on one MI355X: the pageable uploads are the difference)."""
import sys, time, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, bench, inputs, phase2_bn254_amd as zk
L=zk.lib.load(); w=zk.Worker(0); dev=torch.device('cuda',0)
n=(1<<24)
k=bench.gen_scalars(n+1,1,dev); d_v=torch.empty((n+1,8),dtype=torch.int64,device=dev)
gen=np.ascontiguousarray(inputs.G1_GEN_RAW)
assert L.mi355zk_bn254_g1_batch_mul_dev(C.c_void_p(d_v.data_ptr()),gen.ctypes.data_as(C.c_void_p),C.c_void_p(k.data_ptr()),n+1,None)==0
d_rho=bench.gen_scalars(n,2,dev)
t=time.perf_counter(); want=zk.ceremony.power_pairs(d_v,d_rho); torch.cuda.synchronize(); t=time.perf_counter()
want=zk.ceremony.power_pairs(d_v,d_rho); dt_dev=time.perf_counter()-t
hv=d_v.cpu().numpy().view(np.uint64); hr=d_rho.cpu().numpy().view(np.uint64)
zk.ceremony.merge_pairs_host(hv[:n],hv[1:],hr)
t=time.perf_counter(); got=zk.ceremony.merge_pairs_host(hv[:n],hv[1:],hr); dt=time.perf_counter()-t
import oracle_lib as O
print("power_pairs 2^24: device-resident %.1f ms; host buffers (pageable, 4 pieces of 2^22, 2 host threads) %.1f ms; same: %s" % (dt_dev*1e3, dt*1e3, np.array_equal(O.G1.to_affine(got[0]),O.G1.to_affine(want[0]))))
