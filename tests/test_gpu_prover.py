"""The prover-shaped caller (phase2-bn254_amd/prover.py = bellman/src/groth16/prover.rs:202-343) end to end against the
oracle: a synthetic 2^16-constraint instance -- random evaluation vectors a, b, c, witness-like assignments full of 0 and 1,
three DensityTrackers, parameter vectors k_i*G / k_i*G2 -- goes through 3 x (ifft, coset_fft), mul / sub / divide_by_z,
icoset_fft, the fused Montgomery -> repr conversion and the eight multiexps (submitted concurrently before the first wait),
and the proof (A, B, C) must equal, byte for byte, the one assembled from the oracle's domain ops, multiexps and group law."""
import ctypes as C

import numpy as np
import pytest

import bn254_model as M
import golden_util as GU
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu
R = M.R_ORDER


def _ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(x[0]) | int(x[1]) << 64 | int(x[2]) << 128 | int(x[3]) << 192 for x in a]


def _limbs(vals):
    return np.array([M.to_limbs(v % R) for v in vals], dtype=np.uint64)


def _mont(vals):
    return _limbs([v * M.MONT_R % R for v in vals])


@pytest.mark.parametrize("log_m,concurrent", [(10, False), (16, True)])
def test_create_proof_matches_the_oracle(zk, worker, log_m, concurrent):
    import torch

    P = zk.prover
    dev = torch.device("cuda", 0)
    L = zk.lib.load()
    m = 1 << log_m
    num_inputs, num_aux = 10, m - 37
    rng = np.random.default_rng(3000 + log_m)

    def synth(group, n, seed):
        import bench

        k = bench.gen_scalars(n, seed, dev)
        p = torch.empty((n, 8 * group), dtype=torch.int64, device=dev)
        gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
        fn = L.mi355zk_bn254_g1_batch_mul_dev if group == 1 else L.mi355zk_bn254_g2_batch_mul_dev
        assert fn(C.c_void_p(p.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
        return p

    # assignments: canonical values with the 0 / 1 mix of a real witness, held in Montgomery form like the prover's Vec<Fr>
    def witness(n, seed):
        v = _ints(inputs.random_scalars(n, seed=seed))
        kind = rng.integers(0, 10, size=n)
        return [0 if k < 2 else 1 if k < 5 else x for k, x in zip(kind, v)]

    inp_c, aux_c = witness(num_inputs, 3101), witness(num_aux, 3102)
    inp_c[0] = 1                                                        # the constant ONE input
    a_aux_bits = rng.random(num_aux) < 0.5
    b_in_bits = rng.random(num_inputs) < 0.5
    b_aux_bits = rng.random(num_aux) < 0.4
    abc = [_ints(inputs.random_fr_mont(m, seed=3110 + i)) for i in range(3)]     # Montgomery values of the evaluation vectors
    na, nb_ = int(a_aux_bits.sum()), int(b_in_bits.sum()) + int(b_aux_bits.sum())
    h_b, l_b = synth(1, m - 1, 3120), synth(1, num_aux, 3121)
    a_b, b1_b, b2_b = synth(1, num_inputs + na, 3122), synth(1, nb_, 3123), synth(2, nb_, 3124)
    vk_pts1 = O.G1.mul_many_affine(inputs.G1_GEN_RAW, inputs.random_scalars(3, seed=3130))
    vk_pts2 = O.G2.mul_many_affine(inputs.G2_GEN_RAW, inputs.random_scalars(2, seed=3131))
    vk = {"alpha_g1": vk_pts1[0], "beta_g1": vk_pts1[1], "delta_g1": vk_pts1[2], "beta_g2": vk_pts2[0], "delta_g2": vk_pts2[1]}
    r, s = 0x1234567890ABCDEF1122334455667788 % R, 0x0FEDCBA9876543210F1E2D3C4B5A6978 % R

    to_dev = lambda arr: torch.from_numpy(np.ascontiguousarray(arr).view(np.int64)).to(dev)  # noqa: E731
    params = P.Parameters(vk, h_b, l_b, a_b, b1_b, b2_b)
    assignment = P.ProvingAssignment(to_dev(_limbs(abc[0])), to_dev(_limbs(abc[1])), to_dev(_limbs(abc[2])), to_dev(_mont(inp_c)),
                                     to_dev(_mont(aux_c)), zk.DensityTracker.from_bools(a_aux_bits), zk.DensityTracker.from_bools(b_in_bits),
                                     zk.DensityTracker.from_bools(b_aux_bits))
    got_a, got_b, got_c = P.create_proof(worker, params, assignment, r, s, concurrent=concurrent)

    # ---- the same proof from the oracle
    host = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
    dom = lambda v, op: _ints(O.fr_domain_op(_limbs(v), log_m, op))  # noqa: E731
    ev = [dom(dom(v, "ifft"), "coset_fft") for v in abc]
    rinv = pow(M.MONT_R, -1, R)
    zinv_mont = pow((pow(7, m, R) - 1) % R, -1, R) * M.MONT_R % R
    mmul = lambda x, y: x * y * rinv % R  # noqa: E731  (Montgomery product of Montgomery values)
    hq = [mmul((mmul(x, y) - z) % R, zinv_mont) for x, y, z in zip(*ev)]
    hq = dom(hq, "icoset_fft")[:m - 1]
    h_canon = _limbs([v * rinv % R for v in hq])                       # scalars_into_representations
    inp_l, aux_l = _limbs(inp_c), _limbs(aux_c)

    def mexp(G, bases, off, scalars, bits=None):
        rc, out = G.multiexp(host(bases), scalars, density=None if bits is None else GU.density_words(bits),
                             density_bits=None if bits is None else len(bits), base_offset=off, threads=8)
        assert rc == 0
        return out

    h_o = mexp(O.G1, h_b, 0, h_canon)
    l_o = mexp(O.G1, l_b, 0, aux_l)
    a_ans = O.G1.add(mexp(O.G1, a_b, 0, inp_l), mexp(O.G1, a_b, num_inputs, aux_l, a_aux_bits))
    nbi = int(b_in_bits.sum())
    b1_ans = O.G1.add(mexp(O.G1, b1_b, 0, inp_l, b_in_bits), mexp(O.G1, b1_b, nbi, aux_l, b_aux_bits))
    b2_ans = O.G2.add(mexp(O.G2, b2_b, 0, inp_l, b_in_bits), mexp(O.G2, b2_b, nbi, aux_l, b_aux_bits))
    k = lambda v: np.array(M.to_limbs(v % R), dtype=np.uint64)  # noqa: E731
    g_a = O.G1.add_mixed(O.G1.mul(O.G1.from_affine(vk["delta_g1"]), k(r)), vk["alpha_g1"])
    g_b = O.G2.add_mixed(O.G2.mul(O.G2.from_affine(vk["delta_g2"]), k(s)), vk["beta_g2"])
    g_c = O.G1.mul(O.G1.from_affine(vk["delta_g1"]), k(r * s))
    g_c = O.G1.add(g_c, O.G1.mul(O.G1.from_affine(vk["alpha_g1"]), k(s)))
    g_c = O.G1.add(g_c, O.G1.mul(O.G1.from_affine(vk["beta_g1"]), k(r)))
    g_a = O.G1.add(g_a, a_ans)
    g_c = O.G1.add(g_c, O.G1.mul(a_ans, k(s)))
    g_b = O.G2.add(g_b, b2_ans)
    g_c = O.G1.add(g_c, O.G1.mul(b1_ans, k(r)))
    g_c = O.G1.add(O.G1.add(g_c, h_o), l_o)
    assert np.array_equal(got_a, O.G1.to_affine(g_a))
    assert np.array_equal(got_b, O.G2.to_affine(g_b))
    assert np.array_equal(got_c, O.G1.to_affine(g_c))
    # the ProvingAssignment is not consumed: EvaluationDomain.from_coeffs copies its input (the reference moves the Vec into the
    # domain, domain.rs:52), so a second proof from the same assignment is the same proof
    again = P.create_proof(worker, params, assignment, r, s, concurrent=concurrent)
    assert np.array_equal(again[0], got_a) and np.array_equal(again[1], got_b) and np.array_equal(again[2], got_c)
    # Parameters with window tables (table mode: every vector long enough here by lowering the thresholds): the same proof
    tabled = params.with_tables(g1_min=1, g2_min=1)
    assert isinstance(tabled.h, zk.MsmTable) and isinstance(tabled.b_g2, zk.MsmTable)
    third = P.create_proof(worker, tabled, assignment, r, s, concurrent=concurrent)
    assert np.array_equal(third[0], got_a) and np.array_equal(third[1], got_b) and np.array_equal(third[2], got_c)
