"""The reference's only hard-coded known-answer vectors on this path: `test_xordemo`
(bellman/src/groth16/tests/mod.rs:28-330) over the DummyEngine (bellman/src/tests/dummy_engine.rs:
Fr = Z/64513, G1 = G2 = Fr as an additive group).  The oracle's multiexp and FFT templates are
instantiated over that engine, so these literals pin the ALGORITHM STRUCTURE of the restatement
(window/zero/one/density handling, summation by parts, window join; ifft / coset_fft / icoset_fft
chain), independently of BN254 arithmetic."""
import numpy as np

import oracle_lib as O

P = 64513
# groth16/tests/mod.rs:30-36, 225-226
ALPHA, BETA, GAMMA, DELTA, TAU, R, S = 48577, 22580, 53332, 5481, 3673, 27134, 17146
# constraint system of XORDemo (groth16/tests/mod.rs:57-67, table :134-146), variables (a_0=1, a_1=c, a_2=a, a_3=b)
A_ROWS = [[1, 0, P - 1, 0], [1, 0, 0, P - 1], [0, 0, 2, 0], [1, 0, 0, 0], [0, 1, 0, 0]]
B_ROWS = [[0, 0, 1, 0], [0, 0, 0, 1], [0, 0, 0, 1], [0, 0, 0, 0], [0, 0, 0, 0]]
C_ROWS = [[0, 0, 0, 0], [0, 0, 0, 0], [0, P - 1, 1, 1], [0, 0, 0, 0], [0, 0, 0, 0]]
U_I = [59158, 48317, 21767, 10402]   # :162
V_I = [0, 0, 60619, 30791]           # :165
W_I = [0, 23320, 41193, 41193]       # :168
H_COEFFS = [5040, 11763, 10755, 63633, 128, 9747, 8739]  # :314


def test_domain_root_of_unity():
    # :72-80: 2^10-th root 57751 squared 7 times is the 8-point domain's omega = 20201
    assert O.dummy_domain_omega(3) == 20201
    assert O.dummy_domain_omega(10) == 57751


def _lagrange_at_tau():
    # generator.rs:300-312: powers of tau, then ifft -> Lagrange coefficients L_j(tau) on the 8-point domain
    powers = [pow(TAU, i, P) for i in range(8)]
    return [int(v) for v in O.dummy_domain_op(powers, 3, "ifft")]


def test_lagrange_evaluations_pin_ifft():
    lag = _lagrange_at_tau()
    for rows, want in ((A_ROWS, U_I), (B_ROWS, V_I), (C_ROWS, W_I)):
        got = [sum(rows[j][i] * lag[j] for j in range(5)) % P for i in range(4)]
        assert got == want


def test_h_coefficients_pin_fft_chain():
    """prover.rs:217-241: a,b,c evaluations -> ifft, coset_fft, a*b-c, divide_by_z_on_coset, icoset_fft;
    the first 7 coefficients are the H-query scalars asserted at groth16/tests/mod.rs:314."""
    assign = [1, 1, 1, 0]  # a_0 = 1, c = a xor b = 1, a = 1, b = 0   (:229-233)
    ev = lambda rows: [sum(r[i] * assign[i] for i in range(4)) % P for r in rows] + [0, 0, 0]  # noqa: E731
    a, b, c = ev(A_ROWS), ev(B_ROWS), ev(C_ROWS)
    chain = lambda v: O.dummy_domain_op(O.dummy_domain_op(v, 3, "ifft"), 3, "coset_fft")  # noqa: E731
    a, b, c = chain(a), chain(b), chain(c)
    ab_c = [(int(x) * int(y) - int(z)) % P for x, y, z in zip(a, b, c)]
    zinv = pow((pow(5, 8, P) - 1) % P, P - 2, P)  # z(g) = g^m - 1 on the coset, g = 5 (domain.rs:207-218)
    h = O.dummy_domain_op([v * zinv % P for v in ab_c], 3, "icoset_fft")
    assert [int(v) for v in h[:7]] == H_COEFFS and int(h[7]) == 0


def test_proof_elements_pin_multiexp():
    """proof.a / proof.b (groth16/tests/mod.rs:250-281) through multiexp with scalars 1 and 0, an
    offset source for the aux part and density maps (prover.rs:282-293)."""
    lag = _lagrange_at_tau()
    u = [sum(A_ROWS[j][i] * lag[j] for j in range(5)) % P for i in range(4)]
    v = [sum(B_ROWS[j][i] * lag[j] for j in range(5)) % P for i in range(4)]
    assert u == U_I and v == V_I
    inputs_assign, aux_assign = [1, 1], [1, 0]
    # A query: all four variables appear (params.a has 4 elements, :151); inputs then aux (offset 2)
    rc, a_in = O.dummy_multiexp(u, inputs_assign)
    rc2, a_aux = O.dummy_multiexp(u, aux_assign, density=[0b11], density_bits=2, base_offset=2)
    assert rc == 0 and rc2 == 0
    assert (DELTA * R + ALPHA + a_in + a_aux) % P == (DELTA * R + ALPHA + U_I[0] + U_I[1] + U_I[2]) % P
    # B query: only a_2, a_3 have non-zero terms (params.b_g1 has 2 elements, :154); the input density is empty
    b_bases = [x for x in V_I if x != 0]
    rc, b_in = O.dummy_multiexp(b_bases, inputs_assign, density=[0b00], density_bits=2)
    rc2, b_aux = O.dummy_multiexp(b_bases, aux_assign, density=[0b11], density_bits=2, base_offset=0)
    assert rc == 0 and rc2 == 0 and b_in == 0
    assert (DELTA * S + BETA + b_in + b_aux) % P == (DELTA * S + BETA + V_I[0] + V_I[1] + V_I[2]) % P


def test_dummy_multiexp_general_scalars_equal_naive():
    import random
    rnd = random.Random(1)
    for n in (1, 5, 31, 32, 200):
        bases = [rnd.randrange(1, P) for _ in range(n)]
        sc = [rnd.randrange(P) for _ in range(n)]
        rc, got = O.dummy_multiexp(bases, sc)
        assert rc == 0 and got == sum(b * s for b, s in zip(bases, sc)) % P
    rc, _ = O.dummy_multiexp([3, 0, 5], [2, 9, 4])   # identity base, non-zero scalar (source.rs:50-52)
    assert rc == 1
    rc, got = O.dummy_multiexp([3, 0, 5], [2, 0, 4])  # ... fine under a zero scalar
    assert rc == 0 and got == (6 + 20) % P
    rc, _ = O.dummy_multiexp([3, 4], [2, 9, 4])      # bases exhausted (source.rs:46-48)
    assert rc == 2
