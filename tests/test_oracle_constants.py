"""The literals the reference DOES contain pin the representation conventions of the oracle and of
the HIP library: Montgomery R = 2^256, limb order, moduli, generators, two-adicity."""
import numpy as np

import bn254_model as M
import oracle_lib as O

# pairing/src/bn256/fq.rs:11-16  B_COEFF (Montgomery form of 3)
B_COEFF = [0x7a17caa950ad28d7, 0x1f6ac17ae15521b9, 0x334bea4e696bd284, 0x2a1f6744ce179d8e]
# pairing/src/bn256/fq.rs:39-50  G1 generator (1, 2) in Montgomery form
G1_GENERATOR_X = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
G1_GENERATOR_Y = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]
# pairing/src/bn256/fq.rs:60-83  G2 generator coordinates in Montgomery form
G2_GENERATOR_X_C0 = [0x8e83b5d102bc2026, 0xdceb1935497b0172, 0xfbb8264797811adf, 0x19573841af96503b]
G2_GENERATOR_X_C1 = [0xafb4737da84c6140, 0x6043dd5a5802d8c4, 0x09e950fc52a02f86, 0x14fef0833aea7b6b]
G2_GENERATOR_Y_C0 = [0x619dfa9d886be9f6, 0xfe7fd297f59e9b78, 0xff9e1a62231b7dfe, 0x28fd7eebae9e4206]
G2_GENERATOR_Y_C1 = [0x64095b56c71856ee, 0xdc57f922327d3cbb, 0x55f935be33351076, 0x0da4a0e693fd6482]
# pairing/src/bn256/fq.rs:18-31  B_COEFF_FQ2 = 3/(9+u)
B_COEFF_FQ2_C0 = [0x3bf938e377b802a8, 0x020b1b273633535d, 0x26b7edf049755260, 0x2514c6324384a86d]
B_COEFF_FQ2_C1 = [0x38e7ecccd1dcff67, 0x65f0b37d93ce0d3e, 0xd749d0dd22ac00aa, 0x0141b9ce4a688d4d]


def test_montgomery_r_is_2_256():
    assert M.from_limbs(G1_GENERATOR_X) == M.MONT_R % M.Q
    assert M.from_limbs(G1_GENERATOR_Y) == 2 * M.MONT_R % M.Q
    assert M.from_limbs(B_COEFF) == 3 * M.MONT_R % M.Q
    assert list(O.fe_from_canonical(O.FQ, [1, 0, 0, 0])) == G1_GENERATOR_X
    assert list(O.fe_from_canonical(O.FQ, [2, 0, 0, 0])) == G1_GENERATOR_Y
    assert list(O.fe_from_canonical(O.FQ, [3, 0, 0, 0])) == B_COEFF
    assert list(O.fe_to_canonical(O.FQ, G1_GENERATOR_Y)) == [2, 0, 0, 0]


def test_g2_generator_literals_match_model_and_curve():
    raw = M.g2_affine_to_raw(M.G2_GEN)
    assert raw == G2_GENERATOR_X_C0 + G2_GENERATOR_X_C1 + G2_GENERATOR_Y_C0 + G2_GENERATOR_Y_C1
    assert M.on_curve_g2(M.G2_GEN)
    assert M.to_limbs(M.to_mont(M.B_G2[0], M.Q)) == B_COEFF_FQ2_C0
    assert M.to_limbs(M.to_mont(M.B_G2[1], M.Q)) == B_COEFF_FQ2_C1


def test_fr_two_adicity_and_root_of_unity():
    # fr.rs:31-34 Fr::S == 28 ; root_of_unity = 7^((r-1)/2^28) has order exactly 2^28
    assert (M.R_ORDER - 1) % (1 << 28) == 0 and ((M.R_ORDER - 1) >> 28) % 2 == 1
    w = M.from_mont(M.from_limbs(O.fr_root_of_unity()), M.R_ORDER)
    assert w == M.FR_ROOT_OF_UNITY
    assert pow(w, 1 << 28, M.R_ORDER) == 1 and pow(w, 1 << 27, M.R_ORDER) != 1


def test_num_bits():
    # fq.rs:520-524 Fq::NUM_BITS == 254 (Fr likewise: multiexp's `skip >= NUM_BITS` loop bound)
    assert M.Q.bit_length() == 254 and M.R_ORDER.bit_length() == 254


def test_domain_constants_oracle_vs_model():
    for log_n in (0, 1, 5, 20, 28):
        omega, omegainv, geninv, minv = O.fr_domain(log_n)
        w = M.domain_omega(log_n)
        f = lambda a: M.from_mont(M.from_limbs(a), M.R_ORDER)  # noqa: E731
        assert f(omega) == w and f(omegainv) == pow(w, -1, M.R_ORDER)
        assert f(geninv) == pow(7, -1, M.R_ORDER) and f(minv) == pow(1 << log_n, -1, M.R_ORDER)
    assert O.fr_domain(29) is None  # PolynomialDegreeTooLarge, domain.rs:75-77
