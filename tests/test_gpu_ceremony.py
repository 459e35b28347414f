"""The ceremony-side mirror (phase2-bn254_amd/ceremony.py) driven the way the reference's own tests drive the functions
it replaces: powersoftau/src/utils.rs:90-109 `test_power_pairs` (the pairing check `same_ratio(.., (g2, g2^x))` is
replaced by the equivalent statement with the known x: sx == x * s), the tau-power `batch_exp` of
batched_accumulator.rs:1130-1181, the QAP sums of parameters.rs:281-294 and a codec round trip with its error."""
import numpy as np
import pytest

import bn254_model as M
import inputs
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch

    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).cuda()


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def _limbs(v):
    return np.array(M.to_limbs(v % M.R_ORDER), dtype=np.uint64)


def test_power_pairs_like_the_reference(zk, worker):
    x = 0x1F3C5A7E9B2D4F60718293A4B5C6D7E8F9 % M.R_ORDER
    powers = np.stack([_limbs(pow(x, i, M.R_ORDER)) for i in range(100)])
    one = _dev(np.tile(inputs.G1_GEN_RAW, (100, 1)))
    v = zk.ceremony.batch_exp(one, _dev(powers))                      # v[i] = x^i * G, affine (utils.rs:94-100)
    assert np.array_equal(_host(v)[:3], O.G1.mul_many_affine(inputs.G1_GEN_RAW, powers[:3]))
    rho = _dev(inputs.random_scalars(99, seed=801))
    s, sx = zk.ceremony.power_pairs(v, rho)
    assert np.array_equal(O.G1.to_affine(sx), O.G1.to_affine(O.G1.mul(s, _limbs(x))))          # same_ratio(power_pairs(v), (g2, g2^x))
    hv = _host(v).copy()
    hv[1] = O.G1.to_affine(O.G1.mul(O.G1.from_affine(hv[1]), _limbs(12345)))                    # utils.rs:106
    s2, sx2 = zk.ceremony.power_pairs(_dev(hv), rho)
    assert not np.array_equal(O.G1.to_affine(sx2), O.G1.to_affine(O.G1.mul(s2, _limbs(x))))


def test_contribute_like_batch_exp_same_scalar_g2(zk, worker):
    pts = inputs.bases_progression_cpu(2, 20, seed=810)
    delta_inv = _limbs(pow(0xDEADBEEFCAFE, -1, M.R_ORDER))
    out = zk.ceremony.batch_exp(_dev(pts), _dev(delta_inv.reshape(1, 4)), same_scalar=True)     # parameters.rs:423-470
    for i in (0, 7, 19):
        assert np.array_equal(_host(out)[i], O.G2.to_affine(O.G2.mul(O.G2.from_affine(pts[i]), delta_inv)))


def test_eval_qap_and_dense_multiexp(zk, worker):
    bases = inputs.bases_progression_cpu(1, 16, seed=820)
    row_ptr = np.array([0, 2, 2, 5], dtype=np.int32)
    col = np.array([3, 9, 0, 15, 3], dtype=np.int32)
    coeff = inputs.random_scalars(5, seed=821)
    out = _host(zk.ceremony.eval_qap(_dev(bases), _dev(row_ptr), _dev(col), _dev(coeff)))
    for r in range(3):
        acc = O.G1.from_affine(np.zeros(8, np.uint64))
        for t in range(row_ptr[r], row_ptr[r + 1]):
            acc = O.G1.add(acc, O.G1.mul(O.G1.from_affine(bases[col[t]]), coeff[t]))
        assert np.array_equal(out[r], O.G1.to_affine(acc))
    ks = inputs.random_scalars(16, seed=822)
    got = zk.ceremony.dense_multiexp(_dev(bases), _dev(ks))
    assert np.array_equal(O.G1.to_affine(got), O.G1.to_affine(O.G1.naive_multiexp(bases, ks)))


def test_codec_roundtrip_and_error(zk, worker):
    pts = inputs.bases_progression_cpu(1, 33, seed=830)
    enc = zk.ceremony.encode_points(_dev(pts), compressed=True)
    assert np.array_equal(enc.cpu().numpy(), O.encode_points(1, pts, True))
    back = zk.ceremony.decode_points(enc, 1, compressed=True)
    assert np.array_equal(_host(back), pts)
    bad = enc.clone()
    bad[20, 0] = 0x7F                                               # infinity flag + stray bits
    with pytest.raises(zk.ceremony.GroupDecodingError) as e:
        zk.ceremony.decode_points(bad, 1, compressed=True)
    assert e.value.kind == "UnexpectedInformation" and e.value.index == 20
    lag = zk.ceremony.point_ifft(_dev(inputs.bases_progression_cpu(1, 8, seed=831)))
    assert np.array_equal(_host(zk.ceremony.point_fft(lag)), inputs.bases_progression_cpu(1, 8, seed=831))


def test_accumulator_and_phase1radix_containers(zk, worker):
    """File layouts around the codecs (batched_accumulator.rs:88-170, parameters.rs:74-105 / 147-217): a power-3 accumulator
    and an m = 8 phase1radix2m file are laid out on the CPU with the oracle's encoder, parsed on the device, compared
    element for element, written back byte for byte; a corrupted element and a point at infinity raise what the reference raises."""
    import torch

    power, m = 3, 8
    g1 = inputs.bases_progression_cpu(1, 64, seed=850)
    g2 = inputs.bases_progression_cpu(2, 32, seed=851)
    for compressed in (False, True):
        layout, total = zk.ceremony.accumulator_layout(power, compressed)
        assert [c for _, _, c, _ in layout] == [15, 8, 8, 8, 1]
        g1_sz, g2_sz = (32, 64) if compressed else (64, 128)
        assert total == 64 + 15 * g1_sz + 8 * g2_sz + 8 * g1_sz + 8 * g1_sz + g2_sz  # parameters.rs:83-89 / 99-105
        blob = np.zeros(total, np.uint8)
        blob[:64] = np.arange(64, dtype=np.uint8)
        want, i1, i2 = {}, 0, 0
        for name, g, cnt, off in layout:
            src = g1 if g == 1 else g2
            start = i1 if g == 1 else i2
            want[name] = src[start:start + cnt]
            if g == 1: i1 += cnt
            else: i2 += cnt
            enc = O.encode_points(g, want[name], compressed)
            blob[off:off + enc.size] = enc.reshape(-1)
        acc = zk.ceremony.read_accumulator(torch.from_numpy(blob).cuda(), power, compressed)
        assert bytes(acc["hash"].cpu().numpy()) == bytes(range(64))
        for name in want:
            assert np.array_equal(_host(acc[name]), want[name]), name
        assert np.array_equal(zk.ceremony.write_accumulator(acc, compressed).cpu().numpy(), blob)
        bad = blob.copy()
        off_alpha = [o for nme, _, _, o in layout if nme == "alpha_g1"][0]
        bad[off_alpha + 2 * g1_sz] = 0x7F
        with pytest.raises(zk.ceremony.GroupDecodingError) as e:
            zk.ceremony.read_accumulator(torch.from_numpy(bad).cuda(), power, compressed)
        assert e.value.index == 2
        bad = blob.copy()
        off_tau2 = [o for nme, _, _, o in layout if nme == "tau_g2"][0]
        bad[off_tau2:off_tau2 + g2_sz] = 0
        bad[off_tau2] = 0x40
        with pytest.raises(zk.ceremony.DeserializationError):
            zk.ceremony.read_accumulator(torch.from_numpy(bad).cuda(), power, compressed)
    params = {"alpha_g1": g1[:1], "beta_g1": g1[1:2], "beta_g2": g2[:1], "coeffs_g1": g1[2:10], "coeffs_g2": g2[1:9],
              "alpha_coeffs_g1": g1[10:18], "beta_coeffs_g1": g1[18:26], "h": g1[26:33]}
    blob = np.concatenate([O.encode_points(1 if v.shape[1] == 8 else 2, v, False).reshape(-1) for v in params.values()])
    got = zk.ceremony.read_phase1radix2m(torch.from_numpy(blob).cuda(), m)
    for name, v in params.items():
        assert np.array_equal(_host(got[name]), v), name
    assert np.array_equal(zk.ceremony.write_phase1radix2m(got).cpu().numpy(), blob)


def test_prepare_phase2_flow_closed_form(zk, worker):
    """prepare_phase2.rs:60-160 assembled from the pieces (containers -> point iffts -> H bases -> phase1radix2m file) on a
    power-3 accumulator with KNOWN tau, alpha, beta.  Closed forms: coeffs_g1[j] = L_j(tau) G, alpha_coeffs_g1[j] = alpha L_j(tau) G,
    coeffs_g2[j] = L_j(tau) G2, h[i] = (tau^m - 1) tau^i G;  L_j(tau) = (tau^m - 1) w^j / (m (tau - w^j))."""
    import torch

    power = 3
    m = 1 << power
    r = M.R_ORDER
    tau, alpha, beta = 0x1234567 % r, 0x89ABCDEF01 % r, 0x55AA55AA55 % r
    mul1 = lambda ks: O.G1.mul_many_affine(inputs.G1_GEN_RAW, np.stack([_limbs(k) for k in ks]))  # noqa: E731
    mul2 = lambda ks: O.G2.mul_many_affine(inputs.G2_GEN_RAW, np.stack([_limbs(k) for k in ks]))  # noqa: E731
    tp = [pow(tau, i, r) for i in range(2 * m - 1)]
    acc = {"hash": torch.zeros(64, dtype=torch.uint8).cuda(), "tau_g1": _dev(mul1(tp)), "tau_g2": _dev(mul2(tp[:m])),
           "alpha_g1": _dev(mul1([alpha * t for t in tp[:m]])), "beta_g1": _dev(mul1([beta * t for t in tp[:m]])), "beta_g2": _dev(mul2([beta]))}
    blob = zk.ceremony.write_accumulator(acc, compressed=True)           # a (compressed) response body ...
    acc2 = zk.ceremony.read_accumulator(blob, power, compressed=True)    # ... read back the way prepare_phase2 does
    params = zk.ceremony.prepare_phase2(acc2, m)
    w = M.domain_omega(power)
    lag = [(pow(tau, m, r) - 1) * pow(w, j, r) % r * pow(m * (tau - pow(w, j, r)) % r, -1, r) % r for j in range(m)]
    assert np.array_equal(_host(params["coeffs_g1"]), mul1(lag))
    assert np.array_equal(_host(params["coeffs_g2"]), mul2(lag))
    assert np.array_equal(_host(params["alpha_coeffs_g1"]), mul1([alpha * l for l in lag]))
    assert np.array_equal(_host(params["beta_coeffs_g1"]), mul1([beta * l for l in lag]))
    assert np.array_equal(_host(params["h"]), mul1([(pow(tau, m, r) - 1) * tp[i] for i in range(m - 1)]))
    radix = zk.ceremony.write_phase1radix2m(params)
    assert radix.numel() == 2 * 64 + 128 + m * 64 + m * 128 + 2 * m * 64 + (m - 1) * 64    # parameters.rs:183-217 reads exactly this
    back = zk.ceremony.read_phase1radix2m(radix, m)
    assert all(torch.equal(back[k], params[k]) for k in params)


def test_mpc_parameters_new_from_a_circom_circuit(zk, worker):
    """MPCParameters::new (phase2/src/parameters.rs:99-400) end to end: circom circuit.json -> KeypairAssembly -> the QAP sums over
    the Lagrange bases of a phase1radix2m file -> Groth16 parameters, cs_hash, container round trip.  The radix file comes from a
    power-3 accumulator with KNOWN tau, alpha, beta, so every element has a closed form in the scalar field:
        a[v] = A_v(tau) G,  b_g1[v] = B_v(tau) G,  b_g2[v] = B_v(tau) G2,  ic / l [v] = (beta A_v + alpha B_v + C_v)(tau) G,
    with A_v = sum of coeff * L_j over the variable's terms, L_j(tau) = (tau^m - 1) w^j / (m (tau - w^j))."""
    import hashlib

    import torch

    r = M.R_ORDER
    circuit_json = {   # x1 * x2 = x3;  (x3 + 5) * 1 = out;   0 = ONE, 1 = out, 2 = x1 (public), 3 = x2, 4 = x3
        "constraints": [[{"2": "1"}, {"3": "1"}, {"4": "1"}], [{"4": "1", "0": "5"}, {"0": "1"}, {"1": str(r - 1), "0": "0"}]],
        "nPubInputs": 1, "nOutputs": 1, "nVars": 5}
    circuit = zk.circom.circuit_from_json(circuit_json)
    cs = zk.circom.assemble(circuit)
    power = zk.circom.domain_exponent(cs.num_constraints)
    assert (cs.num_inputs, cs.num_aux, cs.num_constraints, power) == (3, 2, 5, 3)
    m = 1 << power
    tau, alpha, beta = 0x1234567 % r, 0x89ABCDEF01 % r, 0x55AA55AA55 % r
    mul1 = lambda ks: O.G1.mul_many_affine(inputs.G1_GEN_RAW, np.stack([_limbs(k % r) for k in ks]))  # noqa: E731
    mul2 = lambda ks: O.G2.mul_many_affine(inputs.G2_GEN_RAW, np.stack([_limbs(k % r) for k in ks]))  # noqa: E731
    tp = [pow(tau, i, r) for i in range(2 * m - 1)]
    acc = {"hash": torch.zeros(64, dtype=torch.uint8).cuda(), "tau_g1": _dev(mul1(tp)), "tau_g2": _dev(mul2(tp[:m])),
           "alpha_g1": _dev(mul1([alpha * t for t in tp[:m]])), "beta_g1": _dev(mul1([beta * t for t in tp[:m]])), "beta_g2": _dev(mul2([beta]))}
    radix = zk.ceremony.read_phase1radix2m(zk.ceremony.write_phase1radix2m(zk.ceremony.prepare_phase2(acc, m)), m)

    mpc = zk.circom.mpc_parameters_new(circuit, False, radix)
    w = M.domain_omega(power)
    lag = [(pow(tau, m, r) - 1) * pow(w, j, r) % r * pow(m * (tau - pow(w, j, r)) % r, -1, r) % r for j in range(m)]
    ev = lambda rows: [sum(c * lag[j] for c, j in row) % r for row in rows]  # noqa: E731
    A, B, Cc = ev(cs.at_inputs + cs.at_aux), ev(cs.bt_inputs + cs.bt_aux), ev(cs.ct_inputs + cs.ct_aux)
    P = mpc["params"]
    assert np.array_equal(_host(P["a"]), mul1(A)) and np.array_equal(_host(P["b_g1"]), mul1(B)) and np.array_equal(_host(P["b_g2"]), mul2(B))
    ext = [(beta * a + alpha * b + c) % r for a, b, c in zip(A, B, Cc)]
    assert np.array_equal(_host(P["vk"]["ic"]), mul1(ext[:3])) and np.array_equal(_host(P["l"]), mul1(ext[3:]))
    assert np.array_equal(_host(P["h"]), mul1([(pow(tau, m, r) - 1) * tp[i] for i in range(m - 1)]))
    assert np.array_equal(_host(P["vk"]["alpha_g1"]), mul1([alpha])) and np.array_equal(_host(P["vk"]["beta_g2"]), mul2([beta]))
    assert np.array_equal(_host(P["vk"]["delta_g1"])[0], inputs.G1_GEN_RAW) and np.array_equal(_host(P["vk"]["gamma_g2"])[0], inputs.G2_GEN_RAW)
    assert A[3] == 0 and all(A[i] for i in (0, 1, 2, 4)) and [bool(v) for v in B] == [True, False, False, True, False]   # x2 has no A term (every input has its x * 0 = 0): infinity in the unfiltered queries
    blob = zk.ceremony.write_parameters(P)
    assert bytes(mpc["cs_hash"].cpu().numpy()) == hashlib.blake2b(bytes(blob.cpu().numpy()), digest_size=64).digest()
    back = zk.ceremony.read_mpc_parameters(zk.ceremony.write_mpc_parameters(mpc), disallow_points_at_infinity=False)
    assert torch.equal(back["cs_hash"], mpc["cs_hash"]) and back["contributions"] == []
    assert all(torch.equal(back["params"][k], P[k]) for k in ("h", "l", "a", "b_g1", "b_g2")) and torch.equal(back["params"]["vk"]["ic"], P["vk"]["ic"])

    # ---- prove (circom_circuit.rs:187-191) on these parameters with a witness: delta = gamma = 1 and tau, alpha, beta are known, so the
    # proof has closed-form discrete logarithms:  A = alpha + A(tau) + r,  B = beta + B(tau) + s,
    # C = L_aux + (A(tau) B(tau) - C(tau)) + s A + r B - r s   (the H query contributes h(tau) t(tau) = A B - C)
    x1, x2 = 3, 4
    wit = [1, (-(x1 * x2 + 5)) % r, x1, x2, x1 * x2]                      # ONE, out, x1, x2, x3  (the circuit's C combination is -out)
    circuit.witness = zk.circom.witness_from_json([str(v) for v in wit])
    rr, ss = 0x1234567890ABCDEF1122334455667788 % r, 0x0FEDCBA9876543210F1E2D3C4B5A6978 % r
    pa, pb, pc = zk.circom.prove(worker, circuit, P, rr, ss)
    At, Bt, Ct = (sum(wv * v for wv, v in zip(wit, X)) % r for X in (A, B, Cc))
    log_a, log_b = (alpha + At + rr) % r, (beta + Bt + ss) % r
    log_c = (sum(wv * e for wv, e in zip(wit[3:], ext[3:])) + At * Bt - Ct + ss * log_a + rr * log_b - rr * ss) % r
    assert np.array_equal(pa, mul1([log_a])[0]) and np.array_equal(pb, mul2([log_b])[0]) and np.array_equal(pc, mul1([log_c])[0])
    # Groth16's check in the exponent (gamma = delta = 1):  A B = alpha beta + IC(public inputs) + C
    assert (log_a * log_b - alpha * beta - sum(wv * e for wv, e in zip(wit[:3], ext[:3])) - log_c) % r == 0

    filtered = zk.circom.mpc_parameters_new(circuit, True, radix)["params"]  # should_filter_points_at_infinity
    keep_a = [i for i, v in enumerate(A) if v]
    keep_b = [i for i, v in enumerate(B) if v]
    assert len(keep_a) < len(A) and np.array_equal(_host(filtered["a"]), mul1([A[i] for i in keep_a]))
    assert np.array_equal(_host(filtered["b_g1"]), mul1([B[i] for i in keep_b])) and np.array_equal(_host(filtered["b_g2"]), mul2([B[i] for i in keep_b]))
    assert torch.equal(filtered["l"], P["l"])
    # an auxiliary variable no constraint mentions: the L query would not be fully dense
    with pytest.raises(zk.SynthesisError) as e:
        zk.circom.mpc_parameters_new(zk.circom.circuit_from_json(dict(circuit_json, nVars=6)), False, radix)
    assert e.value.kind == zk.SynthesisError.UNCONSTRAINED_VARIABLE


def test_contribute_accumulator_like_compute_constrained(zk, worker):
    """BASELINE config 1's compute step on the device (batched_accumulator.rs:1119-1292): the blank accumulator of
    `new_constrained` (every element a generator, :1295-1347) contributed with a known key has tau_g1[i] = tau^i G,
    alpha_g1[i] = alpha tau^i G, ...; a second contribution multiplies the keys."""
    import torch

    power = 3
    n, n1 = 1 << power, (2 << power) - 1
    r = M.R_ORDER
    blank = {"hash": torch.zeros(64, dtype=torch.uint8).cuda(), "tau_g1": _dev(np.tile(inputs.G1_GEN_RAW, (n1, 1))),
             "tau_g2": _dev(np.tile(inputs.G2_GEN_RAW, (n, 1))), "alpha_g1": _dev(np.tile(inputs.G1_GEN_RAW, (n, 1))),
             "beta_g1": _dev(np.tile(inputs.G1_GEN_RAW, (n, 1))), "beta_g2": _dev(np.tile(inputs.G2_GEN_RAW, (1, 1)))}
    tau, alpha, beta = 0xABCDEF123456789 % r, 0x1111222233334444 % r, 0x9999AAAABBBB % r
    pw = zk.ceremony.scalar_powers(tau, 9, torch.device("cuda", 0), coeff=alpha)
    assert [M.from_limbs([int(v) for v in row]) for row in _host(pw)] == [alpha * pow(tau, i, r) % r for i in range(9)]
    acc = zk.ceremony.contribute_accumulator(blank, tau, alpha, beta)
    mul1 = lambda ks: O.G1.mul_many_affine(inputs.G1_GEN_RAW, np.stack([_limbs(k) for k in ks]))  # noqa: E731
    mul2 = lambda ks: O.G2.mul_many_affine(inputs.G2_GEN_RAW, np.stack([_limbs(k) for k in ks]))  # noqa: E731
    tp = [pow(tau, i, r) for i in range(n1)]
    assert np.array_equal(_host(acc["tau_g1"]), mul1(tp))
    assert np.array_equal(_host(acc["tau_g2"]), mul2(tp[:n]))
    assert np.array_equal(_host(acc["alpha_g1"]), mul1([alpha * t for t in tp[:n]]))
    assert np.array_equal(_host(acc["beta_g1"]), mul1([beta * t for t in tp[:n]]))
    assert np.array_equal(_host(acc["beta_g2"]), mul2([beta]))
    t2, a2, b2 = 0x31415926535 % r, 0x27182818 % r, 0x16180339 % r
    acc2 = zk.ceremony.contribute_accumulator(acc, t2, a2, b2)
    assert np.array_equal(_host(acc2["tau_g1"]), mul1([pow(tau * t2, i, r) for i in range(n1)]))
    assert np.array_equal(_host(acc2["alpha_g1"]), mul1([alpha * a2 * pow(tau * t2, i, r) for i in range(n)]))


def test_eval_qap_polynomials_like_mpc_parameters_new(zk, worker):
    """parameters.rs:225-300 on a toy QAP (5 variables over m = 8 Lagrange bases, ragged at / bt / ct incl. empty rows):
    a_g1, b_g1, b_g2 and ext against the oracle's mul / add, term by term."""
    import torch

    m, n_vars = 8, 5
    radix = {"coeffs_g1": inputs.bases_progression_cpu(1, m, seed=870), "coeffs_g2": inputs.bases_progression_cpu(2, m, seed=871),
             "alpha_coeffs_g1": inputs.bases_progression_cpu(1, m, seed=872), "beta_coeffs_g1": inputs.bases_progression_cpu(1, m, seed=873)}
    rng = np.random.default_rng(874)

    def poly(lengths, seed):
        rp = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
        col = rng.integers(0, m, size=int(rp[-1])).astype(np.int32)
        cf = inputs.random_scalars(int(rp[-1]), seed=seed)
        cf[::3] = np.array([1, 0, 0, 0], dtype=np.uint64)
        return rp, col, cf

    at, bt, ct = poly([2, 0, 3, 1, 1], 875), poly([1, 2, 0, 0, 4], 876), poly([0, 1, 1, 2, 0], 877)
    dev_radix = {k: _dev(v) for k, v in radix.items()}
    to_dev = lambda t: (_dev(t[0]), _dev(t[1]), _dev(t[2]))  # noqa: E731
    a_g1, b_g1, b_g2, ext = zk.ceremony.eval_qap_polynomials(dev_radix, to_dev(at), to_dev(bt), to_dev(ct))

    def rowsum(G, bases, t, v, acc=None):
        acc = G.from_affine(np.zeros(G.aff, np.uint64)) if acc is None else acc
        for j in range(t[0][v], t[0][v + 1]):
            acc = G.add(acc, G.mul(G.from_affine(bases[t[1][j]]), t[2][j]))
        return acc

    for v in range(n_vars):
        assert np.array_equal(_host(a_g1)[v], O.G1.to_affine(rowsum(O.G1, radix["coeffs_g1"], at, v)))
        assert np.array_equal(_host(b_g1)[v], O.G1.to_affine(rowsum(O.G1, radix["coeffs_g1"], bt, v)))
        assert np.array_equal(_host(b_g2)[v], O.G2.to_affine(rowsum(O.G2, radix["coeffs_g2"], bt, v)))
        e = rowsum(O.G1, radix["beta_coeffs_g1"], at, v)
        e = rowsum(O.G1, radix["alpha_coeffs_g1"], bt, v, e)
        e = rowsum(O.G1, radix["coeffs_g1"], ct, v, e)
        assert np.array_equal(_host(ext)[v], O.G1.to_affine(e))


@pytest.mark.parametrize("power", [10, 12])
def test_config1_new_constrained_challenge_hash(zk, worker, power):
    """BASELINE config 1, `new`: ceremony.new_accumulator (generate_initial, batched_accumulator.rs:1295-1347) serialised by the
    HIP encoders in the layout of batched_accumulator.rs:87-178 must hash (BLAKE2b-512, utils.rs:20-27) to the value the
    reference's layout fixes (tests/test_challenge_hash.py holds the same constants for the oracle's encoder)."""
    import torch

    from test_challenge_hash import CHALLENGE

    acc = zk.ceremony.new_accumulator(power, torch.device("cuda", 0))
    blob = zk.ceremony.write_accumulator(acc, compressed=False)
    size, digest = CHALLENGE[power]
    assert blob.numel() == size
    assert zk.ceremony.calculate_hash(blob).hex() == digest
    back = zk.ceremony.read_accumulator(blob, power, compressed=False)    # what compute_constrained reads
    assert all(torch.equal(back[k], acc[k]) for k in acc)


def test_config1_compute_constrained_power12(zk, worker):
    """BASELINE config 1, `compute` at REQUIRED_POWER = 12 (8191 TauG1 + 4096 x (TauG2, AlphaG1, BetaG1) + BetaG2) with a fixed
    key injected instead of OsRng (compute_constrained.rs:41-80): challenge file -> read_accumulator -> contribute_accumulator
    (batched_accumulator.rs:1119-1292) -> compressed response body.  EVERY element of every vector is checked against the closed
    form the relations of verify_transform (batched_accumulator.rs:182-272) imply -- tau_g1[i] = tau^i G, tau_g2[i] = tau^i G2,
    alpha_g1[i] = alpha tau^i G, beta_g1[i] = beta tau^i G, beta_g2 = beta G2 -- computed by the oracle's mul_assign; the
    response round-trips through the compressed codec and carries the challenge's hash (:1284-1290 writes it first)."""
    import torch

    power = 12
    n, n1 = 1 << power, (2 << power) - 1
    r = M.R_ORDER
    dev = torch.device("cuda", 0)
    challenge = zk.ceremony.write_accumulator(zk.ceremony.new_accumulator(power, dev), compressed=False)
    challenge_hash = zk.ceremony.calculate_hash(challenge)
    acc = zk.ceremony.read_accumulator(challenge, power, compressed=False)
    tau = 0x1F0E2D3C4B5A69788796A5B4C3D2E1F00112233445566778899AABBCCDDEEFF % r
    alpha = 0x2B7E151628AED2A6ABF7158809CF4F3C762E7160F38B4DA56A784D9045190CFE % r
    beta = 0x243F6A8885A308D313198A2E03707344A4093822299F31D0082EFA98EC4E6C89 % r
    out = zk.ceremony.contribute_accumulator(acc, tau, alpha, beta)
    out["hash"] = torch.frombuffer(bytearray(challenge_hash), dtype=torch.uint8).to(dev)
    tp = [1] * n1
    for i in range(1, n1):
        tp[i] = tp[i - 1] * tau % r
    limbs = lambda ks: np.array([M.to_limbs(k % r) for k in ks], dtype=np.uint64)  # noqa: E731
    assert np.array_equal(_host(out["tau_g1"]), O.G1.mul_many_affine(inputs.G1_GEN_RAW, limbs(tp)))
    assert np.array_equal(_host(out["tau_g2"]), O.G2.mul_many_affine(inputs.G2_GEN_RAW, limbs(tp[:n])))
    assert np.array_equal(_host(out["alpha_g1"]), O.G1.mul_many_affine(inputs.G1_GEN_RAW, limbs([alpha * t for t in tp[:n]])))
    assert np.array_equal(_host(out["beta_g1"]), O.G1.mul_many_affine(inputs.G1_GEN_RAW, limbs([beta * t for t in tp[:n]])))
    assert np.array_equal(_host(out["beta_g2"]), O.G2.mul_many_affine(inputs.G2_GEN_RAW, limbs([beta])))
    response = zk.ceremony.write_accumulator(out, compressed=True)
    _, body = zk.ceremony.accumulator_layout(power, compressed=True)
    assert response.numel() == body == 787_296 - (3 * 128 + 6 * 64)       # contribution_size - public_key_size, parameters.rs:97-107
    assert bytes(response[:64].cpu().numpy()) == challenge_hash
    back = zk.ceremony.read_accumulator(response, power, compressed=True)
    assert all(torch.equal(back[k], out[k]) for k in ("tau_g1", "tau_g2", "alpha_g1", "beta_g1", "beta_g2"))
    # the oracle's encoder agrees byte for byte on a sample of the response
    off_tau2 = [o for name, _, _, o in zk.ceremony.accumulator_layout(power, True)[0] if name == "tau_g2"][0]
    sample = response[off_tau2:off_tau2 + 64 * 64].cpu().numpy()
    assert np.array_equal(sample, O.encode_points(2, _host(out["tau_g2"])[:64], True).reshape(-1))


@pytest.mark.parametrize("group,log_n", [(1, 20), (2, 16)])
def test_config5_contribute_at_size(zk, worker, group, log_n):
    """BASELINE config 5: the device work of MPCParameters::contribute (phase2/src/parameters.rs:414-522) at |L| = 2^20,
    |H| = 2^20 - 1 G1 points (and the same kernel family over G2 at 2^16): every point times delta^-1 by `batch_exp`, affine out.
    Checks: (1) >= 1024 random indices of L' and H' against the oracle's mul_assign + into_affine, bit exact;
    (2) the statement verify_contribution checks with a pairing (merge_pairs + same_ratio against (delta_g2_after, delta_g2_before),
    parameters.rs:1038-1075 / utils.rs:59-105), here with the known delta:  sum rho_i L[i] == delta * sum rho_i L'[i];
    (3) an infinity record passes through as infinity."""
    import ctypes as C

    import torch

    import bench

    G = O.G1 if group == 1 else O.G2
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
    dev = torch.device("cuda", 0)
    L = zk.lib.load()
    r = M.R_ORDER

    def synth(n, seed):
        k = bench.gen_scalars(n, seed, dev)
        p = torch.empty((n, 8 * group), dtype=torch.int64, device=dev)
        fn = L.mi355zk_bn254_g1_batch_mul_dev if group == 1 else L.mi355zk_bn254_g2_batch_mul_dev
        assert fn(C.c_void_p(p.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n, None) == 0
        return p

    n_l, n_h = 1 << log_n, (1 << log_n) - 1
    l_before, h_before = synth(n_l, 2001), synth(n_h, 2002)
    l_before[12345 % n_l] = 0                                              # an infinity entry stays infinity
    delta = 0x0123456789ABCDEF0FEDCBA9876543210123456789ABCDEF02468ACE13579B % r
    delta_inv = _limbs(pow(delta, -1, r))
    d_inv = _dev(delta_inv.reshape(1, 4))
    l_after = zk.ceremony.batch_exp(l_before, d_inv, same_scalar=True)
    h_after = zk.ceremony.batch_exp(h_before, d_inv, same_scalar=True)
    rng = np.random.default_rng(2003)
    for before, after, n in ((l_before, l_after, n_l), (h_before, h_after, n_h)):
        idx = np.unique(np.concatenate([[0, 1, n - 2, n - 1, 12345 % n], rng.integers(0, n, size=1100)]))
        t_idx = torch.from_numpy(idx).to(dev)
        hb, ha = _host(before[t_idx]), _host(after[t_idx])
        assert len(idx) >= 1024
        for j in range(len(idx)):
            want = G.to_affine(G.mul(G.from_affine(hb[j]), delta_inv)) if hb[j].any() else np.zeros(8 * group, np.uint64)
            assert np.array_equal(ha[j], want), int(idx[j])
    assert not _host(l_after[12345 % n_l]).any()
    for before, after, n, seed in ((l_before, l_after, n_l, 2004), (h_before, h_after, n_h, 2005)):
        rho = bench.gen_scalars(n, seed, dev)
        s, sx = zk.ceremony.merge_pairs(before, after, rho)
        assert np.array_equal(G.to_affine(s), G.to_affine(G.mul(sx, _limbs(delta))))


def test_groth16_and_mpc_parameter_files(zk, worker):
    """Parameters::{read, write} (bellman/src/groth16/mod.rs:104-198, 252-383) and MPCParameters::{read, write} (phase2/src/
    parameters.rs:661-706) around the codec kernels, then the device work of `contribute` (:414-522) on the parsed file.
    The file is laid out on the CPU with the oracle's encoder; parsed on the device element for element; written back byte
    for byte; the two read flags and the never-infinity rules raise what the reference raises; after contribute(delta) the
    re-read file holds delta^-1 L, delta^-1 H, delta delta_g1, delta delta_g2 (oracle mul_assign), everything else untouched."""
    import torch

    g1 = inputs.bases_progression_cpu(1, 80, seed=880)
    g2 = inputs.bases_progression_cpu(2, 20, seed=881)
    sizes = {"ic": 3, "h": 15, "l": 13, "a": 16, "b_g1": 9}
    vk = {"alpha_g1": g1[0:1], "beta_g1": g1[1:2], "beta_g2": g2[0:1], "gamma_g2": g2[1:2], "delta_g1": g1[2:3], "delta_g2": g2[2:3], "ic": g1[3:6]}
    vecs = {"h": g1[6:21], "l": g1[21:34], "a": g1[34:50], "b_g1": g1[50:59], "b_g2": g2[3:12]}
    be = lambda n: np.frombuffer(int(n).to_bytes(4, "big"), np.uint8)  # noqa: E731
    enc = lambda pts: O.encode_points(1 if pts.shape[1] == 8 else 2, pts, False).reshape(-1)  # noqa: E731
    blob = [enc(vk[k]) for k in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "delta_g1", "delta_g2")] + [be(3), enc(vk["ic"])]
    for name in ("h", "l", "a", "b_g1", "b_g2"):
        blob += [be(len(vecs[name])), enc(vecs[name])]
    params_bytes = np.concatenate(blob)
    pk = {"delta_after": g1[60:61], "s": g1[61:62], "s_delta": g1[62:63], "r_delta": g2[12:13]}
    cs_hash, transcript = np.arange(64, dtype=np.uint8), np.arange(100, 164, dtype=np.uint8)
    mpc_bytes = np.concatenate([params_bytes, cs_hash, be(1), enc(pk["delta_after"]), enc(pk["s"]), enc(pk["s_delta"]), enc(pk["r_delta"]), transcript])
    assert mpc_bytes.size == params_bytes.size + 64 + 4 + 3 * 64 + 128 + 64

    mpc = zk.ceremony.read_mpc_parameters(torch.from_numpy(mpc_bytes).cuda())
    p = mpc["params"]
    for k in vk:
        assert np.array_equal(_host(p["vk"][k]), vk[k]), k
    for k in vecs:
        assert np.array_equal(_host(p[k]), vecs[k]), k
    assert bytes(mpc["cs_hash"].cpu().numpy()) == bytes(cs_hash) and len(mpc["contributions"]) == 1
    assert np.array_equal(_host(mpc["contributions"][0]["r_delta"]), pk["r_delta"])
    assert bytes(mpc["contributions"][0]["transcript"].cpu().numpy()) == bytes(transcript)
    assert np.array_equal(zk.ceremony.write_mpc_parameters(mpc).cpu().numpy(), mpc_bytes)
    assert np.array_equal(zk.ceremony.write_parameters(p).cpu().numpy(), params_bytes)

    # a point at infinity inside l: rejected unless the caller allows it (mod.rs:318-324)
    off_l = 576 + 4 + 3 * 64 + 4 + 15 * 64 + 4
    bad = params_bytes.copy()
    bad[off_l + 5 * 64:off_l + 6 * 64] = 0
    bad[off_l + 5 * 64] = 0x40
    with pytest.raises(zk.ceremony.DeserializationError):
        zk.ceremony.read_parameters(torch.from_numpy(bad).cuda(), disallow_points_at_infinity=True, checked=True)
    ok = zk.ceremony.read_parameters(torch.from_numpy(bad).cuda(), disallow_points_at_infinity=False, checked=True)
    assert not _host(ok["l"])[5].any()
    # a point off the curve inside a: NotOnCurve when checked, accepted unchecked (into_affine_unchecked)
    off_a = off_l + 13 * 64 + 4
    bad = params_bytes.copy()
    bad[off_a + 2 * 64 + 63] ^= 1
    with pytest.raises(zk.ceremony.GroupDecodingError) as e:
        zk.ceremony.read_parameters(torch.from_numpy(bad).cuda(), checked=True)
    assert e.value.kind == "NotOnCurve" and e.value.index == 2
    zk.ceremony.read_parameters(torch.from_numpy(bad).cuda(), checked=False)
    # ... but never in the verifying key (always checked)
    bad = params_bytes.copy()
    bad[63] ^= 1
    with pytest.raises(zk.ceremony.GroupDecodingError):
        zk.ceremony.read_parameters(torch.from_numpy(bad).cuda(), checked=False)
    with pytest.raises(ValueError):
        zk.ceremony.read_parameters(torch.from_numpy(params_bytes[:-10].copy()).cuda())
    # truncated MPC files: at every field boundary, one byte before it and in the middle of the field that follows it, the
    # reader fails like the reference's read_u32 / read_exact (UnexpectedEof) instead of returning short vectors, a short
    # cs_hash or zero contributions
    bounds, o = [], 0
    for sz in [64, 64, 128, 128, 64, 128, 4, 3 * 64, 4, 15 * 64, 4, 13 * 64, 4, 16 * 64, 4, 9 * 64, 4, 9 * 128, 64, 4, 64, 64, 64, 128, 64]:
        bounds.append((o, sz))
        o += sz
    assert o == mpc_bytes.size
    dev_bytes = torch.from_numpy(mpc_bytes).cuda()
    for start, sz in bounds:
        for cut in {start, start + 1, start + sz // 2, start + sz - 1}:
            if 0 <= cut < mpc_bytes.size:
                with pytest.raises(ValueError, match="too short"):
                    zk.ceremony.read_mpc_parameters(dev_bytes[:cut])
    zk.ceremony.read_mpc_parameters(dev_bytes[:mpc_bytes.size])

    # contribute: the heavy part of phase2 `contribute` on the parsed file, written back and re-read
    delta = 0x6A09E667F3BCC908BB67AE8584CAA73B3C6EF372FE94F82BA54FF53A5F1D36F1 % M.R_ORDER
    after = zk.ceremony.read_parameters(zk.ceremony.write_parameters(zk.ceremony.contribute_parameters(p, delta)))
    dinv, d = _limbs(pow(delta, -1, M.R_ORDER)), _limbs(delta)
    mul = lambda G, pts, k: np.stack([G.to_affine(G.mul(G.from_affine(x), k)) for x in pts])  # noqa: E731
    assert np.array_equal(_host(after["l"]), mul(O.G1, vecs["l"], dinv)) and np.array_equal(_host(after["h"]), mul(O.G1, vecs["h"], dinv))
    assert np.array_equal(_host(after["vk"]["delta_g1"]), mul(O.G1, vk["delta_g1"], d))
    assert np.array_equal(_host(after["vk"]["delta_g2"]), mul(O.G2, vk["delta_g2"], d))
    for k in ("a", "b_g1", "b_g2"):
        assert np.array_equal(_host(after[k]), vecs[k])
    for k in ("alpha_g1", "beta_g1", "beta_g2", "gamma_g2", "ic"):
        assert np.array_equal(_host(after["vk"][k]), vk[k])


def _pack16(limbs16):
    """(n, k <= 16) tensor of 16-bit values (int64) -> (n, 4) int64 bit patterns of u64 limbs, little-endian"""
    import torch

    n, k = limbs16.shape
    out = torch.zeros((n, 4), dtype=torch.int64, device=limbs16.device)
    for i in range(k):
        out[:, i // 4] |= limbs16[:, i] << (16 * (i % 4))
    return out


@pytest.mark.parametrize("group,trusted", [(1, 0), (2, 0), (2, 2)])
def test_config5_qap_evaluation_at_size(zk, worker, group, trusted):
    """BASELINE config 5, MPCParameters::new on a ~2^20-constraint circuit (phase2/src/parameters.rs:225-294): the per-variable sums
    a_g1 / b_g1 / b_g2 / ext as ONE CSR-matrix x point-vector product with 2^20 rows over 2^20 Lagrange points and ~3.2 M terms -- thirteen
    2^18-term chunks of the scalar-multiplication loop, the per-base membership split of G2 (default flags), rows that straddle chunk
    boundaries.  Circom-like coefficients: 10 % each of 1, r - 1 and 0; one 10^5-term row (the constant ONE of a circuit sits in most
    constraints); empty rows; an infinity base that several terms name.  Checks:
      (1) >= 1024 sampled rows + the rows at every chunk boundary + the long row + the rows naming the infinity base against the oracle's
          mul_assign / add_assign / into_affine, bit exact;
      (2) EVERY row at once through the linear form  sum_r rho_r out[r] == sum_j (M^T rho)_j bases[j]  with random 24-bit rho_r: both sides
          by the library's dense_multiexp (parity-tested on its own), M^T rho as exact integers on the device (16-bit limb columns, int64
          index_add), split S_j = lo_j + 2^240 hi_j so that both exponent vectors are canonical;
      (3) the host-buffer form over 3 logical devices (row ranges of equal weight) returns the same records."""
    import ctypes as C

    import torch

    import bench

    G = O.G1 if group == 1 else O.G2
    gen = np.ascontiguousarray(inputs.G1_GEN_RAW if group == 1 else inputs.G2_GEN_RAW)
    dev = torch.device("cuda", 0)
    L = zk.lib.load()
    n_rows = n_bases = 1 << 20
    inf_base, long_row, long_len = 777, 345_678, 100_000
    k = bench.gen_scalars(n_bases, 3100 + group, dev)
    bases = torch.empty((n_bases, 8 * group), dtype=torch.int64, device=dev)
    fn = L.mi355zk_bn254_g1_batch_mul_dev if group == 1 else L.mi355zk_bn254_g2_batch_mul_dev
    assert fn(C.c_void_p(bases.data_ptr()), gen.ctypes.data_as(C.c_void_p), C.c_void_p(k.data_ptr()), n_bases, None) == 0
    bases[inf_base] = 0
    g_ = torch.Generator(device=dev)
    g_.manual_seed(3110 + group)
    lens = torch.randint(0, 7, (n_rows,), device=dev, generator=g_, dtype=torch.int64)
    lens[long_row] = long_len
    lens[:3] = 0
    lens[n_rows - 1] = 0
    rp = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(lens, 0)
    nnz = int(rp[-1].item())
    assert nnz > 12 << 18                                                   # >= 13 chunks of 2^18 terms
    col = torch.randint(0, n_bases, (nnz,), device=dev, generator=g_, dtype=torch.int64)
    col[torch.randint(0, nnz, (64,), device=dev, generator=g_)] = inf_base  # terms that name the infinity base
    cf = bench.gen_scalars(nnz, 3120 + group, dev)
    kind = torch.randint(0, 10, (nnz,), device=dev, generator=g_)
    cf[kind == 0] = torch.tensor([1, 0, 0, 0], dtype=torch.int64, device=dev)
    cf[kind == 1] = torch.from_numpy(_limbs(M.R_ORDER - 1).view(np.int64)).to(dev)
    cf[kind == 2] = 0
    rp32, col32 = rp.to(torch.int32), col.to(torch.int32)
    out = zk.ceremony.eval_qap(bases, rp32, col32, cf, trusted_subgroup=bool(trusted))
    torch.cuda.synchronize()

    # (1) sampled rows against the oracle
    h_rp = rp.cpu().numpy()
    rng = np.random.default_rng(3130 + group)
    boundary_rows = np.searchsorted(h_rp, np.arange(1, nnz >> 18) << 18, side="right") - 1   # the row holding term k * 2^18
    inf_rows = np.searchsorted(h_rp, torch.nonzero(col == inf_base).flatten().cpu().numpy(), side="right") - 1
    rows = np.unique(np.concatenate([[0, 1, 2, 3, n_rows - 2, n_rows - 1, long_row - 1, long_row + 1], boundary_rows, boundary_rows + 1, inf_rows,
                                     rng.integers(0, n_rows, size=1060)]))
    rows = rows[(rows != long_row) & (rows < n_rows)]
    assert len(rows) >= 1024
    h_out = _host(out[torch.from_numpy(rows).to(dev)])
    zero = np.zeros(8 * group, np.uint64)
    for i, r_ in enumerate(rows):
        t0, t1 = int(h_rp[r_]), int(h_rp[r_ + 1])
        hb, hc = _host(bases[col[t0:t1]]), _host(cf[t0:t1])
        acc = G.from_affine(zero)
        for t in range(t1 - t0):
            acc = G.add(acc, G.mul(G.from_affine(hb[t]), hc[t]))
        assert np.array_equal(h_out[i], G.to_affine(acc)), int(r_)
    t0, t1 = int(h_rp[long_row]), int(h_rp[long_row + 1])
    assert t1 - t0 == long_len
    want = G.dense_multiexp(_host(bases[col[t0:t1]]), _host(cf[t0:t1]), cpus=8)
    assert np.array_equal(_host(out[long_row]), G.to_affine(want))
    assert not _host(out[:3]).any() and not _host(out[n_rows - 1]).any()   # empty rows are the infinity record

    # (2) every row: sum_r rho_r out[r] == sum_j (M^T rho)_j bases[j]
    rho = torch.randint(1, 1 << 24, (n_rows,), device=dev, generator=g_, dtype=torch.int64)
    rho_t = torch.repeat_interleave(rho, lens)
    s16 = torch.zeros((n_bases, 16), dtype=torch.int64, device=dev)
    c16 = torch.stack([(cf[:, i // 4] >> (16 * (i % 4))) & 0xFFFF for i in range(16)], dim=1)
    s16.index_add_(0, col, c16 * rho_t[:, None])
    del c16
    norm = torch.zeros((n_bases, 20), dtype=torch.int64, device=dev)
    carry = torch.zeros(n_bases, dtype=torch.int64, device=dev)
    for i in range(20):
        v = carry + (s16[:, i] if i < 16 else 0)
        norm[:, i] = v & 0xFFFF
        carry = v >> 16
    assert not carry.any()
    lo, hi = _pack16(norm[:, :15]), _pack16(norm[:, 15:])
    rho4 = torch.zeros((n_rows, 4), dtype=torch.int64, device=dev)
    rho4[:, 0] = rho
    lhs = zk.ceremony.dense_multiexp(out, rho4)
    rhs = G.add(zk.ceremony.dense_multiexp(bases, lo), G.mul(zk.ceremony.dense_multiexp(bases, hi), _limbs(1 << 240)))
    assert np.array_equal(G.to_affine(lhs), G.to_affine(rhs))

    # (3) the host-buffer form over three logical devices: the same records
    if trusted == 0:
        h_bases, h_cf = _host(bases), _host(cf)
        h_rp32, h_col32 = rp32.cpu().numpy().view(np.uint32), col32.cpu().numpy().view(np.uint32)
        zk.Worker(devices=[0, 0, 0])
        try:
            got = zk.ceremony.eval_qap_host(h_bases, h_rp32, h_col32, h_cf)
        finally:
            zk.Worker(0)
        assert np.array_equal(got, _host(out))
