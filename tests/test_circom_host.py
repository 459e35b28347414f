"""The circom circuit reader and the keypair assembly (phase2-bn254_amd/circom.py = phase2/src/circom_circuit.rs:332-360,
135-185 and keypair_assembly.rs + the preamble of MPCParameters::new, parameters.rs:106-145): host logic only."""
import json

import pytest

import bn254_model as M


def _zk():
    import phase2_bn254_amd as zk

    return zk


CIRCUIT = {
    # x1 * x2 = x3;  (x3 + 5) * 1 = out;   variables: 0 = ONE, 1 = out (output), 2 = x1 (public input), 3 = x2, 4 = x3 (aux)
    "constraints": [[{"2": "1"}, {"3": "1"}, {"4": "1"}],
                    [{"4": "1", "0": "5"}, {"0": "1"}, {"1": "1"}],
                    [{"10": "7", "9": "3"}, {}, {}]],                    # keys sort as STRINGS ("10" < "9"); variables 9, 10 exist below
    "nPubInputs": 1, "nOutputs": 1, "nVars": 11,
}


def test_circuit_from_json_counts_and_terms():
    zk = _zk()
    c = zk.circom.circuit_from_json(json.dumps(CIRCUIT))
    assert (c.num_inputs, c.num_aux, c.num_constraints) == (3, 8, 11)       # num_constraints = nVars, as the reference sets it
    assert c.constraints[0] == ([(2, 1)], [(3, 1)], [(4, 1)])
    assert c.constraints[1][0] == [(0, 5), (4, 1)]
    assert c.constraints[2][0] == [(10, 7), (9, 3)]
    assert c.get_public_inputs() is None
    c.witness = zk.circom.witness_from_json('["1", "11", "2", "3", "6", "0", "0", "0", "0", "0", "0"]')
    assert c.get_public_inputs() == [11, 2]


@pytest.mark.parametrize("bad", ["", "-1", "01", "1e3", " 1", str(M.R_ORDER), "0x10"])
def test_field_element_strings_like_from_str(bad):
    zk = _zk()
    cj = dict(CIRCUIT, constraints=[[{"1": bad}, {}, {}]])
    with pytest.raises(ValueError):
        zk.circom.circuit_from_json(cj)
    assert zk.circom.circuit_from_json(dict(CIRCUIT, constraints=[[{"1": str(M.R_ORDER - 1)}, {"1": "0"}, {}]])).constraints[0][0] == [(1, M.R_ORDER - 1)]


@pytest.mark.parametrize("bad", ["-1", " 2", "1_0", "2 ", "", "0x2", "11", "99999999999999999999", "\u0662"])
def test_variable_indices_like_parse_usize(bad):
    """`s.parse::<usize>()` on the keys (circom_circuit.rs:346-350): no sign but '+', no blanks, no underscores, ASCII digits
    only; an index past nVars (11 here) never aliases another variable."""
    zk = _zk()
    with pytest.raises(ValueError):
        zk.circom.circuit_from_json(dict(CIRCUIT, constraints=[[{bad: "1"}, {}, {}]]))
    ok = zk.circom.circuit_from_json(dict(CIRCUIT, constraints=[[{"+3": "1", "10": "2", "007": "5"}, {}, {}]]))
    assert ok.constraints[0][0] == [(3, 1), (7, 5), (10, 2)]           # keys in string order: "+3" < "007" < "10"


def test_assembly_like_mpc_parameters_new():
    zk = _zk()
    cs = zk.circom.assemble(zk.circom.circuit_from_json(CIRCUIT))
    assert (cs.num_inputs, cs.num_aux) == (3, 8)
    assert cs.num_constraints == 3 + 3                                       # the circuit's, then one  x * 0 = 0  per input
    assert cs.at_inputs == [[(5, 1), (1, 3)], [(1, 4)], [(1, 0), (1, 5)]]     # ONE: 5 in constraint 1, then its own input constraint
    assert cs.bt_inputs == [[(1, 1)], [], []] and cs.ct_inputs == [[], [(1, 1)], []]
    assert cs.at_aux[:2] == [[], [(1, 1)]] and cs.bt_aux[0] == [(1, 0)] and cs.ct_aux[1] == [(1, 0)]
    assert cs.at_aux[6] == [(3, 2)] and cs.at_aux[7] == [(7, 2)]             # variables 9 and 10
    assert zk.circom.domain_exponent(cs.num_constraints) == 3
    assert zk.circom.domain_exponent(1) == 0 and zk.circom.domain_exponent(8) == 3 and zk.circom.domain_exponent(9) == 4
    with pytest.raises(zk.SynthesisError):
        zk.circom.domain_exponent((1 << 28) + 1)
