"""The producer side of the phase-2 ceremony, restated over the device library: the circom circuit reader and
`MPCParameters::new` -- the step that turns a constraint system and a `phase1radix2m{k}` file into the initial Groth16
parameters (the real "G1 + G2 mix" of BASELINE config 5).

Mirrored interfaces (same names, argument meaning and order of operations):
  phase2/src/circom_circuit.rs:46-56,332-360   CircuitJson / circuit_from_json: circuit.json of circom -> CircomCircuit
  phase2/src/circom_circuit.rs:319-331         witness_from_json
  phase2/src/circom_circuit.rs:135-185         Circuit::synthesize for CircomCircuit (variable 0 is the constant ONE)
  phase2/src/keypair_assembly.rs:15-25,70-105  KeypairAssembly: at / bt / ct, one (coefficient, constraint) list per variable
  phase2/src/parameters.rs:99-145              MPCParameters::new: the ONE input, synthesis, the `x * 0 = 0` input constraints,
                                               the domain size
  phase2/src/parameters.rs:225-400             eval (-> ceremony.eval_qap_polynomials: four sparse matrix x point-vector products
                                               on the device), UnconstrainedVariable, vk, the optional infinity filter, cs_hash

Host work here is what it is in the reference (JSON parsing, building the term lists); the group arithmetic -- one scalar
multiplication per (variable, constraint) term in G1 and, for the B polynomials, in G2 as well -- runs on the GPU.
"""
from __future__ import annotations

import json

import numpy as np

from . import ceremony
from .bellman import SynthesisError

_R_ORDER = ceremony._R_ORDER


def _fr_from_str(s: str) -> int:
    """PrimeField::from_str (ff_ce): decimal digits only, no sign, no leading zeros except "0" itself, value < r"""
    if not s or not s.isdigit() or (len(s) > 1 and s[0] == "0"):
        raise ValueError(f"not a field element: {s!r}")
    v = int(s)
    if v >= _R_ORDER:
        raise ValueError(f"not a field element: {s!r}")
    return v


class CircomCircuit:
    """circom_circuit.rs:99-111.  constraints: list of (A, B, C), each a list of (variable index, coefficient) with variable 0 the
    constant ONE, variables 1 .. num_inputs-1 the outputs and public inputs, the rest auxiliary."""

    def __init__(self, num_inputs: int, num_aux: int, num_constraints: int, constraints, witness=None):
        self.num_inputs, self.num_aux, self.num_constraints = num_inputs, num_aux, num_constraints
        self.constraints, self.witness = constraints, witness

    def get_public_inputs(self):
        return None if self.witness is None else list(self.witness[1:self.num_inputs])


def circuit_from_json(text) -> CircomCircuit:
    """circuit_from_json (circom_circuit.rs:340-360).  text: the JSON text (str / bytes) or an already parsed dict."""
    cj = text if isinstance(text, dict) else json.loads(text)
    num_inputs = int(cj["nPubInputs"]) + int(cj["nOutputs"]) + 1
    num_variables = int(cj["nVars"])
    num_aux = num_variables - num_inputs
    if num_aux < 0:
        raise ValueError("nVars < nPubInputs + nOutputs + 1")

    def index(k) -> int:
        # Rust's `parse::<usize>()` (circom_circuit.rs:346-350, unwrap): ASCII digits with an optional leading '+', nothing else
        # -- Python's int() would also take "-1", " 2" or "1_0", and a negative index would alias the LAST variable.  A variable
        # index past nVars panics in the reference when the constraint is synthesized; here it is rejected up front.
        digits = k[1:] if isinstance(k, str) and k.startswith("+") else k
        if not (isinstance(digits, str) and digits.isascii() and digits.isdigit()):
            raise ValueError(f"circuit.json: {k!r} is not a variable index")
        idx = int(digits)
        if idx >= num_variables:
            raise ValueError(f"circuit.json: variable index {idx} out of range (nVars = {num_variables})")
        return idx

    def convert(lc):   # a BTreeMap<String, String>: iterated in the order of the KEYS AS STRINGS
        return [(index(k), _fr_from_str(lc[k])) for k in sorted(lc.keys())]

    constraints = [(convert(c[0]), convert(c[1]), convert(c[2])) for c in cj["constraints"]]
    return CircomCircuit(num_inputs, num_aux, num_variables, constraints)


def circuit_from_json_file(path: str) -> CircomCircuit:
    with open(path, "rb") as f:
        return circuit_from_json(f.read())


def witness_from_json(text):
    return [_fr_from_str(x) for x in (text if isinstance(text, list) else json.loads(text))]


class KeypairAssembly:
    """keypair_assembly.rs:15-25: for every variable the (coefficient, constraint index) terms of its A, B and C polynomials, in
    the order the constraints were enforced.  Inputs first (index 0 = ONE), then the auxiliary variables."""

    def __init__(self):
        self.num_inputs = self.num_aux = self.num_constraints = 0
        self.at_inputs, self.bt_inputs, self.ct_inputs = [], [], []
        self.at_aux, self.bt_aux, self.ct_aux = [], [], []

    def alloc_input(self) -> int:
        self.num_inputs += 1
        for lst in (self.at_inputs, self.bt_inputs, self.ct_inputs):
            lst.append([])
        return self.num_inputs - 1

    def alloc(self) -> int:
        self.num_aux += 1
        for lst in (self.at_aux, self.bt_aux, self.ct_aux):
            lst.append([])
        return self.num_aux - 1

    def enforce(self, a, b, c):
        """a, b, c: lists of (is_input, index, coefficient)"""
        for lc, inputs, aux in ((a, self.at_inputs, self.at_aux), (b, self.bt_inputs, self.bt_aux), (c, self.ct_inputs, self.ct_aux)):
            for is_input, idx, coeff in lc:
                (inputs if is_input else aux)[idx].append((coeff, self.num_constraints))
        self.num_constraints += 1


def synthesize(circuit: CircomCircuit, cs: KeypairAssembly):
    """Circuit::synthesize (circom_circuit.rs:135-185) into a KeypairAssembly whose ONE input is already allocated."""
    for _ in range(1, circuit.num_inputs):
        cs.alloc_input()
    for _ in range(circuit.num_aux):
        cs.alloc()

    def lc(terms):
        # LinearCombination `+` appends; variable index < num_inputs is an input (0 = ONE), the rest auxiliary
        return [(idx < circuit.num_inputs, idx if idx < circuit.num_inputs else idx - circuit.num_inputs, coeff) for idx, coeff in terms]

    for a, b, c in circuit.constraints:
        cs.enforce(lc(a), lc(b), lc(c))


def _csr(rows, device):
    """per-variable term lists -> (row_ptr int32, col int32 = constraint / Lagrange index, coeff (nnz, 4) canonical limbs)"""
    import torch

    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(lens, out=row_ptr[1:])
    nnz = int(row_ptr[-1])
    col = np.empty(nnz, dtype=np.int32)
    coeff = np.empty((nnz, 4), dtype=np.uint64)
    t = 0
    mask = (1 << 64) - 1
    for r in rows:
        for c, lag in r:
            col[t] = lag
            coeff[t, 0], coeff[t, 1], coeff[t, 2], coeff[t, 3] = c & mask, (c >> 64) & mask, (c >> 128) & mask, c >> 192
            t += 1
    return (torch.from_numpy(row_ptr.astype(np.int32)).to(device), torch.from_numpy(col).to(device),
            torch.from_numpy(coeff.view(np.int64)).to(device))


def domain_exponent(num_constraints: int) -> int:
    """parameters.rs:134-145: the power of two the evaluation domain needs"""
    m, exp = 1, 0
    while m < num_constraints:
        m *= 2
        exp += 1
        if exp > 28:
            raise SynthesisError(SynthesisError.POLYNOMIAL_DEGREE_TOO_LARGE)
    return exp


def assemble(circuit: CircomCircuit) -> KeypairAssembly:
    """parameters.rs:106-132: the ONE input, the circuit, and one `x * 0 = 0` constraint per input (full density of the IC query)"""
    cs = KeypairAssembly()
    cs.alloc_input()
    synthesize(circuit, cs)
    for i in range(cs.num_inputs):
        cs.enforce([(True, i, 1)], [], [])
    return cs


def mpc_parameters_new(circuit: CircomCircuit, should_filter_points_at_infinity: bool, radix):
    """MPCParameters::new (phase2/src/parameters.rs:99-400).  radix: what ceremony.read_phase1radix2m returned for
    phase1radix2m{domain_exponent(...)} (device records).  Returns the dict ceremony.write_mpc_parameters takes:
    {"params": {"vk", "h", "l", "a", "b_g1", "b_g2"}, "cs_hash": 64 bytes (device uint8), "contributions": []}."""
    import torch

    cs = assemble(circuit)
    m = 1 << domain_exponent(cs.num_constraints)
    if radix["coeffs_g1"].shape[0] != m:
        raise ValueError(f"the phase1radix2m file is for a domain of {radix['coeffs_g1'].shape[0]}, the circuit needs {m}")
    dev = radix["coeffs_g1"].device
    at = _csr(cs.at_inputs + cs.at_aux, dev)
    bt = _csr(cs.bt_inputs + cs.bt_aux, dev)
    ct = _csr(cs.ct_inputs + cs.ct_aux, dev)
    a_g1, b_g1, b_g2, ext = ceremony.eval_qap_polynomials(radix, at, bt, ct)
    ic, l = ext[:cs.num_inputs], ext[cs.num_inputs:]  # noqa: E741
    # "Don't allow any elements be unconstrained, so that the L query is always fully dense" (parameters.rs:340-346)
    if l.shape[0] and bool((l == 0).all(dim=1).any().item()):
        raise SynthesisError(SynthesisError.UNCONSTRAINED_VARIABLE)
    one1 = torch.from_numpy(ceremony.G1_ONE_RAW.view(np.int64).reshape(1, 8)).to(dev)
    one2 = torch.from_numpy(ceremony.G2_ONE_RAW.view(np.int64).reshape(1, 16)).to(dev)
    vk = {"alpha_g1": radix["alpha_g1"], "beta_g1": radix["beta_g1"], "beta_g2": radix["beta_g2"], "gamma_g2": one2, "delta_g1": one1,
          "delta_g2": one2, "ic": ic.contiguous()}
    if should_filter_points_at_infinity:
        keep = lambda pts: pts[~(pts == 0).all(dim=1)].contiguous()  # noqa: E731
        a_g1, b_g1, b_g2 = keep(a_g1), keep(b_g1), keep(b_g2)
    params = {"vk": vk, "h": radix["h"], "l": l.contiguous(), "a": a_g1, "b_g1": b_g1, "b_g2": b_g2}
    digest = ceremony.calculate_hash(ceremony.write_parameters(params))     # HashWriter over Parameters::write (parameters.rs:382-392)
    cs_hash = torch.frombuffer(bytearray(digest), dtype=torch.uint8).to(dev)
    return {"params": params, "cs_hash": cs_hash, "contributions": []}


# ---------------------------------------------------------------------------------------------------------------
# proving a circom circuit (circom_circuit.rs:187-191 `prove` = filter_params + create_random_proof; bellman's prepare_prover,
# groth16/prover.rs:49-200)
def prepare_prover(circuit: CircomCircuit, device):
    """prepare_prover (groth16/prover.rs:153-187): the ONE input, the circuit, one `x * 0 = 0` constraint per input, with the
    witness: per constraint the values of its A, B and C combinations, the assignments, and the three density maps (a variable
    counts as dense in a query as soon as it OCCURS in a combination, whatever its coefficient: prover.rs:49-88).  Returns the
    prover.ProvingAssignment (Montgomery Fr on the device, like Vec<Scalar<E>> / Vec<E::Fr>)."""
    import torch

    from . import prover as _prover
    from .bellman import DensityTracker

    if circuit.witness is None:
        raise SynthesisError("AssignmentMissing")
    w = circuit.witness
    if len(w) != circuit.num_inputs + circuit.num_aux:
        raise ValueError("the witness does not have one value per variable")
    num_inputs, num_aux = circuit.num_inputs, circuit.num_aux
    a_aux, b_in, b_aux = np.zeros(num_aux, dtype=bool), np.zeros(num_inputs, dtype=bool), np.zeros(num_aux, dtype=bool)
    a, b, c = [], [], []
    for ca, cb, cc in circuit.constraints:
        acc = 0
        for idx, coeff in ca:
            acc += coeff * w[idx]
            if idx >= num_inputs:
                a_aux[idx - num_inputs] = True
        a.append(acc % _R_ORDER)
        acc = 0
        for idx, coeff in cb:
            acc += coeff * w[idx]
            if idx >= num_inputs:
                b_aux[idx - num_inputs] = True
            else:
                b_in[idx] = True
        b.append(acc % _R_ORDER)
        c.append(sum(coeff * w[idx] for idx, coeff in cc) % _R_ORDER)
    for i in range(num_inputs):                      # x_i * 0 = 0
        a.append(w[i] % _R_ORDER)
        b.append(0)
        c.append(0)

    mont_r = (1 << 256) % _R_ORDER
    mask = (1 << 64) - 1

    def mont(vals):
        arr = np.empty((len(vals), 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            v = v % _R_ORDER * mont_r % _R_ORDER
            arr[i, 0], arr[i, 1], arr[i, 2], arr[i, 3] = v & mask, (v >> 64) & mask, (v >> 128) & mask, v >> 192
        return torch.from_numpy(arr.view(np.int64)).to(device)

    return _prover.ProvingAssignment(mont(a), mont(b), mont(c), mont(w[:num_inputs]), mont(w[num_inputs:]), DensityTracker.from_bools(a_aux),
                                     DensityTracker.from_bools(b_in), DensityTracker.from_bools(b_aux))


def filter_params(params):
    """filter_params (circom_circuit.rs:271-277): vk.ic, h, a, b_g1 and b_g2 without their points at infinity (the A / B queries
    are what the density maps index; l is left as it is, as in the reference)."""
    keep = lambda pts: pts[~(pts == 0).all(dim=1)].contiguous()  # noqa: E731
    vk = dict(params["vk"], ic=keep(params["vk"]["ic"]))
    return dict(params, vk=vk, h=keep(params["h"]), a=keep(params["a"]), b_g1=keep(params["b_g1"]), b_g2=keep(params["b_g2"]))


def prove(pool, circuit: CircomCircuit, params, r: int, s: int):
    """prove (circom_circuit.rs:187-191) with the blinding scalars given (create_random_proof draws them from the RNG):
    returns the proof (a, b, c) as raw affine records.  params: the "params" dict of mpc_parameters_new / read_mpc_parameters."""
    from . import prover as _prover

    p = filter_params(params)
    host = lambda t: t.cpu().numpy().view(np.uint64).reshape(-1)  # noqa: E731
    vk = {k: host(p["vk"][k]) for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2")}
    assignment = prepare_prover(circuit, p["h"].device)
    return _prover.create_proof(pool, _prover.Parameters(vk, p["h"], p["l"], p["a"], p["b_g1"], p["b_g2"]), assignment, r, s)
