"""BASELINE config 1, the `new_constrained` half, on the CPU: the challenge file of a fresh ceremony is fully determined by
the reference's layout and literals -- 64-byte blank BLAKE2b hash, then every element = the group generator, uncompressed
(powersoftau/src/batched_accumulator.rs:87-178, 1295-1347; src/bin/new_constrained.rs:42-77; sizes parameters.rs:74-107) --
so its BLAKE2b-512 (utils.rs:20-27) is a reference-layout-anchored BN254 fixture.  Here the ORACLE's encoder
(oracle/codec.h) and the generator literals of pairing/src/bn256/fq.rs:39-83 are pinned on it; tests/test_gpu_ceremony.py
pins the HIP encoder on the same values."""
import hashlib

import numpy as np
import pytest

import oracle_lib as O

# power -> (challenge bytes, BLAKE2b-512 of the file)
CHALLENGE = {
    12: (1_572_992, "9e63a5f62b96538daaed2372481920d1a40b91959ea38ef9f5f6a3033b8865160710d067c09d09615f928ea517bcdf49ad75abd2c8340b400e3b18e968b4ffef"),
    10: (393_344, "95f0b4499e50f8da383b0d74c174c1698bdffe1b35066754005889a147849bbf8d64ff6c989bd89a4736b569a99a1c83a50dc181e9fe1d4d23d1888a98b3157e"),
}
BLANK = "786a02f742015903c6c6fd852552d272912f4740e15847618a86e217f71f5419d25e1031afee585313896444934eb04b903a685b1448b755d56f701afe9be2ce"


def test_blank_hash_and_sizes(zk):
    assert zk.ceremony.blank_hash().hex() == BLANK                        # utils.rs:138-140
    for power, (size, _) in CHALLENGE.items():
        _, total = zk.ceremony.accumulator_layout(power, compressed=False)
        assert total == size                                              # parameters.rs:83-89 accumulator_size
    _, total_c = zk.ceremony.accumulator_layout(12, compressed=True)
    assert total_c + 3 * 128 + 6 * 64 == 787_296                         # parameters.rs:97-107 contribution_size(12)


@pytest.mark.parametrize("power", [10, 12])
def test_new_constrained_challenge_hash_with_the_oracle_encoder(zk, power):
    n = 1 << power
    one1 = O.encode_points(1, zk.ceremony.G1_ONE_RAW.reshape(1, 8), False).reshape(-1)
    one2 = O.encode_points(2, zk.ceremony.G2_ONE_RAW.reshape(1, 16), False).reshape(-1)
    assert one1.tobytes() == (1).to_bytes(32, "big") + (2).to_bytes(32, "big")   # G1 one = BE(1) || BE(2), ec.rs:827-842
    blob = np.concatenate([np.frombuffer(zk.ceremony.blank_hash(), np.uint8), np.tile(one1, 2 * n - 1), np.tile(one2, n),
                           np.tile(one1, n), np.tile(one1, n), one2])
    size, digest = CHALLENGE[power]
    assert blob.size == size
    assert hashlib.blake2b(blob.tobytes(), digest_size=64).hexdigest() == digest
    assert zk.ceremony.calculate_hash(blob).hex() == digest
