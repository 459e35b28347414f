#!/usr/bin/env python3
"""Regenerates integration/bellman_mi355zk.patch: copies /root/reference/bellman twice, applies the edits below to one copy (and drops
integration/mi355zk.rs in as src/mi355zk.rs), and writes `diff -urN -U2` of the two trees with the timestamps stripped.  Needs the reference
tree, so it runs in the build container only; tests/test_integration_patch.py checks the committed patch still applies (`patch --dry-run`).
The edits are the whole Rust-side change a maintainer reviews: one feature line, one `mod`, two defaulted trait methods with their
implementations, and one early return in each of `multiexp` and `best_fft`."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/bellman"

EDITS = [
    ("Cargo.toml", 'nightly = ["prefetch"]\n',
     'nightly = ["prefetch"]\n# multiexp / best_fft of BN254 on an AMD MI355X through libmi355zk.so (src/mi355zk.rs); link with RUSTFLAGS="-L <dir of libmi355zk.so>"\n'
     'mi355zk = []\n'),
    ("src/lib.rs", "mod group;\nmod source;\nmod multiexp;\n", 'mod group;\nmod source;\nmod multiexp;\n\n#[cfg(feature = "mi355zk")]\npub mod mi355zk;\n'),
    ("src/source.rs", "    fn new(self) -> Self::Source;\n}\n",
     "    fn new(self) -> Self::Source;\n\n"
     "    /// The bases as one slice plus the cursor's start, when the source is one (mi355zk: what crosses the FFI boundary).\n"
     "    fn as_contiguous(&self) -> Option<(&[G], usize)> {\n        None\n    }\n}\n"),
    ("src/source.rs", "    fn new(self) -> (Arc<Vec<G>>, usize) {\n        (self.0.clone(), self.1)\n    }\n",
     "    fn new(self) -> (Arc<Vec<G>>, usize) {\n        (self.0.clone(), self.1)\n    }\n\n"
     "    fn as_contiguous(&self) -> Option<(&[G], usize)> {\n        Some((&self.0[..], self.1))\n    }\n"),
    ("src/source.rs", "    fn iter(self) -> Self::Iter;\n    fn get_query_size(self) -> Option<usize>;\n}\n",
     "    fn iter(self) -> Self::Iter;\n    fn get_query_size(self) -> Option<usize>;\n\n"
     "    /// The map as (u32 words: bit i = word i / 32, bit i % 32; number of bits); no words = every exponent has a base (mi355zk).\n"
     "    fn density_words(self) -> Option<(Vec<u32>, usize)> where Self: Sized {\n        None\n    }\n}\n"),
    ("src/source.rs", "    fn get_query_size(self) -> Option<usize> {\n        None\n    }\n}\n",
     "    fn get_query_size(self) -> Option<usize> {\n        None\n    }\n\n"
     "    fn density_words(self) -> Option<(Vec<u32>, usize)> {\n        Some((Vec::new(), 0))\n    }\n}\n"),
    ("src/source.rs", "    fn get_query_size(self) -> Option<usize> {\n        Some(self.bv.len())\n    }\n}\n",
     "    fn get_query_size(self) -> Option<usize> {\n        Some(self.bv.len())\n    }\n\n"
     "    fn density_words(self) -> Option<(Vec<u32>, usize)> {\n"
     "        // built from the iterator (bit-vec's block storage is private to that crate); one spare word, so a tracker is never \"no words\"\n"
     "        let mut words = vec![0u32; (self.bv.len() + 31) / 32 + 1];\n"
     "        for (i, b) in self.bv.iter().enumerate() {\n            if b {\n                words[i / 32] |= 1u32 << (i % 32);\n            }\n        }\n"
     "        Some((words, self.bv.len()))\n    }\n}\n"),
    ("src/multiexp.rs", "        assert!(query_size == exponents.len());\n    }\n\n    multiexp_inner_impl(pool, bases, density_map, exponents, 0, c, true)\n",
     "        assert!(query_size == exponents.len());\n    }\n\n"
     "    // BN254 G1 / G2 over a contiguous source: the whole multiexp on the MI355X; anything else, and any device failure, continues below\n"
     "    #[cfg(feature = \"mi355zk\")]\n    {\n"
     "        if let Some(done) = crate::mi355zk::try_multiexp::<Q, D, G, S>(&bases, &density_map, &exponents) {\n            return done;\n        }\n    }\n\n"
     "    multiexp_inner_impl(pool, bases, density_map, exponents, 0, c, true)\n"),
    ("src/domain.rs", "pub(crate) fn best_fft<E: Engine, T: Group<E>>(a: &mut [T], worker: &Worker, omega: &E::Fr, log_n: u32)\n{\n",
     "pub(crate) fn best_fft<E: Engine, T: Group<E>>(a: &mut [T], worker: &Worker, omega: &E::Fr, log_n: u32)\n{\n"
     "    // Scalar<Bn256>: in place on the MI355X (the array is written only on success); Point<G>, other engines and device failures continue below\n"
     "    #[cfg(feature = \"mi355zk\")]\n    {\n        if crate::mi355zk::try_best_fft::<E, T>(a, omega, log_n) {\n            return;\n        }\n    }\n\n"),
]


def main() -> int:
    tmp = tempfile.mkdtemp()
    try:
        for side in ("a", "b"):
            shutil.copytree(REF, os.path.join(tmp, side))
        b = os.path.join(tmp, "b")
        shutil.copy(os.path.join(HERE, "mi355zk.rs"), os.path.join(b, "src", "mi355zk.rs"))
        for rel, old, new in EDITS:
            path = os.path.join(b, rel)
            s = open(path).read()
            assert s.count(old) == 1, "%s: expected exactly one occurrence of %r" % (rel, old)
            open(path, "w").write(s.replace(old, new))
        out = subprocess.run(["diff", "-urN", "-U2", "a", "b"], cwd=tmp, capture_output=True, text=True).stdout
        lines = []
        for ln in out.splitlines():
            if ln.startswith(("--- ", "+++ ")):
                ln = ln.split("\t")[0]
            lines.append(ln)
        open(os.path.join(HERE, "bellman_mi355zk.patch"), "w").write("\n".join(lines) + "\n")
    finally:
        shutil.rmtree(tmp)
    return 0


if __name__ == "__main__":
    sys.exit(main())
