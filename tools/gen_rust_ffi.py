#!/usr/bin/env python3
"""include/mi355zk.h -> the `extern "C"` block of integration/mi355zk.rs (between the GENERATED markers).

  python tools/gen_rust_ffi.py            rewrite the block in place
  python tools/gen_rust_ffi.py --check    exit 1 if the committed block differs from what the header gives (tests/test_integration_patch.py)

The Rust side of the boundary cannot be compiled in this image (no rustc); what CAN be checked is that every item of the block is derived
from the header mechanically -- same name, same argument count and order, C types mapped by the table below -- so the block is generated,
never typed.  Self-test and micro-benchmark hooks (mi355zk_selftest_*, mi355zk_ubench_*) are test infrastructure and stay out of it."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355zk.h")
TARGET = os.path.join(ROOT, "integration", "mi355zk.rs")
BEGIN, END = "// ---- BEGIN GENERATED (tools/gen_rust_ffi.py from include/mi355zk.h) ----", "// ---- END GENERATED ----"

SCALAR = {"int": "c_int", "long": "c_long", "long long": "c_longlong", "size_t": "usize", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64",
          "int32_t": "i32", "int8_t": "i8", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void"}


def prototypes(text=None):
    """[(return type, name, [(c type, name)])] of every function the header declares, in header order"""
    text = open(HEADER).read() if text is None else text
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = "\n".join(ln for ln in text.splitlines() if not ln.lstrip().startswith("#"))
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mi355zk_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        ret = ret.replace('extern "C"', "").strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"(.*?)(\w+)\s*(\[\d*\])?$", a)
                ctype, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
                params.append((ctype + (" " + arr if arr else ""), pname))
        out.append((ret, name, params))
    return out


def rust_type(ctype: str) -> str:
    arr = re.search(r"\[(\d*)\]$", ctype)
    if arr:
        base = ctype[:arr.start()].strip()
        const = base.startswith("const ")
        base = base[6:] if const else base
        return "%s %s /* [%s] */" % ("*const" if const else "*mut", SCALAR[base], arr.group(1))
    t = ctype.replace(" *", "*").replace("* ", "*").strip()
    if t == "void*const*":
        return "*const *mut c_void"
    stars = len(t) - len(t.rstrip("*"))
    base = t.rstrip("*").strip()
    const = base.startswith("const ")
    base = base[6:].strip() if const else base
    r = SCALAR[base]
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def block() -> str:
    lines = [BEGIN, '#[link(name = "mi355zk")]', 'extern "C" {']
    for ret, name, params in prototypes():
        if name.startswith(("mi355zk_selftest_", "mi355zk_ubench_")):
            continue
        args = ", ".join("%s: %s" % (pn if pn not in ("in", "type", "ref", "fn") else pn + "_", rust_type(ct)) for ct, pn in params)
        rt = "" if ret == "void" else " -> " + ("*const c_char" if ret == "const char *" else SCALAR[ret])
        lines.append("    pub fn %s(%s)%s;" % (name, args, rt))
    lines += ["}", END]
    return "\n".join(lines)


def main() -> int:
    src = open(TARGET).read()
    i, j = src.index(BEGIN), src.index(END) + len(END)
    new = src[:i] + block() + src[j:]
    if "--check" in sys.argv:
        if new != src:
            print("integration/mi355zk.rs: the generated extern block is stale -- run tools/gen_rust_ffi.py", file=sys.stderr)
            return 1
        return 0
    open(TARGET, "w").write(new)
    return 0


if __name__ == "__main__":
    sys.exit(main())
