#!/usr/bin/env python3
"""Bank model of the LDS accesses of ntt_pass_kernel (ntt.hip): which access sites conflict, and by how much -- the "why" behind
SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 20 % on the 2^20 pass (profiles/r04_final_ntt20_pass_sq_pmc.txt: 0.885 M of 4.42 M cycles) with an
element-major layout whose odd 9-word stride was called conflict-free.  The model reproduces the counter: 20.0 % for round 4's layout.

Model: 32 banks of 4 bytes; a wave's ds_read2_b32 / ds_write2_b32 (two dwords of an element per lane) is served dword by dword, 32 lanes at a time; a
group of 32 lane-dwords takes as many cycles as its most loaded bank has DISTINCT addresses.  word address = 9 * position(g, x) + limb.

  round 4:  position(g, x) = g * (np + 1) + (x ^ ((x >> 5) & 31))
            -- two sites 2-way conflicted: the stage pair with m = 16 (lanes 16 apart in a wave hold x and x + 64: the xor term moves x + 64 by 2, inside
               the same 16 banks) and the closing read of a G >= 2 tile (lanes alternate rows: +9 banks per row against +9 per element)
  round 5:  position(g, x) = g * np + ((x ^ ((((x >> 5) * 25) ^ (x >> 10)) & 31)) ^ ((g * a) & 31)),  a = 21 (20 for eight-row tiles)
            -- the multiplier sends x + 32, + 64, + 128 ... each to another part of the banks, bits 10 and 11 of the long rows are folded in, the
               row term separates neighbouring rows where lanes alternate rows; found by searching multipliers and row terms over every tile shape
               the planner uses (table below: what is left is the eight-row tile of 2^24 and the short radix-2 rows)

  round 5, wave-local kernel (ntt_pass_wl_kernel: full tiles, lane q = row q mod G, group q / G in every stage pair):
            position(g, x) = g * np + ((x ^ ((((x >> 5) * 13) ^ (x >> 10)) & 31)) ^ ((g * a_G) & 31)),  a_2 = 26, a_4 = 21, a_8 = 25
            -- chosen by the same search for THAT lane order: in a wave's 32-lane group the varying index bits are (row, a few bits of the group
               index); multiplier and row term must make them independent modulo 32 banks in every stage pair (second table below)

usage: python tools/ntt_lds_model.py            (all tile shapes, both layouts; then the wave-local kernel's shapes)"""
import sys

def brev(x, bits): return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0

LAYOUTS = {
    "round 4": (lambda np: np + 1 if np >= 32 else np, lambda g, x: x ^ ((x >> 5) & 31)),
    "round 5": (lambda np: np, None),   # (the row term depends on the number of rows: filled in per shape)
}
def round5(G):
    a = 20 if G == 8 else 21
    return lambda g, x: (x ^ ((((x >> 5) * 25) ^ (x >> 10)) & 31)) ^ ((g * a) & 31)

def model(log_np, G, threads, r4, layout, load_x_fastest, verbose=False):
    np_ = 1 << log_np
    pitch_fn, sw = LAYOUTS[layout]
    if sw is None: sw = round5(G)
    pitch = pitch_fn(np_)
    pos = lambda g, x: g * pitch + sw(g, x)
    def cost(idx, l):
        c = 0
        for half in range(0, len(idx), 32):
            banks = {}
            for lane in range(half, min(half + 32, len(idx))):
                a = 9 * idx[lane] + l
                banks.setdefault(a & 31, set()).add(a)
            c += max(len(v) for v in banks.values())
        return c
    sites = []
    def site(name, idx_fn, count, weight):
        ideal = cyc = 0
        for w0 in range(0, count, 64):
            idx = [idx_fn(w0 + i) for i in range(min(64, count - w0))]
            for l in range(9):
                cyc += weight * cost(idx, l)
                ideal += weight * ((len(idx) + 31) // 32)
        sites.append((name, ideal, cyc))
    elems = G * np_
    def gx(e):
        return (e >> log_np, e & (np_ - 1)) if load_x_fastest else (e % G, e // G)
    site("load -> LDS write", lambda e: pos(gx(e)[0], brev(gx(e)[1], log_np)), elems, 1)
    if r4:
        s0 = log_np & 1
        if s0:
            site("stage 0", lambda b: pos(b >> (log_np - 1), ((b & ((np_ >> 1) - 1)) << 1)), elems // 2, 2)
            site("stage 0 (+1)", lambda b: pos(b >> (log_np - 1), ((b & ((np_ >> 1) - 1)) << 1) + 1), elems // 2, 2)
        for s in range(s0, log_np, 2):
            m = 1 << s
            j_slow = s in (1, 2) and (np_ >> (s + 2)) >= 64
            def x0_of(q, s=s, m=m, j_slow=j_slow):
                qf = q & ((np_ >> 2) - 1)
                if j_slow:
                    bl = log_np - 2 - s
                    return ((qf & ((1 << bl) - 1)) << (s + 2)) + (qf >> bl)
                return ((qf >> s) << (s + 2)) + (qf & (m - 1))
            for r in range(4):
                site(f"stages {s},{s + 1} element {'abcd'[r]}", lambda q, r=r, m=m, x0_of=x0_of: pos(q >> (log_np - 2), x0_of(q) + r * m), elems // 4, 2)
    else:
        skip_max = 1 if log_np > 10 else 2
        for s in range(log_np):
            m = 1 << s
            j_slow = 1 <= s <= skip_max and (np_ >> (s + 1)) >= 64
            def x0_of(b, s=s, m=m, j_slow=j_slow):
                bf = b & ((np_ >> 1) - 1)
                if j_slow:
                    bl = log_np - 1 - s
                    return ((bf & ((1 << bl) - 1)) << (s + 1)) + (bf >> bl)
                return ((bf >> s) << (s + 1)) + (bf & (m - 1))
            for r in range(2):
                site(f"stage {s} element {'ut'[r]}", lambda b, r=r, m=m, x0_of=x0_of: pos(b >> (log_np - 1), x0_of(b) + r * m), elems // 2, 2)
    site("LDS read -> store", lambda e: pos(e % G, e // G), elems, 1)
    ti = sum(s[1] for s in sites); tc = sum(s[2] for s in sites)
    if verbose:
        for name, ideal, cyc in sites:
            if cyc != ideal: print(f"      {name:28s} ideal {ideal:6d}  modelled {cyc:6d}")
    return ti, tc

# (log_np, G, threads, radix-4, load_x_fastest): the tile shapes ntt_run_scaled plans
SHAPES = [
    ("2^20 both passes: 2 x 2^10", 10, 2, 512, True, 0), ("2^20 last pass (x fastest)", 10, 2, 512, True, 1),
    ("2^16 .. 2^19: 4 x 2^10 / narrower", 10, 4, 1024, True, 0), ("2^18: 2^9 rows, G = 4", 9, 4, 512, True, 0), ("2^16: 2^8 rows, G = 1", 8, 1, 128, False, 1),
    ("2^21 / 2^22: 2^11 rows, G = 1", 11, 1, 512, True, 1), ("2^23: 2^12 rows", 12, 1, 1024, True, 1),
    ("2^24: 8 x 2^8", 8, 8, 512, True, 0), ("2^25 / 2^26: 4 x 2^9", 9, 4, 512, True, 0), ("2^14: 2^7 rows, G = 1", 7, 1, 64, False, 1),
    ("2^12: 2^6 rows, G = 1", 6, 1, 64, False, 1),
]
for name, log_np, G, threads, r4, lxf in SHAPES:
    out = []
    for layout in LAYOUTS:
        ti, tc = model(log_np, G, threads, r4, layout, lxf)
        out.append(f"{layout}: {100 * (tc - ti) / tc:5.1f} % conflict cycles")
    print(f"{name:36s} {'radix-4' if r4 else 'radix-2'}   " + "   ".join(out))
    if "-v" in sys.argv:
        for layout in LAYOUTS:
            print(f"   {layout}:"); model(log_np, G, threads, r4, layout, lxf, verbose=True)


# ---- the wave-local kernel (ntt_pass_wl_kernel)
def model_wl(log_np, G, column_pass):
    np_ = 1 << log_np
    threads = G * np_ // 4
    a = {1: 0, 2: 26, 4: 21, 8: 25}[G]
    pos = lambda g, x: g * np_ + ((x ^ ((((x >> 5) * 13) ^ (x >> 10)) & 31)) ^ ((g * a) & 31))
    def cost(idx, l):
        c = 0
        for half in range(0, len(idx), 32):
            banks = {}
            for lane in range(half, half + 32):
                ad = 9 * idx[lane] + l
                banks.setdefault(ad & 31, set()).add(ad)
            c += max(len(v) for v in banks.values())
        return c
    ideal = cyc = 0
    def site(idx_fn, count, weight):
        nonlocal ideal, cyc
        for w0 in range(0, count, 64):
            idx = [idx_fn(w0 + i) for i in range(64)]
            for l in range(9):
                cyc += weight * cost(idx, l); ideal += weight * 2
    s0 = log_np & 1
    direct = column_pass and not s0
    if not direct:   # tile load through LDS (last pass: x fastest; odd row lengths)
        site((lambda e: pos(e >> log_np, brev(e & (np_ - 1), log_np))) if not column_pass else (lambda e: pos(e % G, brev(e // G, log_np))), G * np_, 1)
    if s0:
        for r in range(4): site(lambda q, r=r: pos(q % G, 4 * (q // G) + r), threads, 2)
    rounds = list(range(s0, log_np, 2))
    for s in rounds:
        m = 1 << s
        w = 2
        if s == rounds[0] and direct: w -= 1    # first pair: operands come from memory
        if s == rounds[-1]: w -= 1              # last pair: results go to memory
        for r in range(4):
            site(lambda q, r=r, s=s, m=m: pos(q % G, (((q // G) >> s) << (s + 2)) + ((q // G) & (m - 1)) + r * m), threads, w)
    return ideal, cyc
print()
for name, log_np, G in (("2^20: 2 x 2^10", 10, 2), ("2^16 .. 2^19: 4 x 2^10", 10, 4), ("2^21 / 2^22: 2^11 rows", 11, 1), ("2^23: 2^12 rows", 12, 1), ("2^24: 8 x 2^8", 8, 8),
                        ("2^25 / 2^26: 4 x 2^9", 9, 4)):
    out = []
    for col in (1, 0):
        ti, tc = model_wl(log_np, G, col)
        out.append(f"{'column pass' if col else 'last pass'}: {100 * (tc - ti) / tc:5.1f} %")
    print(f"{name:28s} wave-local kernel   " + "   ".join(out))
