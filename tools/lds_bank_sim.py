#!/usr/bin/env python3
"""LDS bank-conflict model of gfx950 (rules: /opt/skills/guides/MI355X_MICROARCH.md §LDS) for the GEMM tile
images of csrc/gemm.hip.  Prints the worst-case LDS cycles per wave-instruction vs the conflict-free count."""
import itertools

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
               list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALVES = [list(range(0, 32)), list(range(32, 64))]
W128_GROUPS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]
W64_GROUPS = [list(range(16 * g, 16 * g + 16)) for g in range(4)]


def cycles(addr_fn, nbytes, groups, modulus):
    """sum over lane groups of the max number of DISTINCT addresses landing on one bank"""
    total = 0
    for grp in groups:
        banks = {}
        for lane in grp:
            a = addr_fn(lane)
            for w in range(nbytes // 4):
                bank = ((a // 4) + w) % modulus
                banks.setdefault(bank, set()).add((a // 4) + w)
        total += max(len(v) for v in banks.values())
    return total


def report(name, addr_fn, nbytes, groups, modulus):
    c = cycles(addr_fn, nbytes, groups, modulus)
    print(f"{name:58s} {c:3d} cycles (conflict-free {len(groups)})")
    return c


def main(kc_stride_b=144, ks_stride_b=288):
    print(f"KC row stride {kc_stride_b} B, KS row stride {ks_stride_b} B")
    for ks in (0, 1):
        report(f"bf16 KC frag ds_read_b128 (ks={ks})",
               lambda l: (l & 15) * kc_stride_b + ks * 64 + 16 * (l >> 4), 16, B128_GROUPS, 64)
    def tr(l, hi=0):
        q, pp = (l & 15) >> 2, l & 3
        kr = 8 * (l >> 4) + q + 4 * hi
        return kr * ks_stride_b + 8 * pp
    report("bf16 KS frag ds_read_b64_tr_b16 (lo)", lambda l: tr(l, 0), 8, HALVES, 64)
    report("bf16 KS frag ds_read_b64_tr_b16 (hi)", lambda l: tr(l, 1), 8, HALVES, 64)
    # staging writes: thread tid -> piece p = tid (q = 0); one wave = 64 consecutive pieces
    for wave in (0, 1):
        report(f"bf16 KC stage ds_write_b128 (wave {wave})",
               lambda l: ((wave * 64 + l) >> 3) * kc_stride_b + ((wave * 64 + l) & 7) * 16, 16, W128_GROUPS, 32)
        report(f"bf16 KS stage ds_write_b128 (wave {wave})",
               lambda l: ((wave * 64 + l) // 16) * ks_stride_b + ((wave * 64 + l) % 16) * 16, 16, W128_GROUPS, 32)
    # fp32
    report("f32 KC frag ds_read_b32", lambda l: ((l & 15) * 34 + (l >> 4)) * 4, 4, HALVES, 32)
    report("f32 KS frag ds_read_b32", lambda l: ((l >> 4) * 144 + (l & 15)) * 4, 4, HALVES, 32)


if __name__ == "__main__":
    import sys
    main(*(int(a) for a in sys.argv[1:]))
