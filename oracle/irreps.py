"""Tiny irreps bookkeeping for the oracle (independent of nequip_amd.o3).

Follows e3nn.o3.Irreps semantics as restated in SURVEY.md A.1: ``(mul, (l, p))`` entries, mul_ir layout,
``sort`` by tuple order ``(l, p)`` (so 0o < 0e < 1o < 1e ...), ``simplify`` merges adjacent equals.
"""

import re


def parse(s):
    """'4x0e + 3x1o + 2e' -> [(4, 0, 1), (3, 1, -1), (1, 2, 1)]"""
    if not isinstance(s, str):
        return [tuple(t) for t in s]
    out = []
    s = s.strip()
    if not s:
        return out
    for term in s.split("+"):
        term = term.strip()
        m = re.fullmatch(r"(?:(\d+)x)?(\d+)([eo])", term)
        assert m is not None, f"bad irreps term {term!r}"
        mul = int(m.group(1)) if m.group(1) else 1
        out.append((mul, int(m.group(2)), 1 if m.group(3) == "e" else -1))
    return out


def to_str(irreps):
    return "+".join(f"{mul}x{l}{'e' if p == 1 else 'o'}" for mul, l, p in irreps)


def dim(irreps):
    return sum(mul * (2 * l + 1) for mul, l, _ in irreps)


def num_irreps(irreps):
    return sum(mul for mul, _, _ in irreps)


def slices(irreps):
    out, i = [], 0
    for mul, l, _ in irreps:
        n = mul * (2 * l + 1)
        out.append(slice(i, i + n))
        i += n
    return out


def simplify(irreps):
    out = []
    for mul, l, p in irreps:
        if mul == 0:
            continue
        if out and out[-1][1:] == (l, p):
            out[-1] = (out[-1][0] + mul, l, p)
        else:
            out.append((mul, l, p))
    return out


def sort(irreps):
    """returns (sorted, p, inv) with p[i_old] = i_new (e3nn Irreps.sort)."""
    keyed = sorted(((l, p), i, mul) for i, (mul, l, p) in enumerate(irreps))
    inv = [i for _, i, _ in keyed]
    perm = [0] * len(inv)
    for new, old in enumerate(inv):
        perm[old] = new
    return [(mul, lp[0], lp[1]) for lp, _, mul in keyed], perm, inv


def product(ir1, ir2):
    """selection rule: (l, p1*p2) for |l1-l2| <= l <= l1+l2, ascending."""
    (l1, p1), (l2, p2) = ir1, ir2
    return [(l, p1 * p2) for l in range(abs(l1 - l2), l1 + l2 + 1)]


def contains(irreps, ir):
    return any((l, p) == tuple(ir) for _, l, p in irreps)


def spherical_harmonics(lmax):
    return [(1, l, (-1) ** l) for l in range(lmax + 1)]
