"""Reader of the GV ``.diag`` catalogs -> graphs.

Reference: src/frontend/GV.jl:77-93 (``diagsGV(type, order)``) and
src/frontend/GV_diagrams/readfile.jl:5-28 (regex int parsing, ``_exchange``),
:412-473 (``read_diagrams``), :475-588 (``read_one_diagram!``); leaf identity
src/frontend/diagram_id.jl:19-69, mirror symmetry :81-96.

Only what decides the *structure* of the graphs is mirrored (which leaves,
which products, which factors, grouping by external tau); the quantum-operator
bookkeeping is not needed by the evaluator.
"""
from __future__ import annotations

import math
import re
from typing import List, Optional, Sequence, Tuple

from .graph import Graph, Prod, Sum, linear_combination, multi_product

__all__ = ["read_diagrams", "diagsGV", "BareGreenId", "BareInteractionId", "SigmaId", "PolarId", "GenericId"]

_INT = re.compile(r"[-+]?\d+")


def _ints(s: str) -> List[int]:
    return [int(m) for m in _INT.findall(s)]       # readfile.jl:5-8


def mirror_symmetrize(k: Sequence[float]) -> Tuple[float, ...]:
    # diagram_id.jl:81-96
    for v in k:
        if v != 0:
            if v > 0:
                return tuple(float(x) for x in k)
            return tuple(0.0 if x == 0 else -float(x) for x in k)
    return tuple(float(x) for x in k)


class _Id:
    def __eq__(self, other):
        return type(self) is type(other) and self.equiv_key() == other.equiv_key()

    def __hash__(self):
        return hash(self.equiv_key())

    def __repr__(self):
        return f"{type(self).__name__}{self.equiv_key()}"


class BareGreenId(_Id):
    """diagram_id.jl:19-33: equality on (type, extT, extK)."""

    def __init__(self, k, t, type: str = "Dynamic"):
        self.type, self.extK, self.extT = type, mirror_symmetrize(k), tuple(t)

    def equiv_key(self):
        return ("G", self.type, self.extT, self.extK)


class BareInteractionId(_Id):
    """diagram_id.jl:35-69: all equal-time extT pairs compare equal."""

    def __init__(self, response: str, k, t=(0, 0), type: str = "Instant"):
        self.response, self.type, self.extK, self.extT = response, type, mirror_symmetrize(k), tuple(t)

    def equiv_key(self):
        t = "equal-time" if self.extT[0] == self.extT[1] else self.extT
        return ("V", self.response, self.type, self.extK, t)


class SigmaId(_Id):
    def __init__(self, para, type: str, k, t):
        self.para, self.type, self.extK, self.extT = para, type, tuple(float(x) for x in k), tuple(t)

    def equiv_key(self):
        return ("Sigma", self.para, self.type, self.extK, self.extT)


class PolarId(_Id):
    def __init__(self, para, response: str, k, t):
        self.para, self.response, self.extK, self.extT = para, response, tuple(float(x) for x in k), tuple(t)

    def equiv_key(self):
        return ("Polar", self.para, self.response, self.extK, self.extT)


class GenericId(_Id):
    def __init__(self, para, extra=None):
        self.para, self.extra = para, extra
        self.extT = ()

    def equiv_key(self):
        return ("Generic", self.para, self.extra)


def _exchange(perm: List[int], legs: List[List[int]], index: int, ext_num: int, offset_ver4: int):
    # readfile.jl:15-28 (perm holds 1-based values)
    pad = len(legs) - offset_ver4
    inds = [((index - 1) >> b) & 1 for b in range(pad)]          # digits(..., base=2, pad): little endian
    permu = list(perm)
    legs_ex = [list(l) for l in legs]
    for i, v in enumerate(reversed(inds), start=1):
        if v == 0:
            continue
        loc1 = perm.index(2 * i - 1 + ext_num)
        loc2 = perm.index(2 * i + ext_num)
        permu[loc1], permu[loc2] = permu[loc2], permu[loc1]
        j = i + offset_ver4 - 1
        legs_ex[j][1], legs_ex[j][3] = legs[j][3], legs[j][1]
    return permu, legs_ex


def _read_one(diag_type: str, lines: List[str], GNum: int, verNum: int, loopNum: int, extIndex: List[int],
              spinPolarPara: float, offset_ver4: int) -> Graph:
    # readfile.jl:475-588
    it = iter(lines)

    def expect(title):
        ln = next(it)
        assert title in ln, (title, ln)

    isDynamic = verNum != 1
    expect("Permutation")
    permutation = [x + 1 for x in _ints(next(it))]
    assert len(permutation) == len(set(permutation)) == GNum
    expect("SymFactor")
    symfactor = float(next(it))
    expect("GType")
    opGType = _ints(next(it))
    assert len(opGType) == GNum
    expect("VertexBasis")
    tau = [x + 1 for x in _ints(next(it))]
    next(it)
    expect("LoopBasis")
    basis = [[0] * loopNum for _ in range(GNum)]
    for i in range(loopNum):
        x = [int(v) for v in next(it).split()]
        assert len(x) == GNum
        for g in range(GNum):
            basis[g][i] = x[g]
    expect("Ver4Legs")
    if verNum == 0:
        ver4Legs: List[List[int]] = []
    else:
        ver4Legs = [_ints(s) for s in next(it).split("|")[:verNum]]
    expect("WType")
    if verNum > 0:
        next(it)
    expect("SpinFactor")
    spinFactors = _ints(next(it))

    ext = [x + 1 for x in extIndex]
    if diag_type == "sigma":
        ext[1] = permutation.index(ext[0]) + 1
    extNum = len(ext)
    extK = [0.0] * loopNum

    greens = []
    for ind1, ind2 in enumerate(permutation, start=1):
        if opGType[ind1 - 1] == -2:
            continue
        greens.append(Graph([], properties=BareGreenId(k=basis[ind1 - 1], t=(tau[ind1 - 1], tau[ind2 - 1]))))
    fermi_greenProd = Graph(greens, operator=Prod())

    interactions: List[Graph] = []
    spinfactors_existed: List[float] = []
    for iex, sf in enumerate(spinFactors, start=1):
        if sf == 0:
            continue
        spinfactors_existed.append(math.copysign(1.0, sf) * (2 / (1 + spinPolarPara)) ** math.log2(abs(sf)))
        _, legs_ex = _exchange(permutation, ver4Legs, iex, extNum, offset_ver4)
        leafs = []
        for leg in legs_ex:
            ind1, ind2 = leg[1] + 1, leg[3] + 1
            cur = [a - b for a, b in zip(basis[leg[0]], basis[ind1 - 1])]
            assert cur == [a - b for a, b in zip(basis[ind2 - 1], basis[leg[2]])]     # momentum conservation
            leafs.append(Graph([], properties=BareInteractionId("ChargeCharge", k=cur, t=(tau[ind1 - 1], tau[ind2 - 1]))))
        if not leafs:
            continue
        interactions.append(Graph(leafs, operator=Prod()))

    innerLoopNum = loopNum - extNum + 1
    extT = tuple(tau[i - 1] for i in ext)
    if diag_type == "freeEnergy":
        diagid = GenericId(innerLoopNum - 1)
    elif diag_type == "chargePolar":
        diagid = PolarId(innerLoopNum, "ChargeCharge", extK, extT)
    elif diag_type == "spinPolar":
        diagid = PolarId(innerLoopNum, "SpinSpin", extK, extT)
    elif diag_type == "sigma":
        diagid = SigmaId(innerLoopNum, "Dynamic" if isDynamic else "Instant", extK, extT)
    else:
        raise ValueError(f"no support for {diag_type} diagram")
    facs = [s * symfactor for s in spinfactors_existed]
    if not interactions:
        return Graph([fermi_greenProd], subgraph_factors=facs, operator=Sum(), properties=diagid)
    inters = Graph(interactions, subgraph_factors=facs, operator=Sum())
    return multi_product(fermi_greenProd, inters, properties=diagid)


_KEYWORDS = ["SelfEnergy", "DiagNum", "Order", "GNum", "Ver4Num", "LoopNum", "ExtLoopIndex",
             "DummyLoopIndex", "TauNum", "ExtTauIndex", "DummyTauIndex"]


def read_diagrams(filename: str, diag_type: str = "sigma", spinPolarPara: float = 0.0) -> List[Graph]:
    """readfile.jl:412-473 (the ``Graph`` flavour used by ``diagsGV(type, order)``)."""
    with open(filename) as f:
        text = f.read()
    lines = text.split("\n")
    diagNum, loopNum, verNum, GNum = 1, 1, 0, 2
    extIndex: List[int] = []
    pos = 0
    for kw in _KEYWORDS:                       # matched positionally until the first empty line
        line = lines[pos]
        if len(line) == 0:
            break
        v = _ints(line)
        if kw == "DiagNum":
            diagNum = v[0]
        elif kw == "GNum":
            GNum = v[0]
        elif kw == "Ver4Num":
            verNum = v[1]                      # "#Ver4Num: 3" -> [4, 3]
        elif kw == "LoopNum":
            loopNum = v[0]
        elif kw == "ExtTauIndex":
            extIndex = v
        pos += 1
    assert lines[pos] == ""
    pos += 1
    offset_ver4 = 1 if diag_type == "sigma" else 0
    diagrams: List[Graph] = []
    for _ in range(diagNum):
        blk = []
        while pos < len(lines) and lines[pos] != "":
            blk.append(lines[pos])
            pos += 1
        pos += 1
        diagrams.append(_read_one(diag_type, blk, GNum, verNum, loopNum, extIndex, spinPolarPara, offset_ver4))
    if diag_type == "freeEnergy":
        return [linear_combination(diagrams, properties=diagrams[0].properties)]
    keys: List[tuple] = []
    groups = {}
    for d in diagrams:
        k = d.properties.extT
        if k not in groups:
            groups[k] = []
            keys.append(k)
        groups[k].append(d)
    return [linear_combination(groups[k], properties=groups[k][0].properties) for k in keys]


def diagsGV(diag_type: str, order: int, root_dir: str, spinPolarPara: float = 0.0) -> List[Graph]:
    """GV.jl:77-93; ``root_dir`` = .../src/frontend/GV_diagrams."""
    sub = {"spinPolar": ("groups_spin", "Polar"), "chargePolar": ("groups_charge", "Polar"),
           "sigma": ("groups_sigma", "Sigma"), "green": ("groups_green", "Green"),
           "freeEnergy": ("groups_free_energy", "FreeEnergy")}
    if diag_type not in sub:
        raise ValueError(f"no support for {diag_type} diagram")
    d, stem = sub[diag_type]
    return read_diagrams(f"{root_dir}/{d}/{stem}{order}_0_0.diag", diag_type, spinPolarPara)
