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

from ..graph import Graph, Prod, Sum, linear_combination, multi_product

__all__ = ["read_diagrams", "diagsGV", "read_vertex4diagrams", "diagsGV_ver4", "parse_vertex4_catalog", "BareGreenId", "BareInteractionId",
           "SigmaId", "PolarId", "GenericId", "Ver4Id"]

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

    def __init__(self, k, t, type: str = "Dynamic", symmetrize: bool = True):
        self.type, self.extK, self.extT = type, (mirror_symmetrize(k) if symmetrize else tuple(float(x) for x in k)), tuple(t)

    def equiv_key(self):
        return ("G", self.type, self.extT, self.extK)


class BareInteractionId(_Id):
    """diagram_id.jl:35-69: all equal-time extT pairs compare equal."""

    def __init__(self, response: str, k, t=(0, 0), type: str = "Instant", symmetrize: bool = True):
        self.response, self.type, self.extT = response, type, tuple(t)
        self.extK = mirror_symmetrize(k) if symmetrize else tuple(float(x) for x in k)

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


class Ver4Id(_Id):
    """diagram_id.jl:166-185.  ``para`` is a ``DiagPara`` for Parquet-built vertices and the pair (Di/Ex, innerLoopNum)
    for the ones read from a GV catalog (readfile.jl:397-398)."""

    def __init__(self, para, response: str, type: str = "Dynamic", *, k, t=(0, 0, 0, 0), chan: str = "AnyChan"):
        self.para, self.response, self.type, self.channel = para, response, type, chan
        self.extK, self.extT = tuple(tuple(float(x) for x in kk) for kk in k), tuple(t)

    def equiv_key(self):
        return ("Ver4", self.para, self.response, self.type, self.channel, self.extK, self.extT)


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


# --- the 4-point vertex catalogs (groups_vertex4/Vertex4<n>_0_0.diag, Vertex4I<n>_0_0.diag) ---------------------------------
_KEYWORDS_VER4 = ["Vertex4", "DiagNum", "Order", "GNum", "Ver4Num", "LoopNum", "ExtLoopIndex", "DummyLoopIndex", "TauNum", "DummyTauIndex"]


def parse_vertex4_catalog(filename: str) -> dict:
    """The numbers of a vertex catalog as arrays (header: readfile.jl:191-218; blocks: :267-333): ``GNum, verNum, loopNum``
    and, per Hugenholtz diagram, ``permutation`` (0-based as in the file), ``symfactor``, ``channel``, ``gtype``, ``tau`` (first
    VertexBasis row), ``loopbasis [loopNum][GNum]``, ``ver4legs``, ``spin``, ``diex``, ``proper``.  This is the form the
    package ships the two fully-irreducible catalogs in (data/vertex4I<n>.npz): data of the reference, not its text."""
    import numpy as np
    with open(filename) as f:
        lines = f.read().split("\n")
    diagNum, loopNum, verNum, GNum = 1, 1, 0, 2
    pos = 0
    for kw in _KEYWORDS_VER4:
        line = lines[pos]
        if len(line) == 0:
            break
        v = _ints(line)
        if kw == "DiagNum":
            diagNum = v[0]
        elif kw == "GNum":
            GNum = v[0]
        elif kw == "Ver4Num":
            verNum = v[1]
        elif kw == "LoopNum":
            loopNum = v[0]
        pos += 1
    assert lines[pos] == ""
    pos += 1
    out = dict(GNum=GNum, verNum=verNum, loopNum=loopNum, permutation=[], symfactor=[], channel=[], gtype=[], tau=[], loopbasis=[],
               ver4legs=[], spin=[], diex=[], proper=[])
    for _ in range(diagNum):
        blk = []
        while pos < len(lines) and lines[pos] != "":
            blk.append(lines[pos])
            pos += 1
        pos += 1
        it = iter(blk)

        def expect(title):
            ln = next(it)
            assert title in ln, (title, ln)

        expect("Permutation")
        out["permutation"].append(_ints(next(it)))
        expect("SymFactor")
        out["symfactor"].append(float(next(it)))
        expect("Channel")
        out["channel"].append(next(it).strip())
        expect("GType")
        out["gtype"].append(_ints(next(it)))
        expect("VertexBasis")
        out["tau"].append(_ints(next(it)))
        next(it)
        expect("LoopBasis")
        out["loopbasis"].append([[int(v) for v in next(it).split()] for _ in range(loopNum)])
        expect("Ver4Legs")
        out["ver4legs"].append([] if verNum == 0 else [_ints(x) for x in next(it).split("|")[:verNum]])
        expect("WType")
        if verNum > 0:
            next(it)
        expect("SpinFactor")
        out["spin"].append(_ints(next(it)))
        expect("Di/Ex")
        out["diex"].append(_ints(next(it)))
        expect("Proper/ImProper")
        out["proper"].append(_ints(next(it)))
    chans = ["Alli", "PHr", "PHEr", "PPr"]
    return dict(GNum=np.int64(GNum), verNum=np.int64(verNum), loopNum=np.int64(loopNum), permutation=np.array(out["permutation"], np.int32),
                symfactor=np.array(out["symfactor"], np.float64), channel=np.array([chans.index(c) for c in out["channel"]], np.int32),
                gtype=np.array(out["gtype"], np.int32), tau=np.array(out["tau"], np.int32), loopbasis=np.array(out["loopbasis"], np.int32),
                ver4legs=np.array(out["ver4legs"], np.int32).reshape(len(out["spin"]), verNum, 4), spin=np.array(out["spin"], np.int32),
                diex=np.array(out["diex"], np.int32), proper=np.array(out["proper"], np.int32))


def _one_vertex4(cat: dict, d: int, spinPolarPara: float, channels, filter) -> Optional[Tuple[Graph, Graph]]:
    # readfile.jl:267-410
    GNum, verNum, loopNum = int(cat["GNum"]), int(cat["verNum"]), int(cat["loopNum"])
    flag_proper = "Proper" in filter
    isDynamic = verNum != 1
    permutation = [int(x) + 1 for x in cat["permutation"][d]]
    assert len(permutation) == len(set(permutation)) == GNum
    symfactor = float(cat["symfactor"][d])
    channel = ["Alli", "PHr", "PHEr", "PPr"][int(cat["channel"][d])]
    if channel not in channels:
        return None
    opGType = [int(x) for x in cat["gtype"][d]]
    tau = [int(x) for x in cat["tau"][d]]                  # (not shifted: readfile.jl:301)
    basis = [[int(cat["loopbasis"][d][i][g]) for i in range(loopNum)] for g in range(GNum)]
    ver4Legs = [[int(x) for x in leg] for leg in cat["ver4legs"][d]]
    spinFactors = [int(x) for x in cat["spin"][d]]
    DiEx = [int(x) for x in cat["diex"][d]]
    proper = [int(x) for x in cat["proper"][d]]
    innerLoopNum = loopNum - 3
    extK = [[0.0] * loopNum for _ in range(4)]
    for i in range(3):
        extK[i][i] = 1.0
        extK[3][i] = float((-1) ** i)
    extIndex = [1, 0, 2, 0]
    for ind1, ind2 in enumerate(permutation, start=1):
        if ind1 in (1, 2):
            continue
        if opGType[ind1 - 1] == -2:
            if ind2 == 1:
                extIndex[1] = ind1
            elif ind2 == 2:
                extIndex[3] = ind1
            else:
                raise ValueError(f"error GType for ({ind1}, {ind2}).")
    greens = []
    for ind1, ind2 in enumerate(permutation, start=1):
        if opGType[ind1 - 1] == -2:
            continue
        greens.append(Graph([], properties=BareGreenId(k=basis[ind1 - 1], t=(tau[ind1 - 1], tau[ind2 - 1]))))
    fermi_greenProd = Graph(greens, operator=Prod())
    inter_Di: List[Graph] = []
    inter_Ex: List[Graph] = []
    for iex, sf in enumerate(spinFactors, start=1):
        if sf == 0:
            continue
        if flag_proper and proper[iex - 1] == 1:
            continue
        permu, legs_ex = _exchange(permutation, ver4Legs, iex, 2, 0)
        extIndex[0], extIndex[2] = permu[0], permu[1]
        leafs = []
        for leg in legs_ex:
            ind1, ind2 = leg[1] + 1, leg[3] + 1
            cur = [a - b for a, b in zip(basis[leg[0]], basis[ind1 - 1])]
            assert cur == [a - b for a, b in zip(basis[ind2 - 1], basis[leg[2]])]     # momentum conservation
            leafs.append(Graph([], properties=BareInteractionId("ChargeCharge", k=cur, t=(tau[ind1 - 1], tau[ind2 - 1]))))
        (inter_Di if DiEx[iex - 1] == 0 else inter_Ex).append(Graph.new(leafs, operator=Prod(), factor=sf * symfactor))
    typ = "Dynamic" if isDynamic else "Instant"
    extT = tuple(tau[i - 1] for i in extIndex)
    id_Di = Ver4Id((0, innerLoopNum), "UpDown", typ, k=extK, t=extT, chan=channel)
    id_Ex = Ver4Id((1, innerLoopNum), "ChargeCharge", typ, k=extK, t=extT, chan=channel)
    if not fermi_greenProd.subgraphs:
        return Graph(inter_Di, operator=Sum(), properties=id_Di), Graph(inter_Ex, operator=Sum(), properties=id_Ex)
    return (multi_product(fermi_greenProd, Graph(inter_Di, operator=Sum()), properties=id_Di),
            multi_product(fermi_greenProd, Graph(inter_Ex, operator=Sum()), properties=id_Ex))


def read_vertex4diagrams(catalog, spinPolarPara: float = 0.0, filter=("NoHartree",), channels=("PHr", "PHEr", "PPr", "Alli")) -> List[Graph]:
    """readfile.jl:191-265.  ``catalog``: a ``.diag`` file name or the arrays of ``parse_vertex4_catalog``.  Returns, per group of
    external times and channel, the pair (UpUp, UpDown) -- UpDown = the direct diagrams, UpUp = direct + exchange.  (The
    reference walks the groups in the order of a ``Dict``'s keys; here in order of first appearance.)"""
    cat = parse_vertex4_catalog(catalog) if isinstance(catalog, str) else catalog
    diagrams: List[Graph] = []
    for d in range(len(cat["symfactor"])):
        pair = _one_vertex4(cat, d, spinPolarPara, channels, filter)
        if pair is not None:
            diagrams.extend(pair)
    innerLoopNum = int(cat["loopNum"]) - 3
    groups = {}
    keys: List[tuple] = []
    for g in diagrams:
        pr = g.properties
        groups.setdefault((pr.extT, pr.channel, pr.para[0]), []).append(g)
        if (pr.extT, pr.channel) not in keys:
            keys.append((pr.extT, pr.channel))
    out: List[Graph] = []
    for extT, channel in keys:
        di, ex = groups[(extT, channel, 0)], groups[(extT, channel, 1)]
        gId = di[0].properties
        gud = linear_combination(di, properties=gId)                    # direct = UpDown
        gEx = linear_combination(ex, properties=ex[0].properties)
        guu = Graph([gud, gEx], properties=Ver4Id((2, innerLoopNum), "UpUp", gId.type, k=gId.extK, t=gId.extT, chan=gId.channel))
        out.extend([guu, gud])
    return out


def diagsGV_ver4(order: int, root_dir: str, spinPolarPara: float = 0.0, channels=("PHr", "PHEr", "PPr", "Alli"), filter=("NoHartree",)) -> List[Graph]:
    """GV.jl:106-114."""
    stem = "Vertex4I" if list(channels) == ["Alli"] else "Vertex4"
    return read_vertex4diagrams(f"{root_dir}/groups_vertex4/{stem}{order}_0_0.diag", spinPolarPara, filter, channels)
