"""The Parquet front end for the self-energy, restated on the host-side graph mirror: the caller that produces
BASELINE.json's configs 1-4 (``Parquet.build(DiagPara(type=SigmaDiag, innerLoopNum=n))``), so that the evaluator is
measured on the *real* 4-loop Parquet self-energy rather than on a stand-in.

Reference: src/frontend/parquet/parquet.jl:57-143 (Interaction, ParquetBlocks, DiagPara), common.jl (build,
orderedPartition, index helpers), filter.jl (notProper, isValidG, isValidSigma), operation.jl:1-176 (mergeby),
vertex4.jl (vertex4, bubble!, bubble2diag!, RPA_chain!, bareVer4, legBasis, tauBasis), sigma.jl, green.jl,
vertex3.jl, polarization.jl; ids: src/frontend/diagram_id.jl.

What is reproduced, and what cannot be.  The *graph* -- nodes, operators, factors, leaves and their identities (so the
leaf merging of ``optimize!``), the loop-momentum and time indices -- follows the reference line by line.  Two places of
the reference iterate a hash container, whose order depends on the Julia version: ``Set(permutations(p))`` in
``orderedPartition`` (the reference's own test compares it as a set, test/front_end.jl:149-156) and ``keys(ver8)`` of a
``Dict{Any,Any}`` in ``bubble!``.  They decide in which order the terms of some Sums are listed, i.e. the
floating-point association of those Sums, not the graph; here the first uses the order in which Combinatorics.jl yields
the permutations and the second insertion order.  Pinned by: the optimized 2-loop graph of assets/sigma_o2.svg
 reproduced up to the order of operands, and the diagram counts of the reference's own tests: the
self-energy (1, 3, 18, 171; test/front_end.jl:600-652), the 3-point vertex (1, 10, 109; :701-755) and the polarization
in three variants (2, 2, 20, 218 / 2, 2, 32, 326 / 2, 2, 28, 274; :758-826) -- the last two exercise ``vertex4`` with
all three channels at the top level; and by the reference's other front end: with fermionic signs the all-ones sums of
the self-energy (2..6 loops), the vertex function (1..4) and the polarization (1..5) equal the sums of
SymFactor * SpinFactor of the GV catalogs (tests/test_parquet.py).

The fully irreducible vertex ``Alli`` at 3 and 4 loops comes, as in the reference, from the GV vertex catalogs
(``vertex4I_diags``, parquet.jl:216-231; their numbers ship as data/vertex4I<n>.npz) through ``update_extKT``.  Not
restated: ep_coupling.
"""
from __future__ import annotations

import copy
import functools
import itertools
import math
import os
from dataclasses import dataclass, replace
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

from ..graph import Graph, Prod, Sum, uid
from .gv import BareGreenId, BareInteractionId, GenericId, PolarId, SigmaId, Ver4Id, _Id

__all__ = ["DiagPara", "Interaction", "ParquetBlocks", "build", "sigma", "vertex4", "vertex3", "polarization", "green", "orderedPartition",
           "findFirstLoopIdx", "findFirstTauIdx", "isValidG", "isValidSigma", "mergeby", "count_sigma_G2v"]

# enums (frontends.jl:9-47, parquet.jl:43-54); the integer is the enum's value, which DataFrames sorts groups by
Alli, PHr, PHEr, PPr, AnyChan = "Alli", "PHr", "PHEr", "PPr", "AnyChan"
_CHAN_INDEX = {Alli: 1, PHr: 2, PHEr: 3, PPr: 4, AnyChan: 5}
Wirreducible, Girreducible, NoHartree, NoFock, NoBubble, Proper, DirectOnly = (
    "Wirreducible", "Girreducible", "NoHartree", "NoFock", "NoBubble", "Proper", "DirectOnly")
Composite, ChargeCharge, SpinSpin, ProperChargeCharge, ProperSpinSpin, UpUp, UpDown = (
    "Composite", "ChargeCharge", "SpinSpin", "ProperChargeCharge", "ProperSpinSpin", "UpUp", "UpDown")
_RESPONSE_INDEX = {Composite: 0, ChargeCharge: 1, SpinSpin: 2, ProperChargeCharge: 3, ProperSpinSpin: 4, UpUp: 5, UpDown: 6}
Instant, Dynamic = "Instant", "Dynamic"
_TYPE_INDEX = {Instant: 0, Dynamic: 1}
VacuumDiag, SigmaDiag, GreenDiag, PolarDiag, Ver3Diag, Ver4Diag = "VacuumDiag", "SigmaDiag", "GreenDiag", "PolarDiag", "Ver3Diag", "Ver4Diag"
Di, Ex = "Di", "Ex"
DI, EX = 0, 1
INL, OUTL, INR, OUTR = 0, 1, 2, 3
SymFactor = [1.0, -1.0, 1.0, -0.5, +1.0, -1.0]          # parquet.jl:32


@dataclass(frozen=True, init=False)
class Interaction:                                        # parquet.jl:57-66
    response: str
    type: Tuple[str, ...]

    def __init__(self, response, type):
        object.__setattr__(self, "response", response)
        object.__setattr__(self, "type", tuple(sorted({type} if isinstance(type, str) else set(type))))


@dataclass(frozen=True)
class ParquetBlocks:                                      # parquet.jl:84-92
    phi: Tuple[str, ...] = (Alli, PHEr, PPr)
    ppi: Tuple[str, ...] = (Alli, PHr, PHEr)
    G4: Optional[Tuple[str, ...]] = None                  # the reference's field is called Γ4

    def __post_init__(self):
        object.__setattr__(self, "phi", tuple(self.phi))
        object.__setattr__(self, "ppi", tuple(self.ppi))
        if self.G4 is None:                               # union(phi, ppi): phi, then what ppi adds
            object.__setattr__(self, "G4", self.phi + tuple(c for c in self.ppi if c not in self.phi))
        else:
            object.__setattr__(self, "G4", tuple(self.G4))


def interactionTauNum(hasTau: bool, interactionSet) -> int:       # common.jl:89-99
    if not hasTau:
        return 0
    return 2 if any(Dynamic in i.type for i in interactionSet) else 1


def innerTauNum(type: str, innerLoopNum: int, interactionTauNum_: int) -> int:    # common.jl:70-87
    if type == Ver4Diag:
        return (innerLoopNum + 1) * interactionTauNum_
    if type in (SigmaDiag, GreenDiag):
        return innerLoopNum * interactionTauNum_
    if type == VacuumDiag:
        return (innerLoopNum - 1) * interactionTauNum_
    if type == PolarDiag:
        return 1 + innerTauNum(Ver3Diag, innerLoopNum - 1, interactionTauNum_)
    if type == Ver3Diag:
        return 1 + innerTauNum(Ver4Diag, innerLoopNum - 1, interactionTauNum_)
    raise ValueError("not implemented!")


def firstTauIdx(type: str, offset: int = 0) -> int:               # common.jl:101-111
    return (3 if type == GreenDiag else 1) + offset


def firstLoopIdx(type: str, offset: int = 0) -> int:              # common.jl:113-129
    return {Ver4Diag: 4, SigmaDiag: 2, GreenDiag: 2, PolarDiag: 2, Ver3Diag: 3, VacuumDiag: 1}[type] + offset


@dataclass(frozen=True)
class DiagPara:
    """parquet.jl:104-125 (``@with_kw``: the later defaults are computed from the earlier fields); ``reconstruct`` keeps
    every field that is not named, including ``totalLoopNum`` / ``totalTauNum`` (parquet.jl:135-156)."""
    type: str
    innerLoopNum: int
    isFermi: bool = True
    spin: int = 2
    interaction: Tuple[Interaction, ...] = (Interaction(ChargeCharge, (Instant,)),)
    firstLoopIdx: Optional[int] = None
    totalLoopNum: Optional[int] = None
    hasTau: bool = True
    firstTauIdx: Optional[int] = None
    totalTauNum: Optional[int] = None
    filter: Tuple[str, ...] = (NoHartree,)
    transferLoop: Tuple[float, ...] = ()
    extra: Any = None

    def __post_init__(self):
        s = lambda k, v: object.__setattr__(self, k, v)
        s("interaction", tuple(self.interaction))
        s("filter", tuple(self.filter))
        s("transferLoop", tuple(float(x) for x in self.transferLoop))
        if self.firstLoopIdx is None:
            s("firstLoopIdx", firstLoopIdx(self.type))
        if self.totalLoopNum is None:
            s("totalLoopNum", self.firstLoopIdx + self.innerLoopNum - 1)
        if self.firstTauIdx is None:
            s("firstTauIdx", firstTauIdx(self.type))
        if self.totalTauNum is None:
            s("totalTauNum", self.firstTauIdx + innerTauNum(self.type, self.innerLoopNum, interactionTauNum(self.hasTau, self.interaction)) - 1)

    def __eq__(self, other):                              # parquet.jl:181-213: filter and interaction as sets
        if not isinstance(other, DiagPara):
            return NotImplemented
        for f in self.__dataclass_fields__:
            a, b = getattr(self, f), getattr(other, f)
            if f in ("filter", "interaction"):
                if set(a) != set(b):
                    return False
            elif f == "transferLoop":
                if (len(a) == 0) != (len(b) == 0) or (a and not _isapprox_vec(a, b)):
                    return False
            elif a != b:
                return False
        return True

    def __hash__(self):
        return hash((self.type, self.innerLoopNum, self.firstLoopIdx, self.firstTauIdx, self.totalLoopNum, self.totalTauNum))


def reconstruct(p: DiagPara, **kw) -> DiagPara:
    return replace(p, **kw)


def _interactionTauNum(para: DiagPara) -> int:
    return interactionTauNum(para.hasTau, para.interaction)


def _isapprox_vec(a, b) -> bool:                          # Julia isapprox on vectors: norm(a-b) <= sqrt(eps) max(norm a, norm b)
    if len(a) != len(b):
        return False
    d = math.sqrt(sum((x - y) ** 2 for x, y in zip(a, b)))
    return d <= 1.4901161193847656e-08 * max(math.sqrt(sum(x * x for x in a)), math.sqrt(sum(y * y for y in b)))


def getK(loopNum: int, loopIdx: int) -> List[float]:              # common.jl:151-155 (loopIdx 1-based)
    k = [0.0] * loopNum
    k[loopIdx - 1] = 1.0
    return k


def _partitions(n: int, m: int) -> List[List[int]]:
    """Combinatorics.partitions(n, m): the partitions of n into exactly m positive parts, each non-increasing, in the
    order of FixedPartitions' iterator (the first part descending: [n-m+1, 1, ..] first)."""
    out: List[List[int]] = []

    def rec(rest: int, parts: int, cap: int, cur: List[int]):
        if parts == 0:
            if rest == 0:
                out.append(list(cur))
            return
        for first in range(min(cap, rest - (parts - 1)), 0, -1):
            if first * parts < rest:
                break
            rec(rest - first, parts - 1, first, cur + [first])

    rec(n, m, n, [])
    return out


def orderedPartition(_total: int, n: int, lowerbound: int = 1) -> List[List[int]]:
    """common.jl:42-61.  Every distinct ordering of every partition of ``_total`` into ``n`` parts >= ``lowerbound``.
    (The reference collects the orderings of one partition in a ``Set``; here they come in the order of
    ``permutations(p)``, first occurrence kept -- see the module docstring.)"""
    assert lowerbound >= 0
    total = _total - n * (lowerbound - 1)
    assert total >= n
    out: List[List[int]] = []
    for p in _partitions(total, n):
        p = [x + (lowerbound - 1) for x in p]
        assert sum(p) == _total and all(i >= lowerbound for i in p)
        seen = []
        for q in itertools.permutations(p):
            if list(q) not in seen:
                seen.append(list(q))
        out.extend(seen)
    return out


def findFirstLoopIdx(partition: Sequence[int], firstidx: int):    # common.jl:158-168
    acc = list(itertools.accumulate(partition, initial=firstidx))[1:]
    return [firstidx] + acc[:-1], acc[-1] - 1


def findFirstTauIdx(partition: Sequence[int], type: Sequence[str], firstidx: int, _tauNum: int):   # common.jl:170-182
    assert len(partition) == len(type) and _tauNum >= 0
    taupartition = [innerTauNum(type[i], p, _tauNum) for i, p in enumerate(partition)]
    acc = list(itertools.accumulate(taupartition, initial=firstidx))[1:]
    return [firstidx] + acc[:-1], acc[-1] - 1


# --- filter.jl ------------------------------------------------------------------------------------------------------
def notProper(para: DiagPara, K) -> bool:                 # filter.jl:20-29
    if Proper in para.filter:
        assert len(para.transferLoop) > 0, "Please initialize para.transferLoop to check proper diagrams."
        if _isapprox_vec(list(para.transferLoop[:len(K)]), list(K)):
            return True
    return False


def isValidG(filter_or_para, innerLoopNum: Optional[int] = None) -> bool:      # filter.jl:32-48
    if isinstance(filter_or_para, DiagPara):
        assert filter_or_para.type == GreenDiag
        filter_or_para, innerLoopNum = filter_or_para.filter, filter_or_para.innerLoopNum
    f = filter_or_para
    if NoFock in f and NoHartree in f and innerLoopNum == 1:
        return False
    if Girreducible in f and innerLoopNum > 0:
        return False
    return True


def isValidSigma(filter, innerLoopNum: int, subdiagram: bool) -> bool:          # filter.jl:50-65
    assert innerLoopNum >= 0
    if innerLoopNum == 0:
        return False
    if subdiagram and Girreducible in filter:
        return False
    if subdiagram and NoFock in filter and NoHartree in filter and innerLoopNum == 1:
        return False
    return True


# --- ids the builder reads back (diagram_id.jl:98-185) ----------------------------------------------------------------
class GreenId(_Id):
    def __init__(self, para, k, t, type: str = Dynamic):
        self.para, self.type, self.extK, self.extT = para, type, tuple(float(x) for x in k), tuple(t)

    def equiv_key(self):
        return ("Green", self.para, self.type, self.extK, self.extT)


class Ver3Id(_Id):
    def __init__(self, para, response: str, *, k, t=(0, 0, 0)):
        self.para, self.response = para, response
        self.extK, self.extT = tuple(tuple(float(x) for x in kk) for kk in k), tuple(t)

    def equiv_key(self):
        return ("Ver3", self.para, self.response, self.extK, self.extT)


# --- operation.jl: mergeby ----------------------------------------------------------------------------------------------
Row = Dict[str, Any]


def _sort_token(v):
    if isinstance(v, str):
        return _RESPONSE_INDEX.get(v, _TYPE_INDEX.get(v, 0))
    return v


def _mergediag(group: List[Row], id, operator, name) -> Graph:    # operation.jl:26-37
    if len(group) == 1:
        d = group[0]["diagram"]
        if isinstance(id, GenericId) or type(id) is type(d.properties):
            return d
    return Graph([r["diagram"] for r in group], properties=id, operator=operator, name=name)


def mergeby(df, fields=None, *, operator=None, name: str = "none", getid: Optional[Callable] = None):
    """operation.jl:93-111 (table form: rows grouped by ``fields``, groups in sorted key order, the rows of a group in
    table order) and :145-163 (vector form)."""
    operator = operator if operator is not None else Sum()
    if fields is None and (not df or isinstance(df[0], Graph)):
        diags = list(df)
        if not diags:
            return diags
        id = getid(diags) if getid else GenericId(diags[0].properties.para)
        if len(diags) == 1 and (isinstance(id, GenericId) or type(id) is type(diags[0].properties)):
            return diags
        return [Graph(diags, properties=id, operator=operator, name=name)]
    if not df:
        return df
    single = isinstance(fields, str)
    fields = [fields] if single else list(fields or [])
    groups: Dict[tuple, List[Row]] = {}
    for r in df:
        groups.setdefault(tuple(r[f] for f in fields), []).append(r)
    if getid is None:
        getid = lambda g: GenericId(g[0]["diagram"].properties.para, tuple(g[0][fields[0]]) if single else tuple(g[0][f] for f in fields))
    out = []
    for key in sorted(groups, key=lambda k: tuple(_sort_token(v) for v in k)):
        g = groups[key]
        row = {f: v for f, v in zip(fields, key)}
        row["diagram"] = _mergediag(g, getid(g), operator, name)
        row["hash"] = row["diagram"].id
        out.append(row)
    return out


# --- vertex4.jl -----------------------------------------------------------------------------------------------------------
def _vadd(a, b, sa=1.0, sb=1.0):
    return [sa * x + sb * y for x, y in zip(a, b)]


def legBasis(chan: str, legK, loopIdx: int):              # vertex4.jl:386-412
    KinL, KoutL, KinR, KoutR = legK
    K = [0.0] * len(KinL)
    K[loopIdx - 1] = 1.0
    if chan == PHr:
        Kx = _vadd(_vadd(KoutL, K), KinL, 1.0, -1.0)
        LLegK, RLegK = [KinL, KoutL, Kx, K], [K, Kx, KinR, KoutR]
    elif chan == PHEr:
        Kx = _vadd(_vadd(KoutR, K), KinL, 1.0, -1.0)
        LLegK, RLegK = [KinL, KoutR, Kx, K], [K, Kx, KinR, KoutL]
    elif chan == PPr:
        Kx = _vadd(_vadd(KinL, KinR), K, 1.0, -1.0)
        LLegK, RLegK = [KinL, Kx, KinR, K], [K, KoutL, Kx, KoutR]
    else:
        raise ValueError("not implemented!")
    return LLegK, K, RLegK, Kx


def tauBasis(chan: str, LvT, RvT):                        # vertex4.jl:414-436
    G0T = (LvT[OUTR], RvT[INL])
    if chan == PHr:
        extT, GxT = (LvT[INL], LvT[OUTL], RvT[INR], RvT[OUTR]), (RvT[OUTL], LvT[INR])
    elif chan == PHEr:
        extT, GxT = (LvT[INL], RvT[OUTR], RvT[INR], LvT[OUTL]), (RvT[OUTL], LvT[INR])
    elif chan == PPr:
        extT, GxT = (LvT[INL], RvT[OUTL], LvT[INR], RvT[OUTR]), (LvT[OUTL], RvT[INR])
    else:
        raise ValueError("not implemented!")
    assert sorted(G0T + GxT + extT) == sorted(tuple(LvT) + tuple(RvT))
    assert extT[INL] == LvT[INL]
    return extT, G0T, GxT


def _factor(para: DiagPara, chan: str) -> float:          # vertex4.jl:439-446
    f = SymFactor[_CHAN_INDEX[chan] - 1]
    return f if para.isFermi else abs(f)


def maxVer4TauIdx(para):
    return (para.innerLoopNum + 1) * _interactionTauNum(para) + para.firstTauIdx - 1


def maxVer4LoopIdx(para):
    return para.firstLoopIdx + para.innerLoopNum - 1


def _bare(para, diex, response, type, _diex, _innerT, _q, _factor_=1.0):        # vertex4.jl:264-284
    sign = -1.0 if _diex == Di else (1.0 if para.isFermi else -1.0)
    if not notProper(para, _q) and _diex in diex:
        return Graph.new([], factor=sign * _factor_, properties=BareInteractionId(response, k=_q, t=_innerT, type=type))
    return None


def _pushbarever4(para, nodes, response, type, _extT, legK, vd, ve):            # vertex4.jl:286-297
    if vd is not None:
        nodes.append(dict(response=response, type=type, extT=_extT[DI],
                          diagram=Graph([vd], operator=Sum(), properties=Ver4Id(para, response, type, k=legK, t=_extT[DI]))))
    if ve is not None:
        nodes.append(dict(response=response, type=type, extT=_extT[EX],
                          diagram=Graph([ve], operator=Sum(), properties=Ver4Id(para, response, type, k=legK, t=_extT[EX]))))


def _pushbarever4_with_response(para, nodes, response, type, legK, q, diex, _extT, _innerT):    # vertex4.jl:299-335
    if response == UpUp:
        vd = _bare(para, diex, response, type, Di, _innerT[DI], q[DI])
        ve = _bare(para, diex, response, type, Ex, _innerT[EX], q[EX])
        _pushbarever4(para, nodes, UpUp, type, _extT, legK, vd, ve)
    elif response == UpDown:
        vd = _bare(para, diex, UpDown, type, Di, _innerT[DI], q[DI])
        _pushbarever4(para, nodes, UpDown, type, _extT, legK, vd, None)
    elif response == ChargeCharge:
        vuud = _bare(para, diex, ChargeCharge, type, Di, _innerT[DI], q[DI])
        vuue = _bare(para, diex, ChargeCharge, type, Ex, _innerT[EX], q[EX])
        _pushbarever4(para, nodes, UpUp, type, _extT, legK, vuud, vuue)
        vupd = _bare(para, diex, ChargeCharge, type, Di, _innerT[DI], q[DI])
        _pushbarever4(para, nodes, UpDown, type, _extT, legK, vupd, None)
    elif response == SpinSpin:
        vuud = _bare(para, diex, SpinSpin, type, Di, _innerT[DI], q[DI])
        vuue = _bare(para, diex, SpinSpin, type, Ex, _innerT[EX], q[EX])
        _pushbarever4(para, nodes, UpUp, type, _extT, legK, vuud, vuue)
        vupd = _bare(para, diex, SpinSpin, type, Di, _innerT[DI], q[DI], -1.0)
        vupe = _bare(para, diex, SpinSpin, type, Ex, _innerT[EX], q[EX], 2.0)
        _pushbarever4(para, nodes, UpDown, type, _extT, legK, vupd, vupe)
    else:
        raise ValueError("not implemented!")


def bareVer4(nodes, para: DiagPara, legK, diex=(Di, Ex), leftalign: bool = True):    # vertex4.jl:337-380
    KinL, KoutL, KinR = legK[0], legK[1], legK[2]
    t0 = para.firstTauIdx
    q = [_vadd(KinL, KoutL, 1.0, -1.0), _vadd(KinR, KoutL, 1.0, -1.0)]
    if para.hasTau:
        extT_ins = [(t0, t0, t0, t0), (t0, t0, t0, t0)]
        extT_ins_rightalign = [(t0 + 1,) * 4, (t0 + 1,) * 4]
        extT_dyn = [(t0, t0, t0 + 1, t0 + 1), (t0, t0 + 1, t0 + 1, t0)]
        innerT_ins = [(1, 1), (1, 1)]
        innerT_dyn = [(t0, t0 + 1), (t0, t0 + 1)]
    else:
        extT_ins = [(t0, t0, t0, t0), (t0, t0, t0, t0)]
        extT_ins_rightalign = extT_ins
        extT_dyn, innerT_ins = extT_ins, [(1, 1), (1, 1)]
        innerT_dyn = innerT_ins
    for interaction in para.interaction:
        response, typeVec = interaction.response, interaction.type
        if Instant in typeVec and Dynamic not in typeVec:
            _pushbarever4_with_response(para, nodes, response, Instant, legK, q, diex, extT_ins, innerT_ins)
        elif Instant not in typeVec and Dynamic in typeVec:
            _pushbarever4_with_response(para, nodes, response, Dynamic, legK, q, diex, extT_dyn, innerT_dyn)
        elif Instant in typeVec and Dynamic in typeVec:
            _pushbarever4_with_response(para, nodes, response, Instant, legK, q, diex,
                                        extT_ins if leftalign else extT_ins_rightalign, innerT_dyn)
            _pushbarever4_with_response(para, nodes, response, Dynamic, legK, q, diex, extT_dyn, innerT_dyn)
    return nodes


def bubble2diag(ver8, para, chan, ldiag, rdiag, extK, extrafactor):             # vertex4.jl:192-262
    lid, rid = ldiag.properties, rdiag.properties
    ln, rn = lid.response, rid.response
    vtype = Dynamic                                       # typeMap, vertex4.jl:448-460
    extT, G0T, GxT = tauBasis(chan, lid.extT, rid.extT)
    Factor = _factor(para, chan) * extrafactor

    def add(Lresponse, Rresponse, Vresponse, factor=1.0):
        key = (G0T, GxT, extT, Vresponse, vtype)
        ver8.setdefault(key, [])
        if ln == Lresponse and rn == Rresponse:
            ver8[key].append(Graph.new([ldiag, rdiag], properties=GenericId(para), operator=Prod(), factor=factor * Factor,
                                       name=f"{Lresponse}x{Rresponse} -> {chan},"))

    if chan == PHr:
        add(UpUp, UpUp, UpUp); add(UpDown, UpDown, UpUp); add(UpUp, UpDown, UpDown); add(UpDown, UpUp, UpDown)
    elif chan == PHEr:
        add(UpUp, UpUp, UpUp); add(UpDown, UpDown, UpUp)
        add(UpUp, UpUp, UpDown); add(UpDown, UpDown, UpDown); add(UpUp, UpDown, UpDown, -1.0); add(UpDown, UpUp, UpDown, -1.0)
    elif chan == PPr:
        add(UpUp, UpUp, UpUp)
        add(UpDown, UpDown, UpDown, -2.0); add(UpUp, UpDown, UpDown); add(UpDown, UpUp, UpDown)
    else:
        raise ValueError(f"chan {chan} isn't implemented!")


def bubble(ver4df, para: DiagPara, legK, chan: str, partition, level: int, name: str, blocks: ParquetBlocks,
           blockstoplevel: ParquetBlocks, extrafactor: float = 1.0):            # vertex4.jl:122-190
    TauNum = _interactionTauNum(para)
    oL, oG0, oR, oGx = partition
    if not isValidG(para.filter, oG0) or not isValidG(para.filter, oGx):
        return
    LoopIdx = para.firstLoopIdx
    idx, maxLoop = findFirstLoopIdx(partition, LoopIdx + 1)
    LfirstLoopIdx, G0firstLoopIdx, RfirstLoopIdx, GxfirstLoopIdx = idx
    assert maxLoop == maxVer4LoopIdx(para)
    idx, maxTau = findFirstTauIdx(partition, [Ver4Diag, GreenDiag, Ver4Diag, GreenDiag], para.firstTauIdx, TauNum)
    LfirstTauIdx, G0firstTauIdx, RfirstTauIdx, GxfirstTauIdx = idx
    assert maxTau == maxVer4TauIdx(para)
    lPara = reconstruct(para, type=Ver4Diag, innerLoopNum=oL, firstLoopIdx=LfirstLoopIdx, firstTauIdx=LfirstTauIdx)
    rPara = reconstruct(para, type=Ver4Diag, innerLoopNum=oR, firstLoopIdx=RfirstLoopIdx, firstTauIdx=RfirstTauIdx)
    gxPara = reconstruct(para, type=GreenDiag, innerLoopNum=oGx, firstLoopIdx=GxfirstLoopIdx, firstTauIdx=GxfirstTauIdx)
    g0Para = reconstruct(para, type=GreenDiag, innerLoopNum=oG0, firstLoopIdx=G0firstLoopIdx, firstTauIdx=G0firstTauIdx)
    if chan in (PHr, PHEr):
        Gi = blockstoplevel.phi if level == 1 else blocks.phi
    elif chan == PPr:
        Gi = blockstoplevel.ppi if level == 1 else blocks.ppi
    else:
        raise ValueError(f"chan {chan} isn't implemented!")
    Gf = blockstoplevel.G4 if level == 1 else blocks.G4
    LLegK, K, RLegK, Kx = legBasis(chan, legK, LoopIdx)
    Lver = vertex4(lPara, LLegK, True, channels=Gi, level=level + 1, name="Gi", blocks=blocks)
    if not Lver:
        return
    Rver = vertex4(rPara, RLegK, True, channels=Gf, level=level + 1, name="Gf", blocks=blocks)
    if not Rver:
        return
    ver8: Dict[tuple, List[Graph]] = {}
    for l in Lver:
        for r in Rver:
            bubble2diag(ver8, para, chan, l["diagram"], r["diagram"], legK, extrafactor)
    for key, terms in ver8.items():
        G0T, GxT, extT, Vresponse, vtype = key
        if not terms:
            continue
        g0 = green(g0Para, K, G0T, True, name="G0", blocks=blocks)
        gx = green(gxPara, Kx, GxT, True, name="Gx", blocks=blocks)
        id = Ver4Id(para, Vresponse, vtype, k=legK, t=extT, chan=chan)
        first = terms[0] if len(terms) == 1 else Graph(terms, properties=GenericId(para), operator=Sum())
        ver4df.append(dict(response=Vresponse, type=vtype, extT=extT, diagram=Graph([first, g0, gx], properties=id, operator=Prod())))


def RPA_chain(ver4df, para, legK, chan, level, name, extrafactor=1.0):          # vertex4.jl:180-190
    if chan not in (PHr, PHEr):
        return
    new_filter = tuple(para.filter) + tuple(f for f in (Girreducible, DirectOnly) if f not in para.filter)
    blocks = ParquetBlocks(phi=(), ppi=(), G4=(PHr,))
    bubble(ver4df, reconstruct(para, filter=new_filter), legK, chan, [0, 0, para.innerLoopNum - 1, 0], level, f"{name}_RPA_CT", blocks, blocks, extrafactor)


def vertex4(para: DiagPara, extK=None, subdiagram: bool = False, *, channels=(PHr, PHEr, PPr, Alli), level: int = 1,
            name: str = "none", resetuid: bool = False, blocks: Optional[ParquetBlocks] = None,
            blockstoplevel: Optional[ParquetBlocks] = None) -> List[Row]:
    """vertex4.jl:27-108.  Returns the table as a list of rows ``{response, type, extT, diagram, hash}``."""
    blocks = blocks or ParquetBlocks()
    blockstoplevel = blockstoplevel or blocks
    if extK is None:
        extK = [getK(para.totalLoopNum, 1), getK(para.totalLoopNum, 2), getK(para.totalLoopNum, 3)]
    for k in extK:
        assert len(k) >= para.totalLoopNum
    legK = [list(k[:para.totalLoopNum]) for k in extK[:3]]
    legK.append(_vadd(_vadd(legK[0], legK[2]), legK[1], 1.0, -1.0))
    assert para.totalTauNum >= maxVer4TauIdx(para), "Increase totalTauNum!"
    assert para.totalLoopNum >= maxVer4LoopIdx(para), "Increase totalLoopNum"
    for b in (blocks, blockstoplevel):
        assert PHr not in b.phi and PPr not in b.ppi
    loopNum = para.innerLoopNum
    ver4df: List[Row] = []
    if loopNum == 0:
        bareVer4(ver4df, para, legK, (Di,) if DirectOnly in para.filter else (Di, Ex))
    else:
        for c in channels:
            if c == Alli:
                if 3 <= loopNum <= 4:
                    addAlli(ver4df, para, legK)
                continue
            for p in orderedPartition(loopNum - 1, 4, 0):
                if c in (PHr, PHEr, PPr):
                    bubble(ver4df, para, legK, c, p, level, name, blocks, blockstoplevel, 1.0)
            if NoBubble in para.filter and c in (PHr, PHEr):
                RPA_chain(ver4df, para, legK, c, level, name, -1.0)
    if ver4df:                                            # merge_vertex4, vertex4.jl:97-110
        ver4df = mergeby(ver4df, ["response", "type", "extT"], name=name,
                         getid=lambda g: Ver4Id(para, g[0]["response"], g[0]["type"], k=legK, t=g[0]["extT"]))
    assert all(r["extT"][0] == para.firstTauIdx for r in ver4df)
    return ver4df


# --- the fully irreducible vertex from the GV catalogs (parquet.jl:216-231, vertex4.jl:112-120, operation.jl:178-257) ---------
@functools.lru_cache(maxsize=None)
def get_ver4I(order: int):
    """``vertex4I_diags[order]`` = ``diagsGV_ver4(order, channels=[Alli], filter=[NoHartree])`` for order 3 and 4, from the
    catalog numbers shipped in data/vertex4I<order>.npz (tests/golden/make_vertex4_catalogs.py)."""
    import numpy as np
    from .gv import read_vertex4diagrams
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", f"vertex4I{order}.npz")
    return read_vertex4diagrams(dict(np.load(path)), 0.0, (NoHartree,), (Alli,))


def _preorder(g: Graph):
    stack = [g]
    while stack:
        n = stack.pop()
        yield n
        stack.extend(reversed(n.subgraphs))


def update_extKT(diags: Sequence[Graph], para: DiagPara, legK, extraLoopIdx: Optional[int] = None) -> List[Graph]:
    """operation.jl:178-257: a copy of the catalog's graphs re-expressed in the caller's loop basis and time indices.  Vertex
    ids take the caller's legs; a propagator's momentum, written in the catalog's basis (three external loops, then the
    inner ones), becomes sum_i K_i legK_i plus its inner components moved to the positions the external legs do not use.
    As in the reference, an id whose time indices do not shift keeps the momentum as computed, one that shifts is rebuilt
    through its constructor (which applies the mirror symmetry)."""
    graphs = copy.deepcopy(list(diags))
    visited = set()
    tauIdx = para.firstTauIdx
    n = len(legK[0])
    extK = [list(k) for k in legK[:-1]]
    for graph in graphs:
        tau_shift = tauIdx - graph.properties.extT[0]
        for node in _preorder(graph):
            if id(node) in visited:
                continue
            visited.add(id(node))
            node.id = uid()
            prop = node.properties
            if prop is None or not hasattr(prop, "extK") or not hasattr(prop, "extT"):
                continue
            T = tuple(t + tau_shift for t in prop.extT)
            if isinstance(prop, (Ver4Id, Ver3Id)):
                node.properties = (Ver4Id(para, prop.response, prop.type, k=[k[:n] for k in legK], t=T, chan=prop.channel)
                                   if isinstance(prop, Ver4Id) else Ver3Id(para, prop.response, k=[k[:n] for k in legK[:3]], t=T))
            elif isinstance(prop, (BareGreenId, BareInteractionId, GreenId, SigmaId, PolarId)):
                K = [float(x) for x in prop.extK]
                orig = len(K)
                if orig < n:
                    K += [0.0] * (n - orig)
                    if extraLoopIdx is not None:
                        K[-1] = K[extraLoopIdx - 1]
                        K[extraLoopIdx - 1] = 0.0
                else:
                    K = K[:n]
                sumK = [0.0] * n
                for i, k in enumerate(extK):
                    sumK = [a + K[i] * b for a, b in zip(sumK, k)]
                chosen: List[int] = []
                for i in sorted(range(len(extK)), key=lambda i: sum(1 for x in extK[i] if x != 0)):
                    j = next(idx for idx in range(n) if idx not in chosen and extK[i][idx] != 0)
                    chosen.append(j)
                    K[i], K[j] = K[j], K[i]
                newK = [sumK[idx] + (K[idx] if idx not in chosen else 0.0) for idx in range(n)]
                sym = tau_shift != 0
                if isinstance(prop, BareGreenId):
                    node.properties = BareGreenId(k=newK, t=T, type=prop.type, symmetrize=sym)
                elif isinstance(prop, BareInteractionId):
                    node.properties = BareInteractionId(prop.response, k=newK, t=T, type=prop.type, symmetrize=sym)
                else:
                    raise NotImplementedError("composite ids do not occur in the catalog graphs")
    return graphs


def addAlli(ver4df: List[Row], para: DiagPara, legK) -> None:          # vertex4.jl:112-120
    for g in update_extKT(get_ver4I(para.innerLoopNum), para, legK, para.firstLoopIdx - 1):
        Id = g.properties
        ver4df.append(dict(response=Id.response, type=Id.type, extT=Id.extT, diagram=g))


# --- green.jl ---------------------------------------------------------------------------------------------------------------
def green(para: DiagPara, extK=None, extT=None, subdiagram: bool = False, *, name: str = "G",
          blocks: Optional[ParquetBlocks] = None) -> Graph:
    """green.jl:21-113: G = g0 * sum over (Sigma, G) splittings of Sigma * G."""
    blocks = blocks or ParquetBlocks()
    extK = list(extK if extK is not None else getK(para.totalLoopNum, 1))
    extT = tuple(extT if extT is not None else ((1, 2) if para.hasTau else (0, 0)))
    assert isValidG(para) and para.type == GreenDiag and para.innerLoopNum >= 0 and len(extT) == 2
    assert len(extK) >= para.totalLoopNum
    extK = extK[:para.totalLoopNum]
    tin, tout = extT
    t0 = para.firstTauIdx
    if para.innerLoopNum == 0:
        return Graph([], properties=BareGreenId(k=extK, t=extT), name=name)
    g0 = Graph([], properties=BareGreenId(k=extK, t=(tin, t0)), name="g0")
    pairs: List[Graph] = []
    for p in orderedPartition(para.innerLoopNum, 2, 0):
        oS, oG = p
        if not isValidSigma(para.filter, oS, True) or not isValidG(para.filter, oG):
            continue
        idx, maxTau = findFirstTauIdx(p, [SigmaDiag, GreenDiag], t0, _interactionTauNum(para))
        assert maxTau <= para.totalTauNum
        if para.hasTau:
            assert (tin < t0 or tin > maxTau) and (tout < t0 or tout > maxTau)
        SfirstTidx, GfirstTidx = idx
        idx, maxLoop = findFirstLoopIdx(p, para.firstLoopIdx)
        assert maxLoop <= para.totalLoopNum
        SfirstKidx, GfirstKidx = idx
        sigmaPara = reconstruct(para, type=SigmaDiag, firstTauIdx=SfirstTidx, firstLoopIdx=SfirstKidx, innerLoopNum=oS)
        sig = sigma(sigmaPara, extK, True, name="Sigma", blocks=blocks)
        assert all(r["extT"][0] == SfirstTidx for r in sig)
        df = [dict(r, Tin=r["extT"][0], GT=(r["extT"][1], extT[1])) for r in sig]
        for g in mergeby(df, "GT", operator=Sum()):
            paraG = reconstruct(para, type=GreenDiag, firstTauIdx=GfirstTidx, firstLoopIdx=GfirstKidx, innerLoopNum=oG)
            G = green(paraG, extK, g["GT"], True, blocks=blocks)
            pairs.append(Graph([g["diagram"], G], properties=GenericId(para, (("t", (SfirstTidx, g["GT"][1])),)), operator=Prod(), name="SigmaG"))
    merged = mergeby(pairs, operator=Sum(), name="gSigmaG")[0]
    return Graph([g0, merged], properties=GreenId(para, k=extK, t=extT), operator=Prod(), name=name)


# --- sigma.jl ---------------------------------------------------------------------------------------------------------------
def sigma(para: DiagPara, extK=None, subdiagram: bool = False, *, name: str = "Sigma",
          blocks: Optional[ParquetBlocks] = None) -> List[Row]:
    """sigma.jl:19-136.  Returns rows ``{type, extT, diagram, hash}`` (instantaneous part first, then the dynamic parts
    by outgoing time)."""
    blocks = blocks or ParquetBlocks()
    extK = list(extK if extK is not None else getK(para.totalLoopNum, 1))
    if para.type != SigmaDiag:
        raise ValueError(f"{para} is not for a sigma diagram")
    if para.innerLoopNum < 1:
        raise ValueError("sigma must has more than one inner loop")
    if len(extK) < para.totalLoopNum:
        raise ValueError(f"expect dim of extK>={para.totalLoopNum}, got {len(extK)}")
    extK = extK[:para.totalLoopNum]
    composite: List[Row] = []
    if not isValidSigma(para.filter, para.innerLoopNum, subdiagram):
        return composite
    LoopIdx = para.firstLoopIdx
    K = [0.0] * len(extK)
    K[LoopIdx - 1] = 1.0
    if _isapprox_vec(K, extK):
        raise ValueError("K and extK can not be the same")
    legK = [extK, K, K, extK]

    def GW(group: Row, oW: int, paraG: DiagPara) -> Row:
        response, type = group["response"], group["type"]
        if response not in (UpUp, UpDown):
            raise ValueError("GW with given ExT to Sigma only works for UpUp or UpDown")
        sid = SigmaId(para, type, k=extK, t=group["extT"])
        g = green(paraG, K, group["GT"], True, name="Gfock" if oW == 0 else "G_Sigma", blocks=blocks)
        spinfactor = 2.0 if response == UpUp else -1.0   # Sigma = G (2 W_uu - W_ud)
        if oW > 0:
            spinfactor *= 0.5                             # composite Sigma: symmetry factor 1/2
        d = Graph.new([g, group["diagram"]], properties=sid, operator=Prod(), factor=spinfactor, name=name)
        return dict(type=type, extT=group["extT"], diagram=d)

    for oG, oW in orderedPartition(para.innerLoopNum - 1, 2, 0):
        idx, maxLoop = findFirstLoopIdx([oW, oG], LoopIdx + 1)
        if maxLoop > para.totalLoopNum:
            raise ValueError(f"maxLoop = {maxLoop} > {para.totalLoopNum}")
        WfirstLoopIdx, GfirstLoopIdx = idx
        idx, maxTau = findFirstTauIdx([oW, oG], [Ver4Diag, GreenDiag], para.firstTauIdx, _interactionTauNum(para))
        if maxTau > para.totalTauNum:
            raise ValueError(f"maxTau = {maxTau} > {para.totalTauNum}")
        WfirstTauIdx, GfirstTauIdx = idx
        paraG = reconstruct(para, type=GreenDiag, innerLoopNum=oG, firstLoopIdx=GfirstLoopIdx, firstTauIdx=GfirstTauIdx)
        paraW = reconstruct(para, type=Ver4Diag, innerLoopNum=oW, firstLoopIdx=WfirstLoopIdx, firstTauIdx=WfirstTauIdx)
        if not isValidG(paraG):
            continue
        if oW == 0:                                       # Fock-type
            if NoHartree in paraW.filter:
                f = tuple(paraW.filter) + ((Proper,) if Proper not in paraW.filter else ())
                ver4 = vertex4(reconstruct(paraW, filter=f, transferLoop=tuple([0.0] * len(K))), legK, True, channels=())
            else:
                ver4 = vertex4(paraW, legK, True, channels=())
        else:                                             # composite
            ver4 = vertex4(paraW, legK, True, channels=(PHr,), blocks=blocks,
                           blockstoplevel=ParquetBlocks(phi=(), G4=(PHr, PHEr, PPr, Alli)))
        df = [dict(r, extT=(r["extT"][INL], r["extT"][OUTR]), GT=(r["extT"][OUTL], r["extT"][INR])) for r in ver4]
        for merged in mergeby(df, ["response", "type", "GT", "extT"], operator=Sum()):
            composite.append(GW(merged, oW, paraG))
    if not composite:
        return composite
    out = mergeby(composite, ["type", "extT"], name=name, getid=lambda g: SigmaId(para, g[0]["type"], k=extK, t=g[0]["extT"]))
    if not all(r["extT"][0] == para.firstTauIdx for r in out):
        raise AssertionError("all sigma should share the same in Tidx")
    return out


# --- vertex3.jl -------------------------------------------------------------------------------------------------------------
def vertex3(para: DiagPara, _extK=None, subdiagram: bool = False, *, name: str = "Gamma3", channels=(PHr, PHEr, PPr, Alli),
            blocks: Optional[ParquetBlocks] = None) -> List[Row]:
    """vertex3.jl:21-112: Gamma3 = G_in G_out Gamma4.  Rows ``{response, extT, diagram, hash}``."""
    blocks = blocks or ParquetBlocks()
    if _extK is None:
        _extK = [getK(para.totalLoopNum, 1), getK(para.totalLoopNum, 2)]
    assert para.type == Ver3Diag and para.innerLoopNum >= 1
    for k in _extK:
        assert len(k) >= para.totalLoopNum
    q, Kin = list(_extK[0][:para.totalLoopNum]), list(_extK[1][:para.totalLoopNum])
    Kout = _vadd(Kin, q, 1.0, -1.0)
    assert not _isapprox_vec(q, Kin) and not _isapprox_vec(q, Kout)
    extK = [q, Kin, Kout]
    if Proper in para.filter and (len(para.transferLoop) != len(q) or not _isapprox_vec(para.transferLoop, q)):   # _properVer3Para
        para = reconstruct(para, transferLoop=tuple(q))
    t0 = para.firstTauIdx
    out: List[Row] = []
    LoopIdx = para.firstLoopIdx
    K = [0.0] * len(q)
    K[LoopIdx - 1] = 1.0
    Kq = _vadd(K, q)
    legK = [Kin, Kout, K, Kq]
    for oVer4, oGin, oGout in orderedPartition(para.innerLoopNum - 1, 3, 0):
        idx, maxLoop = findFirstLoopIdx([oVer4, oGin, oGout], LoopIdx + 1)
        assert maxLoop <= para.totalLoopNum
        Ver4Kidx, GinKidx, GoutKidx = idx
        ver4t0 = para.firstTauIdx + 1 if para.hasTau else para.firstTauIdx
        idx, maxTau = findFirstTauIdx([oVer4, oGin, oGout], [Ver4Diag, GreenDiag, GreenDiag], ver4t0, _interactionTauNum(para))
        assert maxTau <= para.totalTauNum
        Ver4Tidx, GinTidx, GoutTidx = idx
        if not (isValidG(para.filter, oGin) and isValidG(para.filter, oGout)):
            continue
        paraGin = reconstruct(para, type=GreenDiag, innerLoopNum=oGin, firstLoopIdx=GinKidx, firstTauIdx=GinTidx)
        paraGout = reconstruct(para, type=GreenDiag, innerLoopNum=oGout, firstLoopIdx=GoutKidx, firstTauIdx=GoutTidx)
        paraVer4 = reconstruct(para, type=Ver4Diag, innerLoopNum=oVer4, firstLoopIdx=Ver4Kidx, firstTauIdx=Ver4Tidx)
        ver4 = vertex4(paraVer4, legK, True, channels=channels, blocks=blocks)
        if not ver4:
            continue
        if para.hasTau:
            assert all(r["extT"][INL] == ver4t0 for r in ver4)
        df = [dict(r, extT=(t0, r["extT"][INL], r["extT"][OUTL]), GinT=(t0, r["extT"][INR]), GoutT=(r["extT"][OUTR], t0)) for r in ver4]
        for v4 in mergeby(df, ["response", "GinT", "GoutT", "extT"], operator=Sum()):
            response = v4["response"]
            assert response in (UpUp, UpDown)
            gin = green(paraGin, K, v4["GinT"], True, name="Gin", blocks=blocks)
            gout = green(paraGout, Kq, v4["GoutT"], True, name="Gout", blocks=blocks)
            d = Graph([gin, gout, v4["diagram"]], properties=Ver3Id(para, response, k=extK, t=v4["extT"]), operator=Prod(), name=name)
            out.append(dict(response=response, extT=v4["extT"], diagram=d))
    if out:
        out = mergeby(out, ["response", "extT"], name=name, getid=lambda g: Ver3Id(para, g[0]["response"], k=extK, t=g[0]["extT"]))
    return out


# --- polarization.jl --------------------------------------------------------------------------------------------------------
def polarization(para: DiagPara, extK=None, subdiagram: bool = False, *, name: str = "Pi",
                 blocks: Optional[ParquetBlocks] = None) -> List[Row]:
    """polarization.jl:17-127: Pi = -G G (the bare bubble) + G G Gamma3.  Rows ``{response, extT, diagram, hash}``."""
    blocks = blocks or ParquetBlocks()
    extK = list(extK if extK is not None else getK(para.totalLoopNum, 1))
    assert para.type == PolarDiag and para.innerLoopNum >= 1 and len(extK) >= para.totalLoopNum
    if Proper not in para.filter or len(para.transferLoop) != len(extK) or _isapprox_vec(para.transferLoop, extK):   # _properPolarPara
        para = reconstruct(para, transferLoop=tuple(extK), filter=(Proper,) + tuple(f for f in para.filter if f != Proper))
    extK = extK[:para.totalLoopNum]
    LoopIdx = para.firstLoopIdx
    K = [0.0] * len(extK)
    K[LoopIdx - 1] = 1.0
    assert not _isapprox_vec(K, extK)
    t0 = para.firstTauIdx
    extT = (t0, t0 + 1) if para.hasTau else (t0, t0)
    KmQ = _vadd(K, extK, 1.0, -1.0)
    legK = [extK, K, KmQ]
    polar: List[Row] = []
    for oVer3, oGin, oGout in orderedPartition(para.innerLoopNum - 1, 3, 0):
        idx, maxLoop = findFirstLoopIdx([oVer3, oGin, oGout], LoopIdx + 1)
        assert maxLoop <= para.totalLoopNum
        Ver3Kidx, GinKidx, GoutKidx = idx
        if not (isValidG(para.filter, oGin) and isValidG(para.filter, oGout)):
            continue
        if oVer3 == 0:                                    # Pi0 = G G
            gt0 = extT[1] + 1 if para.hasTau else extT[0]
            idx, maxTau = findFirstTauIdx([oGin, oGout], [GreenDiag, GreenDiag], gt0, _interactionTauNum(para))
            assert maxTau <= para.totalTauNum
            GinTidx, GoutTidx = idx
            paraGin = reconstruct(para, type=GreenDiag, innerLoopNum=oGin, firstLoopIdx=GinKidx, firstTauIdx=GinTidx)
            paraGout = reconstruct(para, type=GreenDiag, innerLoopNum=oGout, firstLoopIdx=GoutKidx, firstTauIdx=GoutTidx)
            gin = green(paraGin, K, (extT[0], extT[1]), True, name="Gin")
            gout = green(paraGout, KmQ, (extT[1], extT[0]), True, name="Gout")
            d = Graph.new([gin, gout], properties=PolarId(para, UpUp, k=extK, t=extT), operator=Prod(), name=name,
                          factor=-1.0 if para.isFermi else 1.0)
            polar.append(dict(response=UpUp, extT=extT, diagram=d))
        else:                                             # composite polarization
            idx, maxTau = findFirstTauIdx([oVer3, oGin, oGout], [Ver3Diag, GreenDiag, GreenDiag], extT[1], _interactionTauNum(para))
            assert maxTau <= para.totalTauNum
            Ver3Tidx, GinTidx, GoutTidx = idx
            paraGin = reconstruct(para, type=GreenDiag, innerLoopNum=oGin, firstLoopIdx=GinKidx, firstTauIdx=GinTidx)
            paraGout = reconstruct(para, type=GreenDiag, innerLoopNum=oGout, firstLoopIdx=GoutKidx, firstTauIdx=GoutTidx)
            paraVer3 = reconstruct(para, type=Ver3Diag, innerLoopNum=oVer3, firstLoopIdx=Ver3Kidx, firstTauIdx=Ver3Tidx)
            ver3 = vertex3(paraVer3, legK, True, blocks=blocks)
            if not ver3:
                continue
            if para.hasTau:
                assert all(r["extT"][0] == extT[1] for r in ver3) and all(r["extT"][1] == ver3[0]["extT"][1] for r in ver3)
            df = [dict(r, extT=extT, GinT=(extT[0], r["extT"][1]), GoutT=(r["extT"][2], extT[0])) for r in ver3]
            for v3 in mergeby(df, ["response", "GinT", "GoutT", "extT"], operator=Sum()):
                response = v3["response"]
                assert response in (UpUp, UpDown)
                gin = green(paraGin, K, v3["GinT"], True, name="Gin", blocks=blocks)
                gout = green(paraGout, KmQ, v3["GoutT"], True, name="Gout", blocks=blocks)
                d = Graph([gin, gout, v3["diagram"]], properties=PolarId(para, response, k=extK, t=v3["extT"]), operator=Prod(), name=name)
                polar.append(dict(response=response, extT=v3["extT"], diagram=d))
    if polar:
        polar = mergeby(polar, ["response", "extT"], name=name, getid=lambda g: PolarId(para, g[0]["response"], k=extK, t=extT))
    return polar


def build(para: DiagPara, extK=None, subdiagram: bool = False, *, channels=(PHr, PHEr, PPr, Alli)) -> List[Row]:
    """common.jl:1-28."""
    if para.type == Ver4Diag:
        return vertex4(para, extK, subdiagram, channels=channels)
    if para.type == SigmaDiag:
        return sigma(para, extK if extK is not None else getK(para.totalLoopNum, 1), subdiagram)
    if para.type == PolarDiag:
        return polarization(para, extK if extK is not None else getK(para.totalLoopNum, 1), subdiagram)
    if para.type == Ver3Diag:
        return vertex3(para, extK if extK is not None else [getK(para.totalLoopNum, 1), getK(para.totalLoopNum, 2)], subdiagram, channels=channels)
    raise NotImplementedError("not implemented!")


def count_ver3_G2v(innerLoopNum: int, spin: int) -> int:
    """benchmark/diagram_count.jl:23-36."""
    return {0: 1, 1: 1, 2: 4 + 3 * spin, 3: 27 + 31 * spin + 5 * spin ** 2}[innerLoopNum]


def count_polar_G2v(innerLoopNum: int, spin: int) -> int:
    """benchmark/diagram_count.jl:73-76."""
    return spin * count_ver3_G2v(innerLoopNum - 1, spin)


def count_sigma_G2v(innerLoopNum: int, spin: int) -> int:
    """benchmark/diagram_count.jl:53-66: number of self-energy diagrams of the G^2 v expansion."""
    return {1: 1, 2: 1 + spin, 3: 4 + 5 * spin + spin ** 2, 4: 27 + 40 * spin + 14 * spin ** 2 + spin ** 3}[innerLoopNum]
