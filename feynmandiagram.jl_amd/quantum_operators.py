"""Just enough of ``QuantumOperators`` (src/quantum_operator/operator.jl, expression.jl) for the leaf
builders ``propagator`` / ``interaction`` and the ``FeynmanGraph`` method of ``FrontEnds.leafstates``
(SURVEY.md 8a row a10): operators with a label, products of them, their statistics, and the two orderings
with their fermionic sign.  Pinned by the reference's own tests (test/quantum_operator.jl:30-75).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

__all__ = ["QuantumOperator", "OperatorProduct", "f_plus", "f_minus", "majorana", "b_plus", "b_minus", "phi",
           "isfermionic", "iscreation", "isannihilation", "adjoint", "parity", "normal_order", "correlator_order"]

# operator kinds (operator.jl:9-14) and their conjugates (:32-37)
FERMI_CREATION, FERMI_ANNIHILATION, MAJORANA = "f+", "f-", "f"
BOSON_CREATION, BOSON_ANNIHILATION, CLASSIC = "b+", "b-", "phi"
_ADJ = {FERMI_CREATION: FERMI_ANNIHILATION, FERMI_ANNIHILATION: FERMI_CREATION, MAJORANA: MAJORANA,
        BOSON_CREATION: BOSON_ANNIHILATION, BOSON_ANNIHILATION: BOSON_CREATION, CLASSIC: CLASSIC}


class QuantumOperator:
    """operator.jl:62-75: a kind and a non-negative integer label."""
    __slots__ = ("operator", "label")

    def __init__(self, operator: str, label: int):
        if operator not in _ADJ:
            raise ValueError(f"unknown operator kind {operator!r}")
        if label < 0:
            raise AssertionError("label >= 0")
        self.operator, self.label = operator, int(label)

    def __eq__(self, other):
        return isinstance(other, QuantumOperator) and self.operator == other.operator and self.label == other.label

    def __hash__(self):
        return hash((self.operator, self.label))

    def __repr__(self):
        return f"{self.operator}({self.label})"

    def __mul__(self, other):
        return OperatorProduct([self]) * other

    @property
    def adjoint(self) -> "QuantumOperator":
        return QuantumOperator(_ADJ[self.operator], self.label)


class OperatorProduct(list):
    """expression.jl:1-60: an ordered product of QuantumOperators."""

    def __init__(self, ops: Iterable = ()):
        if isinstance(ops, QuantumOperator):
            ops = [ops]
        super().__init__(ops)
        assert all(isinstance(o, QuantumOperator) for o in self)

    def __mul__(self, other):
        if isinstance(other, QuantumOperator):
            return OperatorProduct(list(self) + [other])
        return OperatorProduct(list(self) + list(other))

    def __getitem__(self, i):
        if isinstance(i, (list, tuple)):                 # o[perm], 0-based here
            return OperatorProduct([list.__getitem__(self, j) for j in i])
        r = list.__getitem__(self, i)
        return OperatorProduct(r) if isinstance(i, slice) else r

    def __hash__(self):
        return hash(tuple(self))

    @property
    def adjoint(self) -> "OperatorProduct":          # expression.jl:93-99: reversed product of adjoints
        return OperatorProduct([o.adjoint for o in reversed(self)])


def f_plus(i): return OperatorProduct([QuantumOperator(FERMI_CREATION, i)])
def f_minus(i): return OperatorProduct([QuantumOperator(FERMI_ANNIHILATION, i)])
def majorana(i): return OperatorProduct([QuantumOperator(MAJORANA, i)])
def b_plus(i): return OperatorProduct([QuantumOperator(BOSON_CREATION, i)])
def b_minus(i): return OperatorProduct([QuantumOperator(BOSON_ANNIHILATION, i)])
def phi(i): return OperatorProduct([QuantumOperator(CLASSIC, i)])


def isfermionic(o) -> bool:
    """operator.jl:40-41,92 for one operator; expression.jl:106-113 for a product: an odd number of
    fermionic factors."""
    if isinstance(o, QuantumOperator):
        return o.operator in (FERMI_CREATION, FERMI_ANNIHILATION, MAJORANA)
    return sum(1 for op in o if isfermionic(op)) % 2 == 1


def iscreation(o: QuantumOperator) -> bool:          # operator.jl:44-45
    return o.operator in (FERMI_CREATION, BOSON_CREATION)


def isannihilation(o: QuantumOperator) -> bool:      # operator.jl:48-49
    return o.operator in (FERMI_ANNIHILATION, BOSON_ANNIHILATION)


def adjoint(o):
    return o.adjoint


def parity(p: Sequence[int]) -> int:
    """expression.jl:194-210: sign of a permutation given 1-based images."""
    q = list(p)
    count = 0
    for i in range(len(q)):
        while q[i] != i + 1:
            count += 1
            j = q[i] - 1
            q[i], q[j] = q[j], q[i]
    return 1 - 2 * (count % 2)


def _sortperm(v: Sequence[int]) -> List[int]:
    return [i + 1 for i in sorted(range(len(v)), key=lambda k: v[k])]      # 1-based, stable like Julia's


def _order(operator: Sequence[QuantumOperator], first) -> Tuple[int, List[int]]:
    """Shared body of normal_order (expression.jl:121-150, ``first`` = not annihilation) and
    correlator_order (:158-188, ``first`` = not creation).  Returns (sign, permutation), 1-based."""
    num = len(operator)
    ind_pair, ind_unpair = 0, num + 1
    ordering: List[int] = []
    for i, op in enumerate(operator):
        adj = op.adjoint
        if adj in operator[i + 1:]:
            ind_pair += 1
            ordering.append(ind_pair if first(op) else num + 1 - ind_pair)
        elif adj in operator[:i]:
            last = max(k for k in range(i) if operator[k] == adj)
            ordering.append(num + 1 - ordering[last])
        else:
            ordering.append(ind_unpair if first(op) else -ind_unpair)
    n_first = n_second = 0
    for i, value in enumerate(ordering):
        if value == ind_unpair:
            n_first += 1
            ordering[i] = ind_pair + n_first
        elif value == -ind_unpair:
            n_second += 1
            ordering[i] = num + 1 - ind_pair - n_second
    permutation = [ordering[i] for i, op in enumerate(operator) if isfermionic(op)]
    sign = 1 if not permutation else parity(_sortperm(permutation))
    return sign, _sortperm(ordering)


def normal_order(operator: Sequence[QuantumOperator]) -> Tuple[int, List[int]]:
    return _order(list(operator), lambda op: not isannihilation(op))


def correlator_order(operator: Sequence[QuantumOperator]) -> Tuple[int, List[int]]:
    return _order(list(operator), lambda op: not iscreation(op))
