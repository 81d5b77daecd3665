"""SURVEY.md 8a row a10, second method: ``leafstates(leaf_maps, labelProd::LabelProduct)`` for FeynmanGraph
leaves, and the pieces it reads -- LabelProduct, the operator statistics, the propagator / interaction leaf
builders.  Known answers are the reference's own: test/front_end.jl:38-68 (LabelProduct),
test/quantum_operator.jl:1-75 (products, orderings, parity), test/computational_graph.jl:242-246 and the
``group`` docstring feynmangraph.jl:636-653 (a propagator f+(1) f-(2) carries factor -1 and external order
[f-(2), f+(1)])."""
import pytest

import feynmandiagram_jl_amd as fd
from feynmandiagram_jl_amd import graph as G
from feynmandiagram_jl_amd.frontends import leafstates
from feynmandiagram_jl_amd.labelproduct import LabelProduct
from feynmandiagram_jl_amd.lowering import lower
from feynmandiagram_jl_amd.quantum_operators import (OperatorProduct, QuantumOperator, b_minus, b_plus, correlator_order, f_minus,
                                                     f_plus, iscreation, isfermionic, majorana, normal_order, parity, phi)


def test_label_product_kats():
    flavors, taus = [1, 2, 3], [1, 2, 3, 4, 5]
    loopbasis = [[1.0, 1.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0], [1.0, 0.0, -1.0, 0.0]]
    lp = LabelProduct(flavors, taus, loopbasis)
    assert len(lp) == 3 * 5 * 5 and lp.size() == (3, 5, 5)
    assert lp.index_to_linear(2, 4, 3) == 2 + 3 * 3 + 3 * 5 * 2
    assert lp.linear_to_index(41) == (2, 4, 3)
    assert lp[38] == lp[2, 3, 3] == (2, 3, [0.0, 0.0, 1.0, 0.0])
    assert lp.push_labelat(6, 2) == 6 and lp.labels[1] == [1, 2, 3, 4, 5, 6]
    assert lp.push_labelat([1.0, 0.0, 1.0, 0.0], 3) == 6 and lp.labels[-1][-1] == [1.0, 0.0, 1.0, 0.0]
    assert lp.push_labelat([1.0, 0.0, -1.0, 0.0], 3) == 5 and lp.labels[-1][-1] == [1.0, 0.0, 1.0, 0.0]
    assert lp.append_label([4, 2, [1.0, 0.0, 0.0, 0.0]]) == (4, 2, 7)
    assert lp.labels[0] == [1, 2, 3, 4] and lp.labels[1] == [1, 2, 3, 4, 5, 6] and len(lp.labels[2]) == 7
    assert lp.labels[2][-1] == [1.0, 0.0, 0.0, 0.0]


def test_operator_products_and_statistics():
    assert majorana(1) == OperatorProduct([QuantumOperator("f", 1)])
    assert isfermionic(majorana(1)[0]) and isfermionic(f_plus(1)[0]) and isfermionic(f_minus(1)[0])
    assert iscreation(f_plus(1)[0]) and iscreation(b_plus(1)[0])
    assert f_minus(1)[0].adjoint == f_plus(1)[0]
    qe1 = OperatorProduct([QuantumOperator("f+", 1), QuantumOperator("f-", 2), QuantumOperator("phi", 3)])
    qe2 = OperatorProduct(list(qe1) + [QuantumOperator("f-", 4)])
    qe3 = OperatorProduct([QuantumOperator("b-", 4)] + list(qe1))
    assert f_plus(1) * f_minus(2) * phi(3) == qe1
    assert qe1 * f_minus(4) == qe2 and qe1 * QuantumOperator("f-", 4) == qe2
    assert QuantumOperator("b-", 4) * qe1 == qe3
    assert not isfermionic(qe1) and isfermionic(qe2) and not isfermionic(qe3)
    assert qe1.adjoint == phi(3) * f_plus(2) * f_minus(1)
    assert qe3.adjoint == phi(3) * f_plus(2) * f_minus(1) * b_plus(4)


def test_orderings_and_parity():
    o1 = f_plus(1) * f_minus(2) * f_plus(5) * f_plus(6) * f_minus(1) * f_minus(5)
    sign, perm = correlator_order(o1)
    assert sign == 1 and o1[[p - 1 for p in perm]] == f_minus(1) * f_minus(5) * f_minus(2) * f_plus(6) * f_plus(5) * f_plus(1)
    sign, perm = normal_order(o1)
    assert sign == -1 and o1[[p - 1 for p in perm]] == f_plus(1) * f_plus(5) * f_plus(6) * f_minus(2) * f_minus(5) * f_minus(1)
    o2 = f_plus(1) * f_minus(2) * b_plus(1) * phi(1) * f_plus(6) * f_plus(5) * f_minus(1) * f_minus(5) * b_minus(1)
    sign, perm = correlator_order(o2)
    assert sign == -1
    assert o2[[p - 1 for p in perm]] == f_minus(1) * b_minus(1) * f_minus(5) * f_minus(2) * phi(1) * f_plus(6) * f_plus(5) * b_plus(1) * f_plus(1)
    sign, perm = normal_order(o2)
    assert sign == 1
    assert o2[[p - 1 for p in perm]] == f_plus(1) * b_plus(1) * f_plus(5) * phi(1) * f_plus(6) * f_minus(2) * f_minus(5) * b_minus(1) * f_minus(1)
    o3 = f_plus(1) * f_minus(2) * b_plus(1) * phi(1) * f_plus(3) * f_minus(1) * majorana(1) * b_minus(1) * phi(1)
    sign, perm = correlator_order(o3)
    assert sign == -1
    assert o3[[p - 1 for p in perm]] == f_minus(1) * b_minus(1) * phi(1) * f_minus(2) * majorana(1) * f_plus(3) * phi(1) * b_plus(1) * f_plus(1)
    sign, perm = normal_order(o3)
    assert sign == -1
    assert o3[[p - 1 for p in perm]] == f_plus(1) * b_plus(1) * phi(1) * f_plus(3) * majorana(1) * f_minus(2) * phi(1) * b_minus(1) * f_minus(1)
    assert parity([1]) == 1 and parity([2, 3, 1, 5, 6, 4]) == 1 and parity([3, 4, 1, 2]) == 1 and parity([3, 5, 1, 2, 4, 6, 7]) == -1


def test_propagator_sign_and_external_order():
    g1 = G.propagator(f_plus(1) * f_minus(2))
    # computational_graph.jl:242-246: 1*g1 + 2*g1 merges to one child with factor -3 => g1 is a -1 wrapper
    assert isinstance(g1.operator, fd.Prod) and g1.subgraph_factors == [-1.0]
    inner = g1.subgraphs[0]
    assert inner.properties.external_indices == [2, 1] and inner.properties.external_legs == [True, True]
    assert [v[0] for v in inner.properties.vertices] == [QuantumOperator("f+", 1), QuantumOperator("f-", 2)]
    h = G.linear_combination([g1, g1], [1, 2])
    assert h.subgraph_factors == [-3.0]
    assert G.propagator(f_minus(1) * f_plus(2)).subgraph_factors == [1.0] or not G.propagator(f_minus(1) * f_plus(2)).subgraphs
    with pytest.raises(AssertionError):
        G.propagator(f_plus(1) * f_plus(2))
    with pytest.raises(AssertionError):
        G.interaction(list(f_plus(1) * b_minus(2)))


def test_leafstates_feynmangraph_method():
    """frontends.jl:115-160 on a hand-made partition: labels are linear indices into
    LabelProduct(taus, loopbasis); leaf order is the compiler's (to_julia_str first visit)."""
    taus = [1, 2, 3]
    loopbasis = [[1.0, 0.0], [0.0, 1.0], [1.0, -1.0]]
    lp = LabelProduct(taus, loopbasis)
    lab = lambda tau, k: lp.index_to_linear(tau, k)
    gf = G.propagator(f_minus(lab(2, 3)) * f_plus(lab(1, 3)), orders=[0, 0])          # out = tau 2, in = tau 1, momentum 3
    gb = G.propagator(phi(lab(3, 2)) * phi(lab(2, 2)), orders=[0, 1])                   # bosonic: out tau 3, in tau 2, momentum 2
    gi = G.interaction(list(phi(lab(3, 1)) * phi(lab(3, 1))), orders=[1, 0])            # vertex at tau 3
    leaves = [x.subgraphs[0] if x.subgraphs else x for x in (gf, gb, gi)]
    root = fd.FeynmanGraph(leaves, operator=fd.Prod(), orders=[1, 1])
    table, leafmap, _ = lower([root])
    assert [leafmap[i].id for i in (1, 2, 3)] == [x.id for x in leaves]
    val, typ, orders, tin, tout, loop = leafstates([leafmap], lp)
    assert val == [[1.0, 1.0, 1.0]]
    assert typ == [[1, 2, 0]]
    assert orders == [[[0, 0], [0, 1], [1, 0]]]
    assert tin == [[1, 2, 3]] and tout == [[2, 3, 3]]
    assert loop == [[3, 2, 1]]
    bad = G.external_vertex(f_plus(1))
    with pytest.raises(NameError):
        leafstates([{1: bad}], lp)
