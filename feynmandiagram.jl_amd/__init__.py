"""MI355X-native evaluator back end for FeynmanDiagram.jl's static
computational graphs (package directory ``feynmandiagram.jl_amd``; import it as
``feynmandiagram_jl_amd`` through the loader module at the repository root).

Scope: the graph *evaluation* hot path only --
``Compilers.compile`` -> ``eval_graph!(root, leafVal)`` -- behind the C ABI of
include/fdg.h, plus the host-side mirror of the reference interface for it.
The restated producers of the benchmark graphs (Parquet / GV / optimize! / Taylor) live in ``producers/``: workload
generators for tests and bench.py, not product, and not exported here.
"""
from . import graph as ComputationalGraphs
from . import compilers as Compilers
from . import frontends as FrontEnds
from .graph import (FeynmanGraph, Graph, PostOrderDFS, Power, Prod, Sum, Unitary, constant_graph, eval_,
                    external_vertex, linear_combination, multi_product)
from .nodetable import NodeTable, synthetic_parquet_like, from_program
from .lowering import lower
from .compilers import GraphFunc, compile_table

__all__ = ["ComputationalGraphs", "Compilers", "FrontEnds", "Graph", "FeynmanGraph", "Sum", "Prod", "Power", "Unitary",
           "constant_graph", "eval_", "external_vertex", "linear_combination", "multi_product", "PostOrderDFS",
           "NodeTable", "synthetic_parquet_like", "from_program", "lower", "GraphFunc", "compile_table"]
