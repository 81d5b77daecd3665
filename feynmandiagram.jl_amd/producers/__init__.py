"""Workload producers -- test and benchmark infrastructure, NOT part of the product path.

The evaluator back end (the product: ``compilers``, ``lowering``, ``capi``, ``csrc/``) takes graphs as the
reference's front ends build them.  Julia cannot run here, so the graphs BASELINE.json names are produced by host-side
restatements of the reference's own producers, once per graph, off the device path:

  parquet   -- ``Parquet.build`` / ``vertex4`` / ``sigma`` / ``green`` / ``polarization`` / ``vertex3``
               (src/frontend/parquet/*.jl).  FROZEN: it exists to obtain configs 1-4 and example/benchmark.jl's graph verbatim.
  gv        -- the GV ``.diag`` catalog reader (src/frontend/GV_diagrams/readfile.jl) and the diagram ids.
  optimize  -- ``optimize!`` (src/computational_graph/optimize.jl, transform.jl).
  taylor    -- ``taylorAD`` (src/utility.jl, src/TaylorSeries/).

They are pinned by the reference's own known answers (tests/test_parquet.py, tests/test_next_rows.py) and used by
``workloads.py``, the tests and the fixture generators under tests/golden/.  Nothing here is exported from the package
root; the north star keeps the reference's front ends "untouched", and a Julia user keeps using them.
"""
