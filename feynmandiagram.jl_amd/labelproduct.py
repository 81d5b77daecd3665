"""``LabelProduct`` (src/frontend/LabelProduct.jl): a Cartesian product of label vectors addressed by
one linear, column-major, 1-based index.  The ``FeynmanGraph`` method of ``leafstates`` reads a leaf's
time and loop-basis labels through it.  KATs: test/front_end.jl:38-68."""
from __future__ import annotations

from typing import List, Sequence, Tuple

__all__ = ["LabelProduct"]


class LabelProduct:
    def __init__(self, *labels: Sequence):
        assert all(isinstance(v, (list, tuple)) for v in labels), "all arguments should be vectors or tuples."
        self.labels: Tuple[list, ...] = tuple(list(v) for v in labels)      # LabelProduct.jl:19-24
        self.dims: Tuple[int, ...] = tuple(len(v) for v in labels)

    def __len__(self) -> int:                                              # :32
        n = 1
        for d in self.dims:
            n *= d
        return n

    def size(self, i: int = None):                                        # :38-44
        return self.dims if i is None else self.dims[i - 1]

    def index_to_linear(self, *I: int) -> int:                            # :63-69
        ex = I[-1] - 1
        for i in range(len(self.dims) - 2, -1, -1):
            ex = I[i] - 1 + self.dims[i] * ex
        return ex + 1

    def linear_to_index(self, I: int) -> Tuple[int, ...]:                 # :90-98
        q = I - 1
        out: List[int] = []
        for d in self.dims[:-1]:
            out.append(q % d + 1)
            q //= d
        out.append(q + 1)
        return tuple(out)

    def __getitem__(self, index):                                         # :113-120
        if isinstance(index, int):
            index = self.linear_to_index(index)
        return tuple(self.labels[i][index[i] - 1] for i in range(len(self.dims)))

    def push_labelat(self, new_label, dim: int) -> int:                   # :141-150
        assert dim <= len(self.dims)
        lab = self.labels[dim - 1]
        for k, v in enumerate(lab):
            if v == new_label:
                return k + 1
        lab.append(new_label)
        self.dims = tuple(d + 1 if i == dim - 1 else d for i, d in enumerate(self.dims))
        return self.dims[dim - 1]

    def append_label(self, new_label: Sequence) -> Tuple[int, ...]:       # :152-171
        if len(new_label) != len(self.dims):
            raise ValueError("Length of new_label must match the existing number of dimensions (N)")
        locs = list(self.dims)
        for dim, label in enumerate(new_label):
            lab = self.labels[dim]
            found = next((k + 1 for k, v in enumerate(lab) if v == label), None)
            if found is None:
                lab.append(label)
                locs[dim] += 1
            else:
                locs[dim] = found
        self.dims = tuple(max(d, self.dims[i]) for i, d in enumerate(locs))
        return tuple(locs)

    def __repr__(self):
        return f"LabelProduct of: {self.labels}"
