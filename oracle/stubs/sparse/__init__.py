"""Stand-in for the un-vendored `sparse` package (pydata/sparse; the reference
lists it unpinned in setup.py:38).  Only the COO surface the reference's DP path
touches is provided.  `COO @ ndarray` restates the package's published kernel
(`sparse._common._dot_coo_ndarray`): for every output row, products are added
one by one, in ascending coordinate order, into an accumulator of the result
dtype (float32 for float32 operands) -- multiply rounded, then add rounded.

Test infrastructure only: imported by oracle/gen_golden.py when the reference is
run in the development container."""
import numpy as np


class COO:
    __array_priority__ = 1000

    def __init__(self, coords, data=None, shape=None):
        if data is None:
            dense = np.asarray(coords)
            nz = np.nonzero(dense)
            self.coords = np.stack([np.asarray(c, np.int64) for c in nz]) if dense.ndim else np.zeros((0, 0), np.int64)
            self.data = dense[nz]
            self.shape = dense.shape
        else:
            coords = np.asarray(coords, np.int64).reshape(len(shape), -1)
            data = np.asarray(data)
            order = np.lexsort(coords[::-1])
            self.coords = coords[:, order]
            self.data = data[order]
            self.shape = tuple(shape)
        self.dtype = self.data.dtype

    # -- basic properties ---------------------------------------------------------
    @property
    def nnz(self):
        return int(self.data.shape[0])

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape))

    def todense(self):
        out = np.zeros(self.shape, self.dtype)
        out[tuple(self.coords)] = self.data
        return out

    def _with_data(self, data):
        new = COO.__new__(COO)
        new.coords, new.data, new.shape, new.dtype = self.coords, data, self.shape, data.dtype
        return new

    # -- scalar scaling -------------------------------------------------------------
    def _scale(self, other):
        other = np.asarray(other)
        if other.size != 1:
            raise NotImplementedError("stub COO only scales by scalars")
        return self._with_data(self.data * other.reshape(()).astype(other.dtype)[()])

    def __mul__(self, other):
        if np.asarray(other).size != 1:  # broadcasting product with a dense array: non-zeros times the matching entries
            return self._elemwise(other, np.multiply)
        return self._scale(other)

    def __add__(self, other):
        other = other.todense() if hasattr(other, "todense") else np.asarray(other)
        return _Dense(self.todense() + other)

    __radd__ = __add__

    __rmul__ = __mul__

    def __neg__(self):
        return self._with_data(-self.data)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if ufunc is np.multiply and method == "__call__" and len(inputs) == 2:
            other = inputs[0] if inputs[1] is self else inputs[1]
            return self.__mul__(other)
        if ufunc is np.add and method == "__call__" and len(inputs) == 2:
            other = inputs[0] if inputs[1] is self else inputs[1]
            return self.__add__(other)
        if ufunc is np.matmul and method == "__call__":
            return NotImplemented
        return NotImplemented

    def sum(self, axis=None):
        # the package reduces with np.add.reduceat over the coordinates sorted by the kept axes: the terms of one output
        # element are added one by one in ascending order of the reduced index, in the data's dtype -- what numpy does
        # for a dense reduction over a leading axis too (the zeros in between leave every partial sum unchanged)
        d = self.todense().sum(axis)
        return _Dense(d)

    # -- what the sparse diameter uses (hardness/measures/diameter.py:382-420) -------------------------------------
    def __getitem__(self, idx):
        return COO(self.todense()[idx])

    def reshape(self, shape):
        return COO(self.todense().reshape(shape))

    def _elemwise(self, other, op):
        return COO(op(self.todense(), np.asarray(other)))

    # -- matmul -----------------------------------------------------------------------
    def __matmul__(self, other):
        other = np.asarray(other)
        lead = self.shape[:-1]
        n_rows = int(np.prod(lead))
        k = self.shape[-1]
        # flatten leading coordinates into a row id (coords are lexicographically sorted)
        row = np.ravel_multi_index(tuple(self.coords[:-1]), lead) if len(lead) else np.zeros(self.nnz, np.int64)
        col = self.coords[-1]
        vec = other.reshape(k, -1)
        n_out = vec.shape[1]
        rdt = np.result_type(self.dtype, other.dtype)
        out = np.zeros((n_rows, n_out), rdt)
        # position of every stored element inside its row
        starts = np.searchsorted(row, np.arange(n_rows), side="left")
        counts = np.diff(np.append(starts, self.nnz))
        pos = np.arange(self.nnz) - starts[row]
        for j in range(int(counts.max()) if self.nnz else 0):
            sel = pos == j
            r = row[sel]
            prod = (self.data[sel, None].astype(rdt) * vec[col[sel]].astype(rdt)).astype(rdt)
            out[r] = (out[r] + prod).astype(rdt)
        if other.ndim == 1:
            return out.reshape(lead)
        return out.reshape(lead + (n_out,))


class _Dense:
    """Result wrapper so that `T.sum(-1).todense()` works like in the real package."""

    def __init__(self, a):
        self._a = a

    def todense(self):
        return self._a

    def min(self, axis=None):
        return self._a.min(axis)

    def max(self, axis=None):
        return self._a.max(axis)

    def __getitem__(self, idx):
        return self._a[idx]

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)
