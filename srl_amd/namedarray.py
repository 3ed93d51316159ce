"""NamedArray: the nested name -> array struct that is the sample layout of the hot path.

Own implementation of the semantics of the reference's ``base/namedarray.py`` (class at :221-538,
helpers at :541-692, wire formats at :24-46,102-218). Behaviour kept byte-for-byte where the
rollout -> GAE -> PPO path depends on it:

* fields are kept in *sorted* name order (reference ``namedarray.py:282``),
* integer / slice indexing is applied to every non-None leaf and returns the same class
  (``:323-352``), string indexing behaves like ``getattr``,
* item assignment broadcasts a scalar / array to all leaves or assigns leaf-wise from a
  NamedArray of identical structure (``:354-384``),
* arithmetic operators are leaf-wise and respect ``None`` (``:49-96``),
* ``recursive_apply`` / ``recursive_aggregate`` (``:598-660``), ``flatten`` /
  ``from_flattened`` with dotted keys (``:663-692``), ``from_dict`` (``:541-570``),
* ``dumps`` / ``loads``: the ``pickle_dict``, ``pickle`` and ``raw_bytes`` wire encodings
  (``:102-218``). The four blosc-compressed encodings need the ``blosc`` package, which this
  image does not have; they raise ``NotImplementedError`` rather than silently degrading.

Leaves are numpy arrays or torch tensors (host or device); nothing here touches their memory.
"""
import ast
import copy
import itertools
import operator
import pickle
import types
from typing import Callable, List, Tuple

import numpy as np

try:  # torch is optional for this module (only `array_like` / `length` look at tensors)
    import torch
    _TENSOR_TYPES = (np.ndarray, torch.Tensor)
except ImportError:  # pragma: no cover
    torch = None
    _TENSOR_TYPES = (np.ndarray,)

# wire-format tags, identical to the reference's NamedArrayEncodingMethod values (:41-48)
_TAG_PICKLE_DICT = b"0001"
_TAG_PICKLE = b"0002"
_TAG_RAW_BYTES = b"0003"
_COMPRESSED_TAGS = {b"0004": "raw_compress", b"0005": "compress_pickle", b"0006": "pickle_compress",
                    b"0007": "obs_compress", b"0008": "compress_except_policy_state"}


class NamedArrayLoadingError(Exception):
    pass


def _binary(op):

    def fn(self, other):
        other = self._match(other)
        out = {}
        for k, mine in self.items():
            theirs = other[k]
            out[k] = None if (mine is None or theirs is None) else op(mine, theirs)
        return NamedArray(**out)

    return fn


def _inplace(op):

    def fn(self, other):
        other = self._match(other)
        for k, mine in self.items():
            theirs = other[k]
            if mine is not None and theirs is not None:
                # numpy / torch in-place operators mutate the leaf; rebind for immutables
                setattr(self, k, op(mine, theirs))
        return self

    return fn


class NamedArray:
    """Nested struct of arrays with shared leading (time / batch) dimensions."""

    def __init__(self, **fields):
        object.__setattr__(self, "_fields", tuple(sorted(fields.keys())))
        object.__setattr__(self, "_NamedArray__metadata", types.MappingProxyType({}))
        for k, v in fields.items():
            object.__setattr__(self, k, v)

    # ---------------------------------------------------------------- metadata
    @property
    def metadata(self):
        return self.__metadata

    def register_metadata(self, **kwargs):
        clash = [k for k in kwargs if k in self._fields]
        if clash:
            raise KeyError("Keys of metadata should be different from data fields!")
        self.__metadata = types.MappingProxyType({**self.__metadata, **kwargs})

    def pop_metadata(self, key):
        d = dict(self.__metadata)
        value = d.pop(key)
        self.__metadata = types.MappingProxyType(d)
        return value

    def clear_metadata(self):
        self.__metadata = types.MappingProxyType({})

    # ---------------------------------------------------------------- dict-like protocol
    def keys(self):
        return iter(self._fields)

    def values(self):
        return (getattr(self, k) for k in self._fields)

    def items(self):
        return ((k, getattr(self, k)) for k in self._fields)

    def __iter__(self):
        return self.values()

    def __len__(self):
        return len(self._fields)

    def __contains__(self, key):
        return key in self._fields

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, NamedArray) else v) for k, v in self.items()}

    # ---------------------------------------------------------------- indexing
    def __getitem__(self, loc):
        if isinstance(loc, str):
            return getattr(self, loc)
        picked = {}
        for k, v in self.items():
            if v is None:
                picked[k] = None
                continue
            try:
                picked[k] = v[loc]
            except IndexError as e:
                raise Exception(f"IndexError occured when slicing `NamedArray`."
                                f"Field {k} with shape {v.shape} and slice {loc}.") from e
        return self.__class__(**picked)

    def _match(self, value):
        """Return something indexable by field name that lines up with self's fields."""
        if isinstance(value, NamedArray):
            if getattr(value, "_fields", None) != self._fields:
                raise ValueError('namedarray - set an item with a different data structure')
            return value
        return {k: (None if v is None else value) for k, v in self.items()}

    def __setitem__(self, loc, value):
        if isinstance(loc, str):
            setattr(self, loc, value)
            return
        value = self._match(value)
        for k, mine in self.items():
            theirs = value[k]
            if mine is None or theirs is None:
                continue
            try:
                mine[loc] = theirs
            except (ValueError, IndexError, TypeError) as e:
                raise Exception(f"Error occured occured in {self.__class__.__name__} when assigning value"
                                f" at field '{k}': {e}") from e

    # ---------------------------------------------------------------- pickling / copying
    def __getstate__(self):
        return {'__metadata': dict(**self.metadata), **dict(self.items())}

    def __setstate__(self, state):
        NamedArray.__init__(self, **{k: v for k, v in state.items() if k != '__metadata'})
        if state.get('__metadata') is not None:
            self.clear_metadata()
            self.register_metadata(**state['__metadata'])

    def __deepcopy__(self, memo=None):
        memo = {} if memo is None else memo
        clone = self.__class__.__new__(self.__class__)
        memo[id(self)] = clone
        NamedArray.__init__(clone, **{k: copy.deepcopy(v, memo) for k, v in self.items()})
        clone.register_metadata(**copy.deepcopy(dict(self.metadata), memo))
        return clone

    # ---------------------------------------------------------------- shape helpers / stats
    def length(self, dim=0):
        for v in self.values():
            if isinstance(v, _TENSOR_TYPES) and dim < v.ndim:
                return v.shape[dim]
        raise IndexError(f"No entry has shape on dim={dim}.")

    @property
    def shape(self):
        return recursive_apply(self, lambda x: x.shape).to_dict()

    def size(self):
        return self.shape

    def unique_of(self, field, exclude_values=(None,)):
        uniq = np.unique(self[field])
        uniq = uniq[np.isin(uniq, exclude_values, invert=True)]
        return uniq[0] if len(uniq) == 1 else None

    def _masked_stat(self, field, fn_all, fn_nan, ignore_negative):
        values = self[field]
        if len(values) == 0:
            return None
        if ignore_negative:
            return fn_nan(np.where(values >= 0, values, np.nan))
        return fn_all(values)

    def average_of(self, field, ignore_negative=True):
        return self._masked_stat(field, np.mean, np.nanmean, ignore_negative)

    def max_of(self, field, ignore_negative=True):
        return self._masked_stat(field, np.max, np.nanmax, ignore_negative)

    def min_of(self, field, ignore_negative=True):
        return self._masked_stat(field, np.min, np.nanmin, ignore_negative)

    def __str__(self):
        return f"{self.__class__.__name__}({', '.join(k + '=' + repr(v) for k, v in self.items())})"

    __repr__ = __str__

    __add__ = _binary(operator.add)
    __sub__ = _binary(operator.sub)
    __mul__ = _binary(operator.mul)
    __truediv__ = _binary(operator.truediv)
    __iadd__ = _inplace(operator.iadd)
    __isub__ = _inplace(operator.isub)
    __imul__ = _inplace(operator.imul)
    __itruediv__ = _inplace(operator.itruediv)


# -------------------------------------------------------------------- construction helpers
def from_dict(values):
    """Nested dict of arrays -> NamedArray; ``None`` / empty dict -> ``None`` (reference :541-570)."""
    if values is None or len(values) == 0:
        return None
    return NamedArray(**{k: (from_dict(v) if isinstance(v, dict) else v) for k, v in values.items()})


def array_like(x, value=0):
    if isinstance(x, NamedArray):
        return NamedArray(**{k: (None if v is None else array_like(v, value)) for k, v in x.items()})
    if isinstance(x, np.ndarray):
        out = np.zeros_like(x)
    else:
        assert torch is not None and isinstance(x, torch.Tensor), (
            f'Currently, namedarray only supports torch.Tensor and numpy.array (input is {type(x)})')
        out = torch.zeros_like(x)
    if value != 0:
        out[:] = value
    return out


def _fill_missing(xs):
    """When some list entries are None and others are not, stand zeros in for the Nones (:573-581)."""
    present = [x is not None for x in xs]
    if all(present) or not any(present):
        return
    template = xs[present.index(True)]
    for i, x in enumerate(xs):
        if x is None:
            xs[i] = array_like(template)


def recursive_aggregate(xs: List, aggregate_fn: Callable):
    """Aggregate (stack / concatenate ...) a list of identically structured NamedArrays leaf-wise."""
    _fill_missing(xs)
    head = xs[0]
    if isinstance(head, NamedArray):
        out = {}
        for k in head.keys():
            try:
                out[k] = recursive_aggregate([x[k] for x in xs], aggregate_fn)
            except Exception as e:
                raise RuntimeError(f"`recursive_aggregate` fails at an entry named `{k}`.") from e
        return NamedArray(**out)
    if head is None:
        return None
    return aggregate_fn(xs)


def recursive_apply(x, fn: Callable):
    """Apply ``fn`` to every non-None leaf; structure (and Nones) preserved."""
    if isinstance(x, NamedArray):
        out = {}
        for k, v in x.items():
            try:
                out[k] = recursive_apply(v, fn)
            except Exception as e:
                raise RuntimeError(f"`recursive_apply` fails at an entry named `{k}`") from e
        return NamedArray(**out)
    if x is None:
        return None
    return fn(x)


def flatten(x: NamedArray) -> List[Tuple]:
    """[(dotted.key, leaf), ...] in sorted-field order (reference :663-672)."""
    out = []
    for k, v in x.items():
        if isinstance(v, NamedArray):
            out.extend((f"{k}.{kk}", vv) for kk, vv in flatten(v))
        else:
            out.append((k, v))
    return out


def from_flattened(entries):
    """Inverse of :func:`flatten` (reference :675-692)."""
    tree = {}
    for key, leaf in entries:
        node = tree
        *parents, last = key.split('.')
        for p in parents:
            node = node.setdefault(p, {})
        node[last] = leaf
    return from_dict(tree)


# -------------------------------------------------------------------- wire formats
_DTYPE_BYTES = {1: (np.bool_, np.int8, np.uint8), 2: (np.uint16, np.int16, np.float16),
                4: (np.uint32, np.int32, np.float32), 8: (np.uint64, np.int64, np.float64)}


def _encode_dtype(dtype) -> str:
    """Same spelling as the reference's base/numpy_utils.py:61-76 (bool travels as uint8)."""
    dtype = np.dtype(dtype)
    if dtype == np.uint8 or dtype == np.bool_:
        return "uint8"
    if dtype in (np.float32, np.float64, np.int32, np.int64):
        return dtype.name
    if str(dtype).startswith("<U"):
        return str(dtype)
    raise NotImplementedError(f"Data type to string not implemented: {dtype}.")


def size_bytes(x) -> int:
    return int(sum(v.dtype.itemsize * int(np.prod(v.shape)) for _, v in flatten(x) if v is not None))


def _to_quadruples(x):
    out = []
    for k, v in flatten(x):
        if v is None:
            out.append((k.encode('ascii'), b'', b'', b''))
        else:
            v = np.asarray(v)
            out.append((k.encode('ascii'), _encode_dtype(v.dtype).encode('ascii'),
                        str(tuple(v.shape)).encode('ascii'), v.tobytes()))
    return list(itertools.chain.from_iterable(out))


def _from_quadruples(chunks):
    entries = []
    for i in range(0, len(chunks) - 3, 4):
        key, dtype, shape, buf = chunks[i:i + 4]
        if dtype == b'':
            entries.append((key.decode('ascii'), None))
        else:
            arr = np.frombuffer(buf, dtype=np.dtype(dtype.decode('ascii')))
            entries.append((key.decode('ascii'), arr.reshape(*ast.literal_eval(shape.decode('ascii')))))
    return from_flattened(entries)


def is_raw_bytes(chunks: List[bytes]) -> bool:
    return len(chunks) > 0 and bytes(chunks[0]) == _TAG_RAW_BYTES


def iter_raw_leaves(chunks: List[bytes]):
    """(dotted key, zero-copy ndarray | None) for every leaf of a ``raw_bytes`` message, without building the tree
    (the ingest ring writes each leaf straight into its pinned block)."""
    body = chunks[1:-1]
    for i in range(0, len(body) - 3, 4):
        key, dtype, shape, buf = body[i:i + 4]
        if dtype == b'':
            yield key.decode('ascii'), None
        else:
            arr = np.frombuffer(buf, dtype=np.dtype(dtype.decode('ascii')))
            yield key.decode('ascii'), arr.reshape(*ast.literal_eval(shape.decode('ascii')))


def dumps(obj: NamedArray, method: str = "pickle_dict") -> List[bytes]:
    """Encode to the reference's list-of-bytes wire format: [tag, payload..., pickled metadata]."""
    if method == "pickle_dict":
        body = [_TAG_PICKLE_DICT, pickle.dumps((obj.__class__.__name__, obj.to_dict()))]
    elif method == "pickle":
        body = [_TAG_PICKLE, pickle.dumps(obj)]
    elif method == "raw_bytes":
        body = [_TAG_RAW_BYTES] + _to_quadruples(obj)
    elif method in _COMPRESSED_TAGS.values():
        raise NotImplementedError(f"encoding `{method}` needs the blosc package, which is not available")
    else:
        raise NotImplementedError(f"Unknown method {method}.")
    return body + [pickle.dumps(dict(**obj.metadata))]


def loads(b: List[bytes]) -> NamedArray:
    tag = b[0]
    if tag == _TAG_PICKLE_DICT:
        _, values = pickle.loads(b[1])
        obj = from_dict(values)
    elif tag == _TAG_PICKLE:
        obj = pickle.loads(b[1])
    elif tag == _TAG_RAW_BYTES:
        obj = _from_quadruples(b[1:-1])
    elif tag in _COMPRESSED_TAGS:
        raise NotImplementedError(f"encoding `{_COMPRESSED_TAGS[tag]}` needs the blosc package")
    else:
        raise NotImplementedError(f"Unknown NamedArrayEncodingMethod value {tag}.")
    obj.clear_metadata()
    obj.register_metadata(**pickle.loads(b[-1]))
    return obj
