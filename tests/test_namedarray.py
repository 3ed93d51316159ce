"""NamedArray semantics (reference base/namedarray.py; cases modelled on base/tests/namedarray_test.py) and the
wire formats against bytes produced by the reference itself."""
import copy
import pickle

import numpy as np
import pytest

from srl_amd import namedarray as na
from srl_amd.namedarray import NamedArray, recursive_aggregate, recursive_apply


class Point(NamedArray):

    def __init__(self, x, y):
        super().__init__(x=x, y=y)


def test_fields_sorted_and_dict_protocol():
    a = NamedArray(z=np.zeros(2), a=np.ones(2), m=None)
    assert list(a.keys()) == ["a", "m", "z"]
    assert "a" in a and "q" not in a and len(a) == 3
    assert a["a"] is a.a
    d = dict(**a)
    assert set(d) == {"a", "m", "z"}
    assert a.to_dict()["m"] is None


def test_slicing_keeps_class_and_none():
    p = Point(np.arange(6).reshape(3, 2), None)
    q = p[1:]
    assert isinstance(q, Point) and q.y is None and q.x.shape == (2, 2)
    assert p[0].x.shape == (2,)
    nested = NamedArray(a=np.arange(4), b=NamedArray(c=np.arange(4) * 2))
    assert np.array_equal(nested[2:].b.c, [4, 6])
    with pytest.raises(Exception):
        p[10]


def test_setitem_broadcast_and_structured():
    p = Point(np.array([1, 2]), np.array([3, 4]))
    p[0] = 0
    assert p.x[0] == 0 and p.y[0] == 0
    p[0] = Point(5, 6)
    assert p.x[0] == 5 and p.y[0] == 6
    with pytest.raises(ValueError):
        p[0] = NamedArray(x=1)
    p["x"] = np.array([9, 9])
    assert p.x[0] == 9


def test_arithmetic_respects_none():
    a = NamedArray(u=np.array([1., 2.]), v=None)
    b = a * 2
    assert np.array_equal(b.u, [2., 4.]) and b.v is None
    c = a + NamedArray(u=np.array([1., 1.]), v=None)
    assert np.array_equal(c.u, [2., 3.])
    a /= 2
    assert np.array_equal(a.u, [0.5, 1.0])


def test_recursive_helpers_and_flatten():
    xs = [NamedArray(a=np.full((2,), i), b=NamedArray(c=np.full((2, 3), i)), n=None) for i in range(3)]
    st = recursive_aggregate(xs, lambda v: np.stack(v, axis=1))
    assert st.a.shape == (2, 3) and st.b.c.shape == (2, 3, 3) and st.n is None
    xs2 = [NamedArray(a=np.ones(2), m=None), NamedArray(a=np.ones(2), m=np.ones(2))]
    agg = recursive_aggregate(xs2, np.stack)  # a missing leaf is zero-filled (reference :573-581)
    assert np.array_equal(agg.m, [[0, 0], [1, 1]])
    ap = recursive_apply(st, lambda v: v.shape)
    assert ap.b.c == (2, 3, 3)
    flat = na.flatten(st)
    assert [k for k, _ in flat] == ["a", "b.c", "n"]
    back = na.from_flattened(flat)
    assert np.array_equal(back.b.c, st.b.c) and back.n is None
    assert na.from_dict({}) is None and na.from_dict(None) is None
    assert na.size_bytes(st) == st.a.nbytes + st.b.c.nbytes


def test_metadata_pickle_deepcopy_length():
    p = Point(np.arange(4), np.arange(8).reshape(4, 2))
    p.register_metadata(tag="t")
    with pytest.raises(KeyError):
        p.register_metadata(x=1)
    q = pickle.loads(pickle.dumps(p))
    assert isinstance(q, Point) and q.metadata["tag"] == "t" and np.array_equal(q.y, p.y)
    r = copy.deepcopy(p)
    r.x[0] = 100
    assert p.x[0] == 0 and r.metadata["tag"] == "t"
    assert p.length(0) == 4 and p.length(1) == 2
    assert p.pop_metadata("tag") == "t" and "tag" not in p.metadata
    assert p.average_of("x", ignore_negative=False) == 1.5 and p.max_of("x") == 3 and p.min_of("x") == 0


def test_wire_formats_roundtrip_and_reference_bytes(golden):
    g = golden("host.npz")
    obj = NamedArray(a=g["wire_a"], b=NamedArray(c=g["wire_c"], d=None), e=g["wire_e"])
    obj.register_metadata(tag="golden")
    for method in ("pickle_dict", "pickle", "raw_bytes"):
        back = na.loads(na.dumps(obj, method=method))
        assert np.array_equal(back.a, obj.a) and np.array_equal(back.b.c, obj.b.c) and back.b.d is None
        assert back.e.dtype == np.int64 and back.metadata["tag"] == "golden"
    # raw_bytes: byte-identical to what the reference emits, and the reference's bytes decode here
    ref_chunks = [g[f"wire_raw_bytes_{i}"].tobytes() for i in range(int(g["wire_raw_bytes_n"]))]
    mine = na.dumps(obj, method="raw_bytes")
    assert mine[:-1] == ref_chunks[:-1]
    dec = na.loads(ref_chunks)
    assert np.array_equal(dec.a, obj.a) and np.array_equal(dec.b.c, obj.b.c) and dec.metadata["tag"] == "golden"
    ref_pd = [g[f"wire_pickle_dict_{i}"].tobytes() for i in range(int(g["wire_pickle_dict_n"]))]
    dec = na.loads(ref_pd)
    assert np.array_equal(dec.e, obj.e) and dec.b.d is None
    with pytest.raises(NotImplementedError):
        na.dumps(obj, method="obs_compress")
