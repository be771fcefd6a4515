"""ps_amd/ps_server.py builds the messages of ps.proto without protoc: pin them to the wire format.

Expected bytes are hand-encoded from the .proto (src/main/resources/proto/ps.proto: field numbers / types), so a
reference worker's protobuf-java stubs and this server agree byte for byte.  No GPU, no library call."""
import struct

import numpy as np
import pytest

S = pytest.importorskip("ps_amd.ps_server")


def f32(*xs):
    return b"".join(struct.pack("<f", x) for x in xs)


def test_matrix_wire_bytes():
    m = S.Matrix(key="a", row=2, cols=1, data=[1.0, 2.0], update=True)
    # 1: string key | 2: int32 row | 3: int32 cols | 4: packed repeated float | 5: bool update   (ps.proto:16-23)
    want = b"\x0a\x01a" + b"\x10\x02" + b"\x18\x01" + b"\x22\x08" + f32(1.0, 2.0) + b"\x28\x01"
    assert m.SerializeToString() == want
    back = S.Matrix.FromString(want)
    assert (back.key, back.row, back.cols, list(back.data), back.update) == ("a", 2, 1, [1.0, 2.0], True)


def test_field_numbers_follow_the_proto():
    # GetMessage.resp is field 4 (not 3), GradientMessage: gradient 2, isAsync 3, updaterKey 4, resp 5   (ps.proto:40-66)
    assert S.GetMessage(resp=S.Resp(ec=204, em="null weights")).SerializeToString() == b"\x22\x11" + b"\x08\xcc\x01" + b"\x12\x0cnull weights"
    g = S.GradientMessage(gradient=S.Matrix(key="k"), isAsync=True, updaterKey="adam", resp=S.Resp(ec=500))
    assert g.SerializeToString() == b"\x12\x03\x0a\x01k" + b"\x18\x01" + b"\x22\x04adam" + b"\x2a\x03\x08\xf4\x03"
    u = S.UpdateListMessage(meta=S.RequestMeta(host="h"), weights=[S.Matrix(key="x"), S.Matrix(key="y")], replace=True)
    assert u.SerializeToString() == b"\x0a\x03\x0a\x01h" + b"\x12\x03\x0a\x01x" + b"\x12\x03\x0a\x01y" + b"\x20\x01"
    assert S.BarrierMessage(resp=S.Resp(ec=200, em="")).SerializeToString() == b"\x12\x03\x08\xc8\x01"


def test_service_surface():
    assert S.SERVICE == "net.PS"                                     # package net; service PS   (ps.proto:2,7)
    assert set(S.RPCS) == {"get", "getList", "upsert", "upsertList", "push", "barrier"}
    for name, (rq, rs) in S.RPCS.items():
        assert rq is rs                                              # every rpc returns its request type (ps.proto:8-13)
    assert S.updater_group("emF13.28305.0") == "emF" and S.updater_group("wide.bias") == "wide"
    assert S.updater_group("fc1.weights") == "fc1.weights"
