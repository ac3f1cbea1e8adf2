"""A minimal protobuf-wire writer for ONNX ModelProto (test helper): a SECOND producer for the importer's parser next to
PyTorch's exporter (tests/golden/make_onnx_fixtures.py).  Encodes with float_data / int64_data and unpacked repeated
fields where PyTorch uses raw_data and packed ones, so both encodings of onnx.proto3 are exercised."""
import struct


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wt):
    return _varint((field << 3) | wt)


def _len(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _str(field, s):
    return _len(field, s.encode())


def _int(field, v):
    return _key(field, 0) + _varint(v)


def tensor(name, dims, values, raw=False, int64=False):
    out = b"".join(_int(1, d) for d in dims)  # unpacked dims
    out += _int(2, 7 if int64 else 1)
    if int64:
        out += _len(9, struct.pack("<%dq" % len(values), *values)) if raw else b"".join(_int(7, v) for v in values)
    elif raw:
        out += _len(9, struct.pack("<%df" % len(values), *values))
    else:
        out += _len(4, struct.pack("<%df" % len(values), *values))  # packed float_data
    return out + _str(8, name)


def attr_ints(name, vals):
    return _str(1, name) + b"".join(_int(8, v) for v in vals) + _int(20, 7)


def attr_int(name, v):
    return _str(1, name) + _int(3, v) + _int(20, 2)


def attr_float(name, v):
    return _str(1, name) + _key(2, 5) + struct.pack("<f", v) + _int(20, 1)


def attr_str(name, s):
    return _str(1, name) + _len(4, s.encode()) + _int(20, 3)


def node(op, inputs, outputs, attrs=(), name=""):
    out = b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outputs)
    if name:
        out += _str(3, name)
    out += _str(4, op)
    return out + b"".join(_len(5, a) for a in attrs)


def value_info(name, dims):
    shape = b""
    for d in dims:
        shape += _len(1, _str(2, d) if isinstance(d, str) else _int(1, d))
    ttype = _int(1, 1) + _len(2, shape)
    return _str(1, name) + _len(2, _len(1, ttype))


def model(nodes, initializers, inputs, outputs, opset=11, graph_name="g"):
    g = b"".join(_len(1, n) for n in nodes) + _str(2, graph_name)
    g += b"".join(_len(5, t) for t in initializers)
    g += b"".join(_len(11, v) for v in inputs) + b"".join(_len(12, v) for v in outputs)
    return _int(1, 7) + _str(2, "tests/onnx_writer.py") + _len(7, g) + _len(8, _str(1, "") + _int(2, opset))
