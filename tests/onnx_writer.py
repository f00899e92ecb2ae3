"""Hand-encoded ONNX ModelProto files for the loader tests (neither `onnx` nor ONNX Runtime exist offline).
Only the protobuf wire format is used: ModelProto{ir_version=1, producer_name=2, graph=7},
GraphProto{node=1, name=2, initializer=5, input=11, output=12}, TensorProto{dims=1, data_type=2, float_data=4,
int64_data=7, name=8, raw_data=9}, NodeProto{input=1, output=2, name=3, op_type=4, attribute=5},
AttributeProto{name=1, i=3, s=4, t=5, ints=8, type=20}."""
import struct

import numpy as np


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _ld(field, payload: bytes):
    return _key(field, 2) + _varint(len(payload)) + payload


def tensor(name, arr, style="raw"):
    """style: raw (raw_data), packed (float_data / int64_data packed), unpacked (one element per tag)."""
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7, np.dtype(np.float16): 10, np.dtype(np.float64): 11}[arr.dtype]
    body = b"".join(_key(1, 0) + _varint(int(d)) for d in arr.shape)  # dims, unpacked
    body += _key(2, 0) + _varint(dt)
    body += _ld(8, name.encode())
    if style == "raw":
        body += _ld(9, arr.tobytes())
    elif dt == 1 and style == "packed":
        body += _ld(4, arr.astype("<f4").tobytes())
    elif dt == 1 and style == "unpacked":
        body += b"".join(_key(4, 5) + struct.pack("<f", float(x)) for x in arr.ravel())
    elif dt == 7:
        body += _ld(7, b"".join(_varint(int(x)) for x in arr.ravel()))
    else:
        raise ValueError(style)
    return body


def attr_int(name, v):
    return _ld(1, name.encode()) + _key(3, 0) + _varint(int(v)) + _key(20, 0) + _varint(2)  # type INT


def attr_ints(name, vs, packed=True):
    body = _ld(1, name.encode())
    if packed:
        body += _ld(8, b"".join(_varint(int(v)) for v in vs))
    else:
        body += b"".join(_key(8, 0) + _varint(int(v)) for v in vs)
    return body + _key(20, 0) + _varint(7)  # type INTS


def attr_str(name, v):
    return _ld(1, name.encode()) + _ld(4, v.encode()) + _key(20, 0) + _varint(3)  # type STRING


def attr_tensor(name, tensor_body):
    return _ld(1, name.encode()) + _ld(5, tensor_body) + _key(20, 0) + _varint(4)  # type TENSOR


def node(op_type, inputs, outputs, name="", attrs=()):
    body = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    return body + _ld(3, name.encode()) + _ld(4, op_type.encode()) + b"".join(_ld(5, a) for a in attrs)


def value_info(name):
    return _ld(1, name.encode())


def model(initializers, nodes=(), inputs=(), outputs=(), producer="stn-tests"):
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in initializers)
    g += b"".join(_ld(11, value_info(i)) for i in inputs) + b"".join(_ld(12, value_info(o)) for o in outputs)
    return _key(1, 0) + _varint(8) + _ld(2, producer.encode()) + _ld(7, g)
