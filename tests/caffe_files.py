"""Hand-written encoders of the two files DeepLCD::DeepLCD reads (reference src/deeplcd.cpp:24-25): a deploy.prototxt (protobuf text
format) and a .caffemodel (protobuf wire format, caffe.proto: NetParameter.layer = 100 -> LayerParameter{name = 1, type = 2,
blobs = 7 -> BlobProto{shape = 7 -> BlobShape{dim = 1}, data = 5 packed float}}).  No Caffe, no protobuf library: test fixtures."""
import struct

import numpy as np


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):                       # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _blob(arr, legacy=False):
    a = np.ascontiguousarray(arr, np.float32)
    if legacy:                                 # old files: num / channels / height / width instead of shape
        dims = list(a.shape) + [1] * (4 - a.ndim) if a.ndim > 1 else [1, 1, 1, a.size]
        head = b"".join(_varint((i + 1) << 3) + _varint(d) for i, d in enumerate(dims))
    else:
        head = _ld(7, _ld(1, b"".join(_varint(d) for d in a.shape)))
    return head + _ld(5, a.tobytes())


def caffemodel(named_blobs, v1=False, legacy_shape=False):
    """named_blobs: [(layer name, type string, [weight array, bias array])] -> bytes"""
    out = _ld(1, b"calc")                      # NetParameter.name
    v1_types = {"Convolution": 4, "ReLU": 18, "Pooling": 17, "LRN": 15}
    for name, typ, blobs in named_blobs:
        if v1:                                 # V1LayerParameter: name = 4, type = 5 (enum), blobs = 6
            body = _ld(4, name.encode()) + _varint(5 << 3) + _varint(v1_types[typ]) + b"".join(_ld(6, _blob(b, legacy_shape)) for b in blobs)
            out += _ld(2, body)
        else:
            body = _ld(1, name.encode()) + _ld(2, typ.encode()) + b"".join(_ld(7, _blob(b, legacy_shape)) for b in blobs)
            out += _ld(100, body)
    return out


def prototxt(layers, input_style="input_dim"):
    """layers: CALC_LAYER_DTYPE-like records (type, num_output, kernel, stride, pad, local_size, alpha, beta, k) -> text"""
    t = ['name: "calc"   # hand-written fixture']
    if input_style == "input_dim":
        t += ['input: "data"', "input_dim: 1", "input_dim: 1", "input_dim: 120", "input_dim: 160"]
    elif input_style == "input_shape":
        t += ['input: "data"', "input_shape { dim: 1 dim: 1 dim: 120 dim: 160 }"]
    else:
        t += ['layer { name: "data" type: "Input" top: "data" input_param { shape: { dim: 1 dim: 1 dim: 120 dim: 160 } } }']
    names, nconv, top = [], 0, "data"
    for i, l in enumerate(layers):
        typ = int(l["type"])
        if typ == 1:
            nconv += 1
            nm = f"conv{nconv}"
            names.append(nm)
            t.append(f'layer {{ name: "{nm}" type: "Convolution" bottom: "{top}" top: "{nm}"\n  convolution_param {{ num_output: {int(l["num_output"])} '
                     f'kernel_size: {int(l["kernel"])} stride: {int(l["stride"])} pad: {int(l["pad"])} }} }}')
            top = nm
        elif typ == 2:
            t.append(f'layer {{ name: "relu{i}" type: "ReLU" bottom: "{top}" top: "{top}" }}')
        elif typ == 3:
            nm = f"pool{i}"
            t.append(f'layer {{ name: "{nm}" type: "Pooling" bottom: "{top}" top: "{nm}" pooling_param {{ pool: MAX kernel_size: {int(l["kernel"])} '
                     f'stride: {int(l["stride"])} }} }}')
            top = nm
        elif typ == 4:
            nm = f"norm{i}"
            t.append(f'layer {{ name: "{nm}" type: "LRN" bottom: "{top}" top: "{nm}" lrn_param {{ local_size: {int(l["local_size"])} '
                     f'alpha: {float(l["alpha"])!r} beta: {float(l["beta"])!r} k: {float(l["k"])!r} norm_region: ACROSS_CHANNELS }} }}')
            top = nm
    return "\n".join(t) + "\n", names


def split_weights(layers, flat):
    """flat blob -> [(w[OC][IC][K][K], b[OC])] per convolution"""
    out, ic, p = [], 1, 0
    for l in layers:
        if int(l["type"]) != 1:
            continue
        oc, k = int(l["num_output"]), int(l["kernel"])
        n = oc * ic * k * k
        out.append((np.asarray(flat[p:p + n], np.float32).reshape(oc, ic, k, k), np.asarray(flat[p + n:p + n + oc], np.float32)))
        p += n + oc
        ic = oc
    assert p == len(flat)
    return out


def write_pair(tmpdir, layers, flat, **kw):
    txt, names = prototxt(layers, kw.pop("input_style", "input_dim"))
    blobs = [(nm, "Convolution", [w, b]) for nm, (w, b) in zip(names, split_weights(layers, flat))]
    blobs.insert(1, ("relu_no_blobs", "ReLU", []))
    pp, mp = str(tmpdir / "deploy.prototxt"), str(tmpdir / "calc.caffemodel")
    open(pp, "w").write(txt)
    open(mp, "wb").write(caffemodel(blobs, **kw))
    return pp, mp
