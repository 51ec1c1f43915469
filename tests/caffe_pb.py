"""The subset of BVLC caffe.proto that DeepLCD's two files use, built at run time with google.protobuf (descriptor_pb2 + message_factory:
no protoc, no Caffe) — an encoder that is NOT this repository's own: `tests/caffe_files.py` writes the wire and text formats by hand, and a
reader tested only against its author's writer proves little.  Field numbers as in caffe.proto (recalled; proto2):
  NetParameter      name 1, layers 2 (V1LayerParameter), input 3, input_dim 4, input_shape 8 (BlobShape), layer 100 (LayerParameter)
  BlobShape         dim 1 (int64, packed)
  BlobProto         num 1, channels 2, height 3, width 4, data 5 (float, packed), diff 6, shape 7
  LayerParameter    name 1, type 2, bottom 3, top 4, blobs 7, convolution_param 106, lrn_param 118, pooling_param 121, relu_param 123,
                    input_param 143
  V1LayerParameter  bottom 2, top 3, name 4, type 5 (enum: CONVOLUTION 4, LRN 15, POOLING 17, RELU 18), blobs 6, convolution_param 10,
                    lrn_param 18, pooling_param 19, relu_param 30
  ConvolutionParameter num_output 1, bias_term 2, pad 3, kernel_size 4, group 5, stride 6 (pad / kernel_size / stride repeated uint32)
  PoolingParameter  pool 1 (MAX 0, AVE 1, STOCHASTIC 2), kernel_size 2, stride 3, pad 4
  LRNParameter      local_size 1, alpha 2, beta 3, norm_region 4 (ACROSS_CHANNELS 0, WITHIN_CHANNEL 1), k 5
  ReLUParameter     negative_slope 1
  InputParameter    shape 1 (BlobShape)"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto
_T = {"string": F.TYPE_STRING, "int32": F.TYPE_INT32, "int64": F.TYPE_INT64, "uint32": F.TYPE_UINT32, "float": F.TYPE_FLOAT, "bool": F.TYPE_BOOL}


def _build(packed_data=True, suffix=""):
    fd = descriptor_pb2.FileDescriptorProto(name=f"caffe_subset{suffix}.proto", package=f"caffe{suffix}", syntax="proto2")
    pk = f".caffe{suffix}."

    def msg(name, fields, enums=()):
        m = fd.message_type.add(name=name)
        for en, values in enums:
            e = m.enum_type.add(name=en)
            for vn, vv in values:
                e.value.add(name=vn, number=vv)
        for fname, num, typ, rep, *opt in fields:
            f = m.field.add(name=fname, number=num, label=F.LABEL_REPEATED if rep else F.LABEL_OPTIONAL)
            if typ in _T:
                f.type = _T[typ]
            elif typ.startswith("enum:"):
                f.type = F.TYPE_ENUM; f.type_name = pk + name + "." + typ[5:]
            else:
                f.type = F.TYPE_MESSAGE; f.type_name = pk + typ
            if opt and opt[0] == "packed":
                f.options.packed = True
        return m
    msg("BlobShape", [("dim", 1, "int64", True, "packed")])
    msg("BlobProto", [("num", 1, "int32", False), ("channels", 2, "int32", False), ("height", 3, "int32", False), ("width", 4, "int32", False),
                      ("data", 5, "float", True) + (("packed",) if packed_data else ()), ("diff", 6, "float", True) + (("packed",) if packed_data else ()),
                      ("shape", 7, "BlobShape", False)])
    msg("ConvolutionParameter", [("num_output", 1, "uint32", False), ("bias_term", 2, "bool", False), ("pad", 3, "uint32", True), ("kernel_size", 4, "uint32", True),
                                 ("group", 5, "uint32", False), ("stride", 6, "uint32", True)])
    msg("PoolingParameter", [("pool", 1, "enum:PoolMethod", False), ("kernel_size", 2, "uint32", False), ("stride", 3, "uint32", False), ("pad", 4, "uint32", False)],
        enums=[("PoolMethod", [("MAX", 0), ("AVE", 1), ("STOCHASTIC", 2)])])
    msg("LRNParameter", [("local_size", 1, "uint32", False), ("alpha", 2, "float", False), ("beta", 3, "float", False), ("norm_region", 4, "enum:NormRegion", False),
                         ("k", 5, "float", False)], enums=[("NormRegion", [("ACROSS_CHANNELS", 0), ("WITHIN_CHANNEL", 1)])])
    msg("ReLUParameter", [("negative_slope", 1, "float", False)])
    msg("InputParameter", [("shape", 1, "BlobShape", True)])
    msg("LayerParameter", [("name", 1, "string", False), ("type", 2, "string", False), ("bottom", 3, "string", True), ("top", 4, "string", True),
                           ("blobs", 7, "BlobProto", True), ("convolution_param", 106, "ConvolutionParameter", False), ("lrn_param", 118, "LRNParameter", False),
                           ("pooling_param", 121, "PoolingParameter", False), ("relu_param", 123, "ReLUParameter", False), ("input_param", 143, "InputParameter", False)])
    msg("V1LayerParameter", [("bottom", 2, "string", True), ("top", 3, "string", True), ("name", 4, "string", False), ("type", 5, "enum:LayerType", False),
                             ("blobs", 6, "BlobProto", True), ("convolution_param", 10, "ConvolutionParameter", False), ("lrn_param", 18, "LRNParameter", False),
                             ("pooling_param", 19, "PoolingParameter", False), ("relu_param", 30, "ReLUParameter", False)],
        enums=[("LayerType", [("NONE", 0), ("CONVOLUTION", 4), ("LRN", 15), ("POOLING", 17), ("RELU", 18)])])
    msg("NetParameter", [("name", 1, "string", False), ("layers", 2, "V1LayerParameter", True), ("input", 3, "string", True), ("input_dim", 4, "int32", True),
                         ("input_shape", 8, "BlobShape", True), ("layer", 100, "LayerParameter", True)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName(f"caffe{suffix}.{n}"))
            for n in ("NetParameter", "LayerParameter", "V1LayerParameter", "BlobProto", "BlobShape")}


_CACHE = {}


def classes(packed_data=True):
    key = bool(packed_data)
    if key not in _CACHE:
        _CACHE[key] = _build(packed_data, "" if packed_data else "_unpacked")
    return _CACHE[key]


def build_net(layers, conv_blobs=None, v1=False, legacy_shape=False, input_style="input_dim", packed_data=True):
    """layers: CALC_LAYER_DTYPE records; conv_blobs: [(w[OC][IC][K][K], b[OC])] per convolution or None (a deploy net without weights)"""
    C = classes(packed_data)
    net = C["NetParameter"](name="calc")
    if input_style == "input_dim":
        net.input.append("data"); net.input_dim.extend([1, 1, 120, 160])
    elif input_style == "input_shape":
        net.input.append("data"); net.input_shape.add().dim.extend([1, 1, 120, 160])
    elif not v1:
        il = net.layer.add(name="data", type="Input"); il.top.append("data"); il.input_param.shape.add().dim.extend([1, 1, 120, 160])
    else:
        net.input.append("data"); net.input_dim.extend([1, 1, 120, 160])
    top, nconv = "data", 0
    v1_type = {1: 4, 2: 18, 3: 17, 4: 15}; v2_type = {1: "Convolution", 2: "ReLU", 3: "Pooling", 4: "LRN"}
    for i, l in enumerate(layers):
        typ = int(l["type"])
        lay = net.layers.add() if v1 else net.layer.add()
        if v1:
            lay.type = v1_type[typ]
        else:
            lay.type = v2_type[typ]
        lay.bottom.append(top)
        if typ == 1:
            nconv += 1
            lay.name = f"conv{nconv}"; top = lay.name
            cp = lay.convolution_param
            cp.num_output = int(l["num_output"]); cp.kernel_size.append(int(l["kernel"])); cp.stride.append(int(l["stride"])); cp.pad.append(int(l["pad"]))
            if conv_blobs is not None:
                for arr in conv_blobs[nconv - 1]:
                    b = lay.blobs.add()
                    if legacy_shape:
                        dims = list(arr.shape) + [1] * (4 - arr.ndim) if arr.ndim > 1 else [1, 1, 1, arr.size]
                        b.num, b.channels, b.height, b.width = [int(d) for d in dims]
                    else:
                        b.shape.dim.extend(int(d) for d in arr.shape)
                    b.data.extend(float(v) for v in arr.ravel())
        elif typ == 2:
            lay.name = f"relu{i}"
        elif typ == 3:
            lay.name = f"pool{i}"; top = lay.name
            lay.pooling_param.pool = 0; lay.pooling_param.kernel_size = int(l["kernel"]); lay.pooling_param.stride = int(l["stride"])
        else:
            lay.name = f"norm{i}"; top = lay.name
            lp = lay.lrn_param
            lp.local_size = int(l["local_size"]); lp.alpha = float(l["alpha"]); lp.beta = float(l["beta"]); lp.k = float(l["k"]); lp.norm_region = 0
        lay.top.append(top)
    return net
