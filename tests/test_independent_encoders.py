"""The host-format readers against encoders that are NOT this repository's own (round-3 review: "a reader tested only against files written
by the repo's own encoder"):
  * DeepLCD's files (reference src/deeplcd.cpp:10-31, get_model.sh): a caffe.proto NetParameter built with google.protobuf and serialised
    by ITS binary encoder (.caffemodel: V2 `layer` and V1 `layers`, packed and unpacked BlobProto.data, `shape` and legacy num / channels /
    height / width) and ITS text_format printer (deploy.prototxt: multi-line, one-line, short repeated fields) -> myslam_calc_parse_caffe;
  * cv::imread(..., IMREAD_GRAYSCALE) (reference app/run_kitti_stereo.cpp:66-67): PNGs written by PIL's encoder (libpng-independent zlib
    settings, optimize on / off, every colour type PIL can write) -> myslam_io_read_png_gray.
Host code only: no GPU."""
import numpy as np
import pytest

import caffe_files
import caffe_pb

text_format = pytest.importorskip("google.protobuf.text_format")


def _pair(tmp_path, layers, w, prototxt_kw=None, **kw):
    blobs = caffe_files.split_weights(layers, w)
    model = caffe_pb.build_net(layers, blobs, **kw)
    deploy = caffe_pb.build_net(layers, None, v1=kw.get("v1", False), input_style=kw.get("input_style", "input_dim"))
    pp, mp = str(tmp_path / "deploy.prototxt"), str(tmp_path / "calc.caffemodel")
    open(pp, "w").write(text_format.MessageToString(deploy, **(prototxt_kw or {})))
    open(mp, "wb").write(model.SerializeToString())
    return pp, mp


@pytest.mark.parametrize("kw", [{}, {"legacy_shape": True}, {"v1": True, "legacy_shape": True}, {"v1": True}, {"packed_data": False},
                                {"input_style": "input_shape"}, {"input_style": "layer"}])
def test_protobuf_written_model_parses_bit_for_bit(pkg, synth, tmp_path, kw):
    api = pkg.api
    L = api.calc_default_layers(); w = synth.calc_weights()
    pp, mp = _pair(tmp_path, L, w, **kw)
    L2, w2 = api.calc_parse_caffe(pp, mp)
    assert L2.tobytes() == L.tobytes()
    assert np.array_equal(w2.view(np.uint32), np.asarray(w, np.float32).ravel().view(np.uint32))
    if kw.get("packed_data") is False:              # the two encodings really differ on the wire: one tag per float against one packed run
        packed = caffe_pb.build_net(L, caffe_files.split_weights(L, w)).SerializeToString()
        assert len(open(mp, "rb").read()) > len(packed) + 100000


@pytest.mark.parametrize("fmt", [{}, {"as_one_line": True}, {"use_short_repeated_primitives": True}, {"indent": 4}, {"use_index_order": True}])
def test_protobuf_printed_prototxt_styles(pkg, synth, tmp_path, fmt):
    """text_format's own layouts of the same deploy net: nested blocks on their own lines, everything on one line, `dim: [1, 1, 120, 160]`
    lists, deeper indentation, fields in declaration order"""
    api = pkg.api
    L = api.calc_default_layers()
    L["alpha"][3] = 3e-3; L["beta"][7] = 0.5; L["k"][7] = 2.0          # non-default LRN constants must survive the float printer
    w = synth.calc_weights()
    pp, mp = _pair(tmp_path, L, w, prototxt_kw=fmt, input_style="input_shape")
    L2, w2 = api.calc_parse_caffe(pp, mp)
    assert L2.tobytes() == L.tobytes() and np.array_equal(w2, np.asarray(w, np.float32).ravel())


def test_protobuf_model_and_handwritten_model_are_the_same_net(pkg, synth, tmp_path):
    """the two encoders agree on what the bytes mean: a prototxt from one with a caffemodel from the other"""
    api = pkg.api
    L = api.calc_default_layers(); w = synth.calc_weights(seed=7)
    d1 = tmp_path / "a"; d2 = tmp_path / "b"; d1.mkdir(); d2.mkdir()
    pp1, mp1 = caffe_files.write_pair(d1, L, w)
    pp2, mp2 = _pair(d2, L, w)
    for pp, mp in ((pp1, mp2), (pp2, mp1)):
        L2, w2 = api.calc_parse_caffe(pp, mp)
        assert L2.tobytes() == L.tobytes() and np.array_equal(w2, np.asarray(w, np.float32).ravel())


# ---- PNG -------------------------------------------------------------------------------------------------------------------------
Image = pytest.importorskip("PIL.Image")


def _img(arr, mode):
    a = np.ascontiguousarray(arr)
    return Image.frombytes(mode, (a.shape[1], a.shape[0]), a.tobytes())


def _grey_of_rgb8(rgb):
    """what cv::imread(IMREAD_GRAYSCALE) gets from libpng (png_set_rgb_to_gray(1, 0.299, 0.587)): truncated 15-bit coefficients, no rounding"""
    r, g, b = [rgb[..., i].astype(np.int64) for i in range(3)]
    out = ((9797 * r + 19234 * g + 3737 * b) >> 15).astype(np.uint8)
    same = (r == g) & (r == b)
    out[same] = rgb[..., 0][same]
    return out


def test_pil_written_pngs(pkg, synth, tmp_path):
    api = pkg.api
    kitti = synth.stereo_pair(0, 0)[0]                              # 1241 x 376, the KITTI shape
    assert kitti.shape == (376, 1241)
    for name, kw in (("default", {}), ("optimize", {"optimize": True}), ("level0", {"compress_level": 0}), ("level1", {"compress_level": 1}),
                     ("level9", {"compress_level": 9})):
        p = str(tmp_path / f"grey_{name}.png")
        _img(kitti, "L").save(p, **kw)
        assert np.array_equal(api.read_png_gray(p), kitti), name
    rng = np.random.default_rng(3)
    # 16-bit grey: the high byte (png_set_strip_16)
    g16 = (rng.integers(0, 65536, (61, 83))).astype(np.uint16)
    p = str(tmp_path / "g16.png"); _img(g16.astype("<u2"), "I;16").save(p)
    assert np.array_equal(api.read_png_gray(p), (g16 >> 8).astype(np.uint8))
    # RGB / RGBA: libpng's grey; alpha dropped
    rgb = rng.integers(0, 256, (45, 70, 3), dtype=np.uint8); rgb[:5] = rgb[:5, :, :1]          # some R = G = B pixels
    p = str(tmp_path / "rgb.png"); _img(rgb, "RGB").save(p, optimize=True)
    assert np.array_equal(api.read_png_gray(p), _grey_of_rgb8(rgb))
    rgba = np.concatenate([rgb, rng.integers(0, 256, (45, 70, 1), dtype=np.uint8)], axis=2)
    p = str(tmp_path / "rgba.png"); _img(rgba, "RGBA").save(p)
    assert np.array_equal(api.read_png_gray(p), _grey_of_rgb8(rgb))
    # grey + alpha (colour type 4)
    la = np.stack([kitti[:50, :90], rng.integers(0, 256, (50, 90), dtype=np.uint8)], axis=2)
    p = str(tmp_path / "la.png"); _img(la, "LA").save(p)
    assert np.array_equal(api.read_png_gray(p), la[..., 0])
    # palette (colour type 3), 8-bit and packed 4 / 2 / 1-bit indices
    for ncol in (200, 16, 4, 2):
        pal = rng.integers(0, 256, (ncol, 3), dtype=np.uint8)
        idx = rng.integers(0, ncol, (33, 57), dtype=np.uint8)
        im = _img(idx, "P"); im.putpalette(pal.ravel().tolist())
        p = str(tmp_path / f"pal{ncol}.png"); im.save(p, bits={200: 8, 16: 4, 4: 2, 2: 1}[ncol])
        assert np.array_equal(api.read_png_gray(p), _grey_of_rgb8(pal[idx])), ncol
    # 1-bit grey (PIL mode "1"): 0 / 255
    bw = rng.integers(0, 2, (20, 37)).astype(bool)
    p = str(tmp_path / "bw.png"); Image.fromarray(bw).save(p)
    assert np.array_equal(api.read_png_gray(p), bw.astype(np.uint8) * 255)
    # what PIL itself calls grey (ITU-R 601-2 luma, rounded) differs from libpng's truncating conversion by at most one level
    assert np.abs(np.asarray(_img(rgb, "RGB").convert("L")).astype(int) - _grey_of_rgb8(rgb).astype(int)).max() <= 1


def test_interlaced_png_is_refused(pkg, tmp_path):
    """Adam7 files are the reader's documented refusal (KITTI's are not interlaced): a status, never a wrong image.  PIL cannot write them;
    the IHDR flag is set by hand on a PIL-written file and the chunk's CRC redone."""
    import struct
    import zlib
    api = pkg.api
    p = str(tmp_path / "x.png")
    _img(np.arange(64, dtype=np.uint8).reshape(8, 8), "L").save(p)
    b = bytearray(open(p, "rb").read())
    assert b[12:16] == b"IHDR"
    b[28] = 1                                                        # interlace method
    b[29:33] = struct.pack(">I", zlib.crc32(bytes(b[12:29])) & 0xffffffff)
    open(p, "wb").write(bytes(b))
    with pytest.raises(api.MyslamError):
        api.read_png_gray(p)
