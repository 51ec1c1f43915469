"""The CALC model as data: the dependency-free deploy.prototxt / .caffemodel reader behind myslam_lcd_create_from_caffe (reference
src/deeplcd.cpp:10-31) — host-only checks through myslam_calc_parse_caffe (no GPU needed)."""
import numpy as np
import pytest

import caffe_files


def test_default_layers_are_survey_a6(pkg):
    L = pkg.api.calc_default_layers()
    assert [int(t) for t in L["type"]] == [1, 2, 3, 4, 1, 2, 3, 4, 1, 2]
    assert [(int(l["num_output"]), int(l["kernel"]), int(l["stride"]), int(l["pad"])) for l in L if l["type"] == 1] == [(64, 5, 2, 4), (128, 4, 1, 2), (4, 3, 1, 0)]
    assert all(int(l["local_size"]) == 5 and abs(l["alpha"] - 1e-4) < 1e-12 and l["beta"] == 0.75 and l["k"] == 1.0 for l in L if l["type"] == 4)


@pytest.mark.parametrize("kw", [{}, {"legacy_shape": True}, {"v1": True, "legacy_shape": True}, {"input_style": "input_shape"}, {"input_style": "layer"}])
def test_parse_caffe_pair_round_trips(pkg, synth, tmp_path, kw):
    api = pkg.api
    L = api.calc_default_layers()
    w = synth.calc_weights()
    pp, mp = caffe_files.write_pair(tmp_path, L, w, **kw)
    L2, w2 = api.calc_parse_caffe(pp, mp)
    assert L2.tobytes() == L.tobytes()
    assert np.array_equal(w2.view(np.uint32), np.asarray(w, np.float32).ravel().view(np.uint32))      # bit for bit


def test_parse_changed_hyperparameters(pkg, synth, tmp_path):
    api = pkg.api
    L = api.calc_default_layers()
    L["alpha"][3] = 3e-3; L["beta"][7] = 0.5; L["k"][7] = 2.0
    pp, mp = caffe_files.write_pair(tmp_path, L, synth.calc_weights())
    L2, _ = api.calc_parse_caffe(pp, mp)
    assert L2.tobytes() == L.tobytes()


def test_parse_rejects_what_it_cannot_run(pkg, synth, tmp_path):
    api = pkg.api
    L = api.calc_default_layers()
    w = synth.calc_weights()
    pp, mp = caffe_files.write_pair(tmp_path, L, w)
    txt = open(pp).read()
    # an unknown layer type -> UNSUPPORTED
    open(pp, "w").write(txt + 'layer { name: "fc" type: "InnerProduct" bottom: "conv3" top: "fc" inner_product_param { num_output: 10 } }\n')
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_UNSUPPORTED
    # average pooling -> UNSUPPORTED
    open(pp, "w").write(txt.replace("pool: MAX", "pool: AVE", 1))
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_UNSUPPORTED
    # a different input size -> UNSUPPORTED (the reference always feeds 160 x 120)
    open(pp, "w").write(txt.replace("input_dim: 120", "input_dim: 128"))
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_UNSUPPORTED
    # weights that do not fit the prototxt -> INVALID
    open(pp, "w").write(txt.replace("num_output: 64", "num_output: 32"))
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_INVALID
    # a truncated caffemodel -> INVALID
    open(pp, "w").write(txt)
    open(mp, "wb").write(open(mp, "rb").read()[:-7])
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_INVALID
    # an output that is not 1064 long -> UNSUPPORTED
    L3 = L.copy(); L3["pad"][8] = 1
    pp, mp = caffe_files.write_pair(tmp_path, L3, w)
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_UNSUPPORTED


def test_parse_refuses_non_chains_and_deep_nesting(pkg, synth, tmp_path):
    """The reader runs the net as a chain: a layer whose `bottom` is not the previous layer's `top` (a skip connection, a second
    branch) is refused instead of being flattened silently; and a prototxt cannot exhaust the parser's stack with nested braces."""
    api = pkg.api
    L = api.calc_default_layers()
    w = synth.calc_weights()
    pp, mp = caffe_files.write_pair(tmp_path, L, w)
    text = open(pp).read()
    # re-wire the second convolution to the network input: a branch, not a chain
    import re
    hits = [m for m in re.finditer(r'type: "Convolution" bottom: "([^"]+)"', text)]
    assert len(hits) == 3
    broken = text[:hits[1].start(1)] + "data" + text[hits[1].end(1):]
    open(pp, "w").write(broken)
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_UNSUPPORTED
    # two bottoms (an Eltwise-style join) -> UNSUPPORTED
    open(pp, "w").write(text.replace('type: "LRN" bottom:', 'type: "LRN" bottom: "data" bottom:', 1))
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_UNSUPPORTED
    # 100 000 nested blocks: INVALID, not a stack overflow
    open(pp, "w").write(text + "\n" + "x { " * 100000 + "}" * 100000)
    with pytest.raises(api.MyslamError) as e:
        api.calc_parse_caffe(pp, mp)
    assert e.value.code == api.ERR_INVALID
    open(pp, "w").write(text)                      # and the untouched file still parses
    L2, _ = api.calc_parse_caffe(pp, mp)
    assert len(L2) == len(L)


def test_real_calc_model_if_present(pkg):
    """The published CALC model (get_model.sh:3-5 of the reference unpacks it into calc_model/) is not in this repository and cannot be
    downloaded here.  A maintainer who drops calc_model/deploy.prototxt + calc_model/calc.caffemodel at the repository root gets the real
    file through the dependency-free reader: the layer list must be a chain the library can run and must end in the 1064 values
    src/deeplcd.cpp:80 asserts.  (tools/pin_kit.md)"""
    import os
    from conftest import ROOT
    pp, mp = os.path.join(ROOT, "calc_model", "deploy.prototxt"), os.path.join(ROOT, "calc_model", "calc.caffemodel")
    if not (os.path.exists(pp) and os.path.exists(mp)):
        pytest.skip("calc_model/ (the published CALC weights) is not present")
    L, w = pkg.api.calc_parse_caffe(pp, mp)
    h, wd, ch = 120, 160, 1
    for l in L:
        t = int(l["type"])
        if t == 1:                                        # convolution
            k, s, p = int(l["kernel"]), int(l["stride"]), int(l["pad"])
            h, wd, ch = (h + 2 * p - k) // s + 1, (wd + 2 * p - k) // s + 1, int(l["num_output"])
        elif t == 3:                                      # max pooling, Caffe's ceil mode
            k, s = int(l["kernel"]), int(l["stride"])
            h, wd = -(-(h - k) // s) + 1, -(-(wd - k) // s) + 1
    assert h * wd * ch == 1064, (h, wd, ch)
    assert w.size > 0 and np.isfinite(w).all()
