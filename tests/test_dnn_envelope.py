"""annonet.dnn envelope: dlib serialize framing of (string anno_classes_json, double downscaling_factor,
string serialized RuntimeNet) — written at annonet_train_main.cpp:557-565, read at annonet_infer_main.cpp:340-351.
dlib is absent from the reference snapshot, so the framing is restated from dlib/serialize.h + float_details.h as published
[UPSTREAM-UNVERIFIED]; the product (C++, hostlogic.cpp) and the oracle (pure Python, oracle/oracle.py) restate it independently."""
import math

import numpy as np
import pytest

import annonet_amd as aa
from oracle import oracle as orc


def test_hand_computed_bytes():
    # string = length (control byte 01 = one value byte) + bytes; 1.0 = frexp 0.5 * 2^53 = 2^52 -> six zero bytes folded:
    # mantissa 0x10, exponent -52 + 48 = -4 (control byte 0x81 = one byte, negative)
    assert aa.dnn_envelope_pack("ab", 1.0, b"xyz").hex() == "01026162" + "0110" + "8104" + "010378797a"
    # 0.0: mantissa 0 survives all eight folds: exponent -53 + 64 = 11
    assert aa.dnn_envelope_pack("", 0.0, b"").hex() == "0100" + "0100" + "010b" + "0100"
    # -0.75 = -24 * 2^-5
    assert aa.dnn_envelope_pack("", -0.75, b"").hex() == "0100" + "8118" + "8105" + "0100"
    # a 300-byte string needs a two-byte length: 0x012c little-endian
    assert aa.dnn_envelope_pack("x" * 300, 2.0, b"")[:3].hex() == "022c01"
    # 0.5 (a typical downscaling factor 1/2): mantissa 0x10, exponent -5
    assert aa.dnn_envelope_pack("", 0.5, b"").hex() == "0100" + "0110" + "8105" + "0100"


def test_round_trip_and_oracle_agreement():
    rng = np.random.default_rng(4)
    specials = [0.0, -0.0, 1.0, 0.25, 1.0 / 3.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, -123456.789, math.inf, -math.inf, math.nan]
    randoms = list(np.ldexp(rng.standard_normal(40), rng.integers(-200, 200, 40)))
    for k, factor in enumerate(specials + randoms):
        js = bytes(rng.integers(0, 256, int(rng.integers(0, 400)), dtype=np.uint8))
        net = bytes(rng.integers(0, 256, int(rng.integers(0, 70000)) if k % 7 == 0 else int(rng.integers(0, 300)), dtype=np.uint8))
        packed = aa.dnn_envelope_pack(js, factor, net)
        assert packed == orc.dnn_envelope_pack(js, factor, net)
        j2, f2, n2 = aa.dnn_envelope_unpack(packed)
        oj, of, on = orc.dnn_envelope_unpack(packed)
        assert j2 == js == oj and n2 == net == on
        if factor != factor:
            assert f2 != f2 and of != of
        else:
            assert f2 == factor == of          # every finite double survives exactly (53-bit mantissa)


def test_damaged_files_are_errors():
    good = aa.dnn_envelope_pack('{"anno_classes": []}', 1.0, b"0123456789")
    for cut in (0, 1, 5, len(good) - 1):
        with pytest.raises(aa.AnnonetHipError):
            aa.dnn_envelope_unpack(good[:cut])
    with pytest.raises(aa.AnnonetHipError):
        aa.dnn_envelope_unpack(b"\x09" + good[1:])      # nine length bytes: not a dlib integer
    with pytest.raises(aa.AnnonetHipError):
        aa.dnn_envelope_unpack(b"\x01\xff" + good[2:])  # string longer than the file


@pytest.mark.gpu
def test_deployable_net_round_trip():
    """GetRuntimeNet -> Serialize -> annonet.dnn -> Deserialize gives a net with the same outputs (annonet_infer_main.cpp:340-351)."""
    t = aa.TrainingNet(1, 3, aa.ANH_FP32, seed=5)
    t.SetNetWidth(0.25, 4); t.SetClassCount(3); t.Initialize()
    net = t.GetRuntimeNet()
    classes = '[{"name":"background"},{"name":"a"},{"name":"b"}]'
    file_bytes = aa.dnn_envelope_pack(classes, 0.5, net.Serialize())
    js, factor, blob = aa.dnn_envelope_unpack(file_bytes)
    assert js.decode() == classes and factor == 0.5
    net2 = aa.RuntimeNet.Deserialize(blob, precision=aa.ANH_FP32)
    d = aa.RuntimeNet.GetRecommendedInputDimension(1, 40)
    img = np.random.default_rng(0).integers(0, 256, (d, d, 3), dtype=np.uint8)
    np.testing.assert_array_equal(net.Forward(img), net2.Forward(img))
