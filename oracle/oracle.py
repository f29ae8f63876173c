"""ctypes binding of the CPU oracle (oracle/annonet_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (annonet_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def default_threads():
    """OpenMP threads for the oracle: the CPUs this process may run on, capped at 16.  A GPU box reports every core of the
    host (256) while a one-GPU job owns a share of 16; 256 threads on that share run the oracle ~10x slower."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


os.environ.setdefault("OMP_NUM_THREADS", str(default_threads()))   # before libgomp starts with the oracle library

IGNORE = 65535


class Layer(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("type", "k", "stride", "pad", "cin", "cout", "in_a", "in_b", "has_bn", "has_bias")] + \
               [(n, C.c_int64) for n in ("w_off", "b_off", "g_off", "beta_off", "rs_off")]


class Tile(C.Structure):
    _fields_ = [("full", C.c_long * 4), ("unique", C.c_long * 4)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "annonet_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_last_error.restype = C.c_char_p
        L.orc_net_create.restype = C.c_void_p
        L.orc_net_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_net_destroy.argtypes = [C.c_void_p]
        L.orc_net_layer_count.argtypes = [C.c_void_p]
        L.orc_net_layer.argtypes = [C.c_void_p, C.c_int, C.POINTER(Layer)]
        for f in ("orc_net_param_count", "orc_net_running_count"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("orc_net_params", "orc_net_momentum", "orc_net_grads", "orc_net_running"):
            getattr(L, f).restype = C.POINTER(C.c_float)
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_net_running_updates.restype = C.POINTER(C.c_double)
        L.orc_net_running_updates.argtypes = [C.c_void_p]
        L.orc_net_set_hyper.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_ulong]
        L.orc_net_set_bf16_emulation.argtypes = [C.c_void_p, C.c_int]
        L.orc_net_set_conv_algorithm.argtypes = [C.c_void_p, C.c_int]
        L.orc_required_input_dim.argtypes = [C.c_void_p]
        L.orc_recommended_input_dim.argtypes = [C.c_int, C.c_int]
        L.orc_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_layer_output.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.orc_layer_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_layer_dact.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
        L.orc_train_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.orc_set_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.orc_random_rect_containing_point.argtypes = [C.c_uint32, C.c_uint32, C.c_long, C.c_long, C.c_long, C.c_long, C.POINTER(C.c_long)]
        L.orc_outpaint.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long, C.c_long]
        L.orc_get_tiles.restype = C.c_int64
        L.orc_get_tiles.argtypes = [C.c_long] * 6 + [C.POINTER(Tile), C.c_int64]
        L.orc_infer.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_long, C.c_long, C.c_long, C.c_long, C.POINTER(Tile), C.c_int64, C.c_void_p, C.c_void_p]
        P = C.c_void_p
        L.orc_op_conv_forward.argtypes = [C.c_int] * 9 + [P] * 8 + [C.c_int, P]
        L.orc_op_conv_backward_data.argtypes = [C.c_int] * 9 + [P, P, C.c_int, P]
        L.orc_op_conv_backward_filter.argtypes = [C.c_int] * 9 + [P] * 7 + [C.c_int, P]
        L.orc_count_steps_without_decrease.restype = C.c_int64
        L.orc_count_steps_without_decrease.argtypes = [C.c_void_p, C.c_int64, C.c_double]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleNet:
    def __init__(self, levels=2, in_ch=3, classes=3, scaler=1.0, min_filters=1):
        self.L = lib()
        self.h = self.L.orc_net_create(levels, in_ch, classes, scaler, min_filters)
        if not self.h:
            raise RuntimeError(self.L.orc_last_error().decode())
        self.levels, self.in_ch, self.classes = levels, in_ch, classes
        n = self.L.orc_net_layer_count(self.h)
        self.layers = []
        for i in range(n):
            l = Layer()
            self.L.orc_net_layer(self.h, i, C.byref(l))
            self.layers.append(l)
        self.n_params = self.L.orc_net_param_count(self.h)
        self.n_running = self.L.orc_net_running_count(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_net_destroy(self.h)
            self.h = None

    def _view(self, fn, n):
        return np.ctypeslib.as_array(fn(self.h), shape=(n,))

    @property
    def params(self):
        return self._view(self.L.orc_net_params, self.n_params)

    @property
    def momentum(self):
        return self._view(self.L.orc_net_momentum, self.n_params)

    @property
    def grads(self):
        return self._view(self.L.orc_net_grads, self.n_params)

    @property
    def running(self):
        return self._view(self.L.orc_net_running, self.n_running)

    @property
    def running_updates(self):
        nbn = sum(1 for l in self.layers if l.has_bn)
        return np.ctypeslib.as_array(self.L.orc_net_running_updates(self.h), shape=(nbn,))

    def set_hyper(self, lr=0.1, wd=0.0005, mom=0.9, bn_window=100):
        self.L.orc_net_set_hyper(self.h, lr, wd, mom, bn_window)

    def set_bf16_emulation(self, on=True):
        """Restate the ANH_BF16 storage points: weights, conv inputs and stored gradients rounded to bf16; training passes store the
        raw conv outputs, inference passes the activations (on=2: raw outputs in inference too, the library under
        ANH_INFER_POST_ACT=0)."""
        self.L.orc_net_set_bf16_emulation(self.h, int(on))

    def set_conv_algorithm(self, algo):
        """0: direct convolution loops (the parity oracle, one k-ordered fmaf chain per output); 1: im2col + blocked SGEMM with OpenMP
        reductions — dlib's CPU design (cpu_dlib.cpp + BLAS), the form bench.py times as cpu_baseline."""
        self.L.orc_net_set_conv_algorithm(self.h, int(algo))

    def required_input_dim(self):
        return self.L.orc_required_input_dim(self.h)

    def recommended_input_dim(self, n):
        return self.L.orc_recommended_input_dim(self.levels, n)

    def forward(self, images):
        """images: u8 [N,H,W,C] -> logits fp32 [N,K,H,W] (inference mode)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n, h, w, c = images.shape
        assert c == self.in_ch
        out = np.empty((n, self.classes, h, w), dtype=np.float32)
        _check(self.L.orc_forward(self.h, _p(images), n, h, w, _p(out)))
        return out

    def layer_output(self, i, which=0):
        dims = (C.c_int * 4)()
        self.L.orc_layer_dims(self.h, i, dims)
        out = np.empty(tuple(dims), dtype=np.float32)
        _check(self.L.orc_layer_output(self.h, i, which, _p(out), out.size))
        return out

    def layer_dact(self, i):
        dims = (C.c_int * 4)()
        self.L.orc_layer_dims(self.h, i, dims)
        out = np.empty(tuple(dims), dtype=np.float32)
        _check(self.L.orc_layer_dact(self.h, i, _p(out), out.size))
        return out

    def train_step(self, images, labels, weights, loss_scale_n=None, apply_update=True):
        images = np.ascontiguousarray(images, dtype=np.uint8)
        labels = np.ascontiguousarray(labels, dtype=np.uint16)
        weights = np.ascontiguousarray(weights, dtype=np.float32)
        n, h, w, c = images.shape
        loss = C.c_double(0)
        _check(self.L.orc_train_step(self.h, _p(images), _p(labels), _p(weights), n, h, w,
                                     float(loss_scale_n if loss_scale_n else n), int(apply_update), C.byref(loss)))
        return loss.value

    def infer(self, image, gains=None, detection_levels=None, max_tile=(1024, 1024), overlap=None, tiles=None,
              want_blended=False):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        H, W, c = image.shape
        ov = self.required_input_dim() if overlap is None else overlap
        res = np.empty((H, W), dtype=np.uint16)
        bl = np.empty((self.classes, H, W), dtype=np.float32) if want_blended else None
        g = np.ascontiguousarray(gains, dtype=np.float64) if gains is not None else None
        d = np.ascontiguousarray(detection_levels, dtype=np.float64) if detection_levels is not None else None
        tarr, nt = None, 0
        if tiles is not None:
            nt = len(tiles)
            tarr = (Tile * nt)()
            for i, (full, uniq) in enumerate(tiles):
                tarr[i].full[:] = full
                tarr[i].unique[:] = uniq
        _check(self.L.orc_infer(self.h, _p(image), H, W, _p(g), _p(d), max_tile[0], max_tile[1], ov, ov, tarr, nt, _p(res), _p(bl)))
        return (res, bl) if want_blended else res


def set_weights(labels, class_weight, image_weight):
    labels = np.ascontiguousarray(labels, dtype=np.uint16)
    out = np.empty(labels.shape, dtype=np.float32)
    _check(lib().orc_set_weights(_p(labels), labels.shape[0], labels.shape[1], class_weight, image_weight, _p(out)))
    return out


def random_rect_containing_point(draw_x, draw_y, px, py, w, h):
    r = (C.c_long * 4)()
    _check(lib().orc_random_rect_containing_point(draw_x, draw_y, px, py, w, h, r))
    return tuple(r)


def outpaint(img, inside):
    img = np.ascontiguousarray(img, dtype=np.uint8).copy()
    nr, nc = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    _check(lib().orc_outpaint(_p(img), nr, nc, ch, *inside))
    return img


def get_tiles(width, height, max_w, max_h, ov_x, ov_y):
    n = lib().orc_get_tiles(width, height, max_w, max_h, ov_x, ov_y, None, 0)
    if n < 0:
        raise RuntimeError(lib().orc_last_error().decode())
    arr = (Tile * max(n, 1))()
    n = lib().orc_get_tiles(width, height, max_w, max_h, ov_x, ov_y, arr, n)
    return [(tuple(arr[i].full), tuple(arr[i].unique)) for i in range(n)]


def noise_draw(seed, counter, level):
    """The counter-based draw of the device's add_random_noise (kernels.h crop_noise_draw): splitmix64 finaliser -> [-level, level]"""
    m = (1 << 64) - 1
    z = (int(seed) + (int(counter) + 1) * 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return int(z % (2 * level + 1)) - level


def crop_sample(image, labels, left, top, dim, flip_lr, flip_ud, brightness_change, class_weight, image_weight,
                further_downscaling_factor=1.0, noise_level=0, noise_seed=0, color_offset=(0, 0, 0)):
    """randomly_crop_image (annonet_train_main.cpp:110-232) for given draws, composed from the oracle's own pieces: integer-rect
    chip of round(dim * further_downscaling_factor) pixels, outpaint (annonet.h:74-120), labels "ignore" outside the image
    (:150-158), the resize to dim x dim when downscaling (:160-171; dlib::resize_image restated [UPSTREAM-UNVERIFIED]: corner-aligned
    grid, bilinear in float rounded half up for the image, nearest neighbour for the labels), set_weights (:178), flips (:183-194),
    multiplicative brightness (:196-216; tuc::round taken as round-half-up [UPSTREAM-UNVERIFIED: tuc is not in the snapshot]),
    add_random_noise (:73-105,218-222, with the device's counter-based draws) and the colour offset (:226-231, offsets given).
    -> (image u8, labels u16, weights f32)"""
    image = np.asarray(image, dtype=np.uint8)
    labels = np.asarray(labels, dtype=np.uint16)
    H, W = labels.shape
    f = 1.0 if not further_downscaling_factor else float(further_downscaling_factor)
    src = int(np.floor(dim * f + 0.5)) if f != 1.0 else dim      # std::round
    chip = np.zeros((src, src) + image.shape[2:], dtype=np.uint8)
    lab = np.full((src, src), 65535, dtype=np.uint16)
    y0, y1, x0, x1 = max(top, 0), min(top + src, H), max(left, 0), min(left + src, W)   # valid_rect_in_full_image
    if y0 < y1 and x0 < x1:
        chip[y0 - top:y1 - top, x0 - left:x1 - left] = image[y0:y1, x0:x1]
        lab[y0 - top:y1 - top, x0 - left:x1 - left] = labels[y0:y1, x0:x1]
        chip = outpaint(chip, (x0 - left, y0 - top, x1 - left - 1, y1 - top - 1))
    else:   # the rectangle misses the image entirely: the device clamps to the nearest edge pixel
        ys = np.clip(np.arange(top, top + src), 0, H - 1)
        xs = np.clip(np.arange(left, left + src), 0, W - 1)
        chip = image[np.ix_(ys, xs)]
    if src != dim:
        scale = (src - 1) / max(dim - 1, 1)
        pos = np.arange(dim) * scale
        near = np.floor(pos + 0.5).astype(np.int64)
        lab = lab[np.ix_(near, near)]
        i0 = np.floor(pos).astype(np.int64)
        i1 = np.minimum(i0 + 1, src - 1)
        fr = (pos - i0).astype(np.float32)
        c = chip.astype(np.float32).reshape(src, src, -1)
        one = np.float32(1.0)
        fx = fr[None, :, None]
        fy = fr[:, None, None]
        topv = (one - fx) * c[np.ix_(i0, i0)] + fx * c[np.ix_(i0, i1)]
        botv = (one - fx) * c[np.ix_(i1, i0)] + fx * c[np.ix_(i1, i1)]
        val = (one - fy) * topv + fy * botv + np.float32(0.5)
        chip = val.astype(np.int32).astype(np.uint8).reshape((dim, dim) + image.shape[2:])
    weights = set_weights(lab, class_weight, image_weight)
    if flip_lr:
        chip, lab, weights = chip[:, ::-1], lab[:, ::-1], weights[:, ::-1]
    if flip_ud:
        chip, lab, weights = chip[::-1], lab[::-1], weights[::-1]
    if brightness_change != 1.0:
        chip = np.floor(np.clip(chip.astype(np.float64) * float(brightness_change), 0.0, 255.0) + 0.5).astype(np.uint8)
    chip = np.ascontiguousarray(chip)
    if noise_level > 0:
        flat = chip.reshape(-1).astype(np.int64)
        draws = np.array([noise_draw(noise_seed, i, noise_level) for i in range(flat.size)], dtype=np.int64)
        chip = np.clip(flat + draws, 0, 255).astype(np.uint8).reshape(chip.shape)
    if chip.ndim == 3 and any(color_offset):
        chip = np.clip(chip.astype(np.int64) + np.asarray(color_offset, dtype=np.int64)[None, None, :], 0, 255).astype(np.uint8)
    return np.ascontiguousarray(chip), np.ascontiguousarray(lab), np.ascontiguousarray(weights)


def _dlib_int(v):
    """dlib/serialize.h integer framing [UPSTREAM-UNVERIFIED]: control byte = byte count | 0x80 if negative, magnitude little-endian."""
    neg, m = v < 0, abs(int(v))
    body = m.to_bytes(max(1, (m.bit_length() + 7) // 8), "little")
    return bytes([len(body) | (0x80 if neg else 0)]) + body


def _dlib_read_int(buf, pos):
    ctl = buf[pos]
    n = ctl & 0x0F
    m = int.from_bytes(buf[pos + 1:pos + 1 + n], "little")
    return (-m if ctl & 0x80 else m), pos + 1 + n


def dnn_envelope_pack(classes_json, downscaling_factor, net_blob):
    """annonet.dnn (annonet_train_main.cpp:557-565) restated in pure Python: dlib serialize of (string, double, string);
    the double goes as (int64 mantissa, int16 exponent) from frexp with 53 digits, zero low bytes folded into the exponent."""
    import math
    v = float(downscaling_factor)
    if v == math.inf:
        man, ex = 0, 32000
    elif v == -math.inf:
        man, ex = 0, 32001
    elif v != v:
        man, ex = 0, 32002
    else:
        f, e = math.frexp(v)
        man, ex = int(f * (1 << 53)), e - 53
        for _ in range(8):
            if man & 0xFF:
                break
            man >>= 8     # Python's >> on a negative int is arithmetic, like the int64 shift it restates
            ex += 8
    js, net = bytes(classes_json), bytes(net_blob)
    return _dlib_int(len(js)) + js + _dlib_int(man) + _dlib_int(ex) + _dlib_int(len(net)) + net


def dnn_envelope_unpack(data):
    import math
    n, pos = _dlib_read_int(data, 0)
    js, pos = data[pos:pos + n], pos + n
    man, pos = _dlib_read_int(data, pos)
    ex, pos = _dlib_read_int(data, pos)
    factor = {32000: math.inf, 32001: -math.inf, 32002: math.nan}.get(ex)
    if factor is None:
        factor = math.ldexp(float(man), ex)
    n, pos = _dlib_read_int(data, pos)
    return js, factor, data[pos:pos + n]


def ignore_large_nonzero_regions(labels, receptive_field_side, by_area=np.inf, by_width=np.inf, by_height=np.inf):
    """annonet_train_main.cpp:434-502 restated with scipy.ndimage (per label value, 8-connectivity): blobs of equal label
    (background = 0 or 65535, annonet.h:26-37) with more than by_area*rf^2 pixels, wider than by_width*rf or taller than
    by_height*rf become 65535.  Returns (labels, pixels ignored)."""
    from scipy import ndimage
    lab = np.array(labels, dtype=np.uint16)
    rf = float(receptive_field_side)
    ignored = 0
    for value in np.unique(lab):
        if value == 0 or value == 65535:
            continue
        blobs, n = ndimage.label(lab == value, structure=np.ones((3, 3), dtype=int))
        for b, sl in enumerate(ndimage.find_objects(blobs), start=1):
            mask = blobs[sl] == b
            count = int(mask.sum())
            height, width = sl[0].stop - sl[0].start, sl[1].stop - sl[1].start
            if count > by_area * rf * rf or width > by_width * rf or height > by_height * rf:
                lab[sl][mask] = 65535
                ignored += count
    return lab, ignored


def count_steps_without_decrease(values, probability_of_decrease=0.51):
    v = np.ascontiguousarray(values, dtype=np.float64)
    return lib().orc_count_steps_without_decrease(_p(v), v.size, probability_of_decrease)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _out_dim(desc, n):
    type_, k, stride, pad = desc[:4]
    return (n + 2 * pad - k) // stride + 1 if type_ == 0 else stride * (n - 1) + k - 2 * pad


def op_conv_forward(desc, xa, sa=None, ta=None, xb=None, sb=None, tb=None, filters=None, bias=None, bf16=False):
    """desc = (type, k, stride, pad, cin, cout); xa [N,H,W,Cin] fp32 (raw producer output); returns y [N,Ho,Wo,Cout]."""
    xa, sa, ta, xb, sb, tb, filters, bias = map(_f32, (xa, sa, ta, xb, sb, tb, filters, bias))
    n, h, w, _ = xa.shape
    y = np.empty((n, _out_dim(desc, h), _out_dim(desc, w), desc[5]), np.float32)
    _check(lib().orc_op_conv_forward(*desc, n, h, w, _p(xa), _p(sa), _p(ta), _p(xb), _p(sb), _p(tb), _p(filters), _p(bias), int(bf16), _p(y)))
    return y


def op_conv_backward_data(desc, dy, filters, in_hw, bf16=False):
    dy, filters = _f32(dy), _f32(filters)
    n = dy.shape[0]
    dx = np.empty((n, in_hw[0], in_hw[1], desc[4]), np.float32)
    _check(lib().orc_op_conv_backward_data(*desc, n, in_hw[0], in_hw[1], _p(dy), _p(filters), int(bf16), _p(dx)))
    return dx


def op_conv_backward_filter(desc, xa, sa=None, ta=None, xb=None, sb=None, tb=None, dy=None, bf16=False):
    xa, sa, ta, xb, sb, tb, dy = map(_f32, (xa, sa, ta, xb, sb, tb, dy))
    n, h, w, _ = xa.shape
    dw = np.empty(desc[1] * desc[1] * desc[4] * desc[5], np.float32)
    _check(lib().orc_op_conv_backward_filter(*desc, n, h, w, _p(xa), _p(sa), _p(ta), _p(xb), _p(sb), _p(tb), _p(dy), int(bf16), _p(dw)))
    return dw


def bf16_round(a):
    """Round-to-nearest-even to bf16, returned as fp32 (numpy)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)
