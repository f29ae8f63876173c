"""Host-side mirror of the reference's NetPimpl interface (dlib-dnn-pimpl-wrapper/NetPimpl.h, known from its call
sites: annonet_infer.cpp:49-100, annonet_train_main.cpp:376-410,558-609, annonet_infer_main.cpp:347-351) plus
annonet_infer() (annonet_infer.h:34-42), set_weights() and tiling::get_tiles, over the C ABI.

Method names follow the C++ ones so that tests read like the reference's own host code.  numpy arrays stand in for
dlib::matrix (row-major, u8 HWC images, u16 label images).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ANH_BF16, ANH_FP32, LABEL_IGNORE, AnnonetHipError, NetConfig, check

label_to_ignore = LABEL_IGNORE


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def net_config(levels=2, in_channels=3, classes=3, width_scaler=1.0, min_filters=1, precision=ANH_BF16):
    return NetConfig(levels, in_channels, classes, width_scaler, min_filters, precision)


def set_devices(devices):
    """anh_set_devices: handles created afterwards on this thread hold one replica per listed device ([] = back to one device)."""
    arr = (C.c_int * max(len(devices), 1))(*devices)
    check(_lib.lib().anh_set_devices(arr, len(devices)))


def shard_range(n, world, rank):
    lo, hi = C.c_int64(), C.c_int64()
    check(_lib.lib().anh_shard_range(n, world, rank, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def cross_replica_overlaps(tiles, world, width, height):
    """tiles: list of (full, unique) rectangles -> sorted list of (l, t, r, b): where tiles of different replicas overlap"""
    arr = (_lib.Tile * max(len(tiles), 1))()
    for i, (full, uniq) in enumerate(tiles):
        arr[i].full_rect = _lib.Rect(*full)
        arr[i].unique_rect = _lib.Rect(*uniq)
    rects, n = C.POINTER(_lib.Rect)(), C.c_size_t()
    check(_lib.lib().anh_cross_replica_overlaps(arr, len(tiles), world, width, height, C.byref(rects), C.byref(n)))
    try:
        return [rects[i].tuple() for i in range(n.value)]
    finally:
        _lib.lib().anh_free(rects)


def net_layers(cfg):
    L = _lib.lib()
    n = L.anh_net_layer_count(C.byref(cfg))
    if n < 0:
        check(1)
    out = []
    for i in range(n):
        d = _lib.LayerDesc()
        check(L.anh_net_layer(C.byref(cfg), i, C.byref(d)))
        out.append(d)
    return out


class tiling:
    """tiling::parameters / tiling::get_tiles (annonet_infer.cpp:42, annonet_infer_main.cpp:423-427)."""

    class parameters:
        def __init__(self, max_tile_width=1024, max_tile_height=1024, overlap_x=0, overlap_y=0):
            self.max_tile_width, self.max_tile_height = max_tile_width, max_tile_height
            self.overlap_x, self.overlap_y = overlap_x, overlap_y

        def _c(self):
            return _lib.TilingParams(self.max_tile_width, self.max_tile_height, self.overlap_x, self.overlap_y)

    @staticmethod
    def get_tiles(width, height, params):
        L = _lib.lib()
        arr = C.POINTER(_lib.Tile)()
        n = C.c_size_t(0)
        p = params._c()
        check(L.anh_get_tiles(width, height, C.byref(p), C.byref(arr), C.byref(n)))
        try:
            return [(arr[i].full_rect.tuple(), arr[i].unique_rect.tuple()) for i in range(n.value)]
        finally:
            L.anh_free(arr)


WLABEL = np.dtype([("label", np.uint16), ("weight", np.float32)], align=True)   # NetPimpl::training_label_type's element


def set_weights(unweighted_label_image, class_weight, image_weight):
    """set_weights() (annonet_train.h:20-83) -> (labels u16, weights f32) = NetPimpl::training_label_type."""
    lab = np.ascontiguousarray(unweighted_label_image, dtype=np.uint16)
    out = np.zeros(lab.shape, dtype=WLABEL)
    assert out.itemsize == C.sizeof(_lib.WLabel)
    check(_lib.lib().anh_set_weights(_ptr(lab), lab.shape[0], lab.shape[1], class_weight, image_weight, _ptr(out)))
    return out


def random_rect_containing_point(draw_x, draw_y, px, py, width, height):
    r = _lib.Rect()
    check(_lib.lib().anh_random_rect_containing_point(draw_x, draw_y, px, py, width, height, C.byref(r)))
    return r.tuple()


def outpaint(img, inside):
    img = np.ascontiguousarray(img, dtype=np.uint8).copy()
    ch = 1 if img.ndim == 2 else img.shape[2]
    r = _lib.Rect(*inside)
    check(_lib.lib().anh_outpaint(_ptr(img), img.shape[0], img.shape[1], ch, C.byref(r)))
    return img


def ignore_large_nonzero_regions(label_image, receptive_field_side, by_area=float("inf"), by_width=float("inf"), by_height=float("inf")):
    """annonet_train_main.cpp:434-502: returns (label image with the too-large blobs set to LABEL_IGNORE, pixels ignored)."""
    lab = np.ascontiguousarray(label_image, dtype=np.uint16).copy()
    n = C.c_int64(0)
    check(_lib.lib().anh_ignore_large_nonzero_regions(_ptr(lab), lab.shape[0], lab.shape[1], by_area, by_width, by_height, receptive_field_side, C.byref(n)))
    return lab, n.value


class Dataset:
    """Full images + label images resident in HBM; training crops are cut on the device (randomly_crop_image,
    annonet_train_main.cpp:110-232, draws supplied by the caller).
    A crop spec is (image index, left, top, flip_left_right, flip_upside_down, brightness_change) optionally followed by
    (further_downscaling_factor, noise_level, noise_seed, (red, green, blue) colour offsets)."""

    def __init__(self, channels=3):
        self.L = _lib.lib()
        self.channels = channels
        self.h = C.c_void_p()
        check(self.L.anh_dataset_create(channels, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.anh_dataset_destroy(self.h)
            self.h = None

    def add(self, image, labels):
        img = np.ascontiguousarray(image, dtype=np.uint8)
        lab = np.ascontiguousarray(labels, dtype=np.uint16)
        assert img.shape[:2] == lab.shape and (img.ndim == 2) == (self.channels == 1)
        idx = C.c_int()
        check(self.L.anh_dataset_add(self.h, _ptr(img), _ptr(lab), lab.shape[0], lab.shape[1], C.byref(idx)))
        return idx.value

    def remove(self, index):
        """frees that image's HBM; its index is handed out again by a later add()"""
        check(self.L.anh_dataset_remove(self.h, int(index)))

    def resident_bytes(self):
        n = C.c_uint64()
        check(self.L.anh_dataset_resident_bytes(self.h, C.byref(n)))
        return int(n.value)

    @staticmethod
    def _specs(specs):
        arr = (_lib.CropSpec * len(specs))()
        for a, s in zip(arr, specs):   # (image, left, top, flip_lr, flip_ud, brightness[, further downscaling, noise level, noise seed, (dr, dg, db)])
            a.image, a.left, a.top, a.flip_left_right, a.flip_upside_down, a.brightness_change = int(s[0]), int(s[1]), int(s[2]), int(s[3]), int(s[4]), float(s[5])
            if len(s) > 6:
                a.further_downscaling_factor, a.noise_level, a.noise_seed = float(s[6]), int(s[7]), int(s[8])
                for k in range(3):
                    a.color_offset[k] = int(s[9][k])
        return arr

    def crop_batch(self, specs, dim, classes, class_weight=0.5, image_weight=0.5):
        """-> (images n x dim x dim [x 3] u8, labels n x dim x dim u16, weights n x dim x dim f32)"""
        n = len(specs)
        shape = (n, dim, dim) if self.channels == 1 else (n, dim, dim, self.channels)
        img = np.empty(shape, dtype=np.uint8)
        wl = np.empty((n, dim, dim), dtype=WLABEL)
        check(self.L.anh_dataset_crop_batch(self.h, self._specs(specs), n, dim, classes, class_weight, image_weight, _ptr(img), _ptr(wl)))
        return img, wl["label"].copy(), wl["weight"].copy()


def dnn_envelope_pack(anno_classes_json, downscaling_factor, serialized_runtime_net):
    """The bytes of annonet.dnn (annonet_train_main.cpp:557-565): dlib serialize of (string, double, string)."""
    L = _lib.lib()
    js = anno_classes_json.encode() if isinstance(anno_classes_json, str) else bytes(anno_classes_json)
    net = bytes(serialized_runtime_net)
    out, n = C.c_void_p(), C.c_size_t()
    check(L.anh_dnn_envelope_pack(js, len(js), float(downscaling_factor), C.create_string_buffer(net, len(net)), len(net), C.byref(out), C.byref(n)))
    try:
        return C.string_at(out, n.value)
    finally:
        L.anh_free(out)


def dnn_envelope_unpack(file_bytes):
    """annonet_infer_main.cpp:340-351: (anno_classes_json bytes, downscaling factor, serialized RuntimeNet bytes)."""
    L = _lib.lib()
    data = bytes(file_bytes)
    js, jn, f, net, nn = C.c_void_p(), C.c_size_t(), C.c_double(), C.c_void_p(), C.c_size_t()
    check(L.anh_dnn_envelope_unpack(C.create_string_buffer(data, len(data)), len(data), C.byref(js), C.byref(jn), C.byref(f), C.byref(net), C.byref(nn)))
    try:
        return C.string_at(js, jn.value), f.value, C.string_at(net, nn.value)
    finally:
        L.anh_free(js); L.anh_free(net)


def count_steps_without_decrease(values, probability_of_decrease=0.51):
    v = np.ascontiguousarray(values, dtype=np.float64)
    return _lib.lib().anh_count_steps_without_decrease(_ptr(v), v.size, probability_of_decrease)


class _Profiled:
    _is_trainer = 0

    def profile_enable(self, on=True):
        check(self.L.anh_profile_enable(self.h, self._is_trainer, int(on)))

    def profile_set_filter(self, substring=""):
        check(self.L.anh_profile_set_filter(self.h, self._is_trainer, (substring or "").encode()))

    def profile_set_sampling(self, every=1):
        check(self.L.anh_profile_set_sampling(self.h, self._is_trainer, int(every)))

    def profile_reset(self):
        check(self.L.anh_profile_reset(self.h, self._is_trainer))

    def profile_launch_order(self):
        """Entry names of the most recent pass in host enqueue order (one per kernel-class launch)."""
        need = C.c_size_t()
        check(self.L.anh_profile_launch_order(self.h, self._is_trainer, None, 0, C.byref(need)))
        buf = C.create_string_buffer(max(need.value, 1))
        check(self.L.anh_profile_launch_order(self.h, self._is_trainer, buf, need.value, None))
        return [x for x in buf.value.decode().split("\n") if x]

    def profile(self):
        n = self.L.anh_profile_count(self.h, self._is_trainer)
        if n < 0:
            check(1)
        out = []
        for i in range(n):
            name = C.create_string_buffer(128)
            ms, cnt, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            check(self.L.anh_profile_entry(self.h, self._is_trainer, i, name, 128, C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by)))
            out.append({"name": name.value.decode(), "total_ms": ms.value, "launches": cnt.value, "flops": fl.value, "bytes": by.value})
        return out


class RuntimeNet(_Profiled):
    """NetPimpl::RuntimeNet."""

    def __init__(self, cfg=None, _handle=None):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        if _handle is not None:
            self.h = _handle
        else:
            cfg = cfg or net_config()
            check(self.L.anh_runtime_create(C.byref(cfg), C.byref(self.h)))
        c = NetConfig()
        check(self.L.anh_runtime_config(self.h, C.byref(c)))
        self.cfg = c
        self.n_params = self.L.anh_net_param_count(C.byref(c))
        self.n_running = self.L.anh_net_running_count(C.byref(c))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.anh_runtime_destroy(self.h)
            self.h = None

    @staticmethod
    def GetRecommendedInputDimension(levels, n):
        return _lib.lib().anh_recommended_input_dim(levels, n)

    def set_params(self, params, running):
        p = np.ascontiguousarray(params, dtype=np.float32)
        r = np.ascontiguousarray(running, dtype=np.float32)
        check(self.L.anh_runtime_set_params(self.h, _ptr(p), p.size, _ptr(r), r.size))

    def get_params(self):
        p = np.empty(self.n_params, np.float32)
        r = np.empty(self.n_running, np.float32)
        check(self.L.anh_runtime_get_params(self.h, _ptr(p), p.size, _ptr(r), r.size))
        return p, r

    def Serialize(self):
        blob, n = C.c_void_p(), C.c_size_t()
        check(self.L.anh_runtime_serialize(self.h, C.byref(blob), C.byref(n)))
        try:
            return C.string_at(blob, n.value)
        finally:
            self.L.anh_free(blob)

    @classmethod
    def Deserialize(cls, data, precision=ANH_BF16):
        L = _lib.lib()
        h = C.c_void_p()
        buf = C.create_string_buffer(data, len(data))
        check(L.anh_runtime_deserialize(buf, len(data), precision, C.byref(h)))
        return cls(_handle=h)

    def Forward(self, input_tile):
        """u8 [H,W,C] (or [N,H,W,C]) -> fp32 [K,H,W] (or [N,K,H,W]); the returned array is a copy."""
        img = np.ascontiguousarray(input_tile, dtype=np.uint8)
        single = img.ndim != 4
        if img.ndim == 2:
            img = img[None, :, :, None]
        elif img.ndim == 3:
            img = img[None]
        n, h, w, c = img.shape
        if c != self.cfg.in_channels:
            raise AnnonetHipError(1, "channel count does not match the net input")
        out = C.POINTER(C.c_float)()
        k, nr, nc = C.c_int(), C.c_int(), C.c_int()
        check(self.L.anh_runtime_forward(self.h, _ptr(img), n, h, w, C.byref(out), C.byref(k), C.byref(nr), C.byref(nc)))
        arr = np.ctypeslib.as_array(out, shape=(n, k.value, nr.value, nc.value)).copy()
        return arr[0] if single else arr

    def synchronize(self):
        check(self.L.anh_runtime_synchronize(self.h))

    def set_stream(self, stream_ptr):
        """Launch on the caller's (non-default) HIP stream.  The legacy default stream (0) cannot be named through this call:
        0 / None means "back to a stream of the handle's own".  To order torch work with the handle, use stream_ptr()."""
        check(self.L.anh_runtime_set_stream(self.h, stream_ptr or None))

    def stream_ptr(self):
        s = C.c_void_p()
        check(self.L.anh_runtime_get_stream(self.h, C.byref(s)))
        return s.value or 0

    def stores_activations(self):
        """whether the handle's last inference pass stored post-activation tensors (anh_runtime_stores_activations)"""
        yes = C.c_int()
        check(self.L.anh_runtime_stores_activations(self.h, C.byref(yes)))
        return bool(yes.value)


def annonet_infer(net, input_image, gains=None, detection_levels=None, tiling_parameters=None, want_blended=False):
    """annonet_infer() (annonet_infer.h:34-42): returns the u16 label image (and the blended class planes)."""
    img = np.ascontiguousarray(input_image, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    H, W, c = img.shape
    if c != net.cfg.in_channels:
        raise AnnonetHipError(1, "channel count does not match the net input")
    K = net.cfg.classes
    res = np.empty((H, W), np.uint16)
    bl = np.empty((K, H, W), np.float32) if want_blended else None
    g = np.ascontiguousarray(gains, dtype=np.float64) if gains is not None else None
    d = np.ascontiguousarray(detection_levels, dtype=np.float64) if detection_levels is not None else None
    if g is not None and g.size != K or d is not None and d.size != K:
        raise AnnonetHipError(1, "gains / detection levels need one value per class")
    tp = tiling_parameters._c() if tiling_parameters is not None else None
    check(net.L.anh_infer(net.h, _ptr(img), H, W, _ptr(g), _ptr(d), C.byref(tp) if tp is not None else None, _ptr(res), _ptr(bl)))
    return (res, bl) if want_blended else res


def annonet_infer_device(net, d_image_ptr, height, width, d_labels_ptr, d_blended_ptr, gains=None, tiling_parameters=None, tiles=None):
    """annonet_infer() with the image, the label map and the blended planes resident in HBM (device pointers as ints).
    `tiles` (list of (full, unique) rects) restricts the call to a shard of the tile list (multi-GPU inference)."""
    g = np.ascontiguousarray(gains, dtype=np.float64) if gains is not None else None
    tp = tiling_parameters._c() if tiling_parameters is not None else None
    arr, n = None, 0
    if tiles is not None:
        n = len(tiles)
        arr = (_lib.Tile * max(n, 1))()
        for i, (full, uniq) in enumerate(tiles):
            arr[i].full_rect = _lib.Rect(*full)
            arr[i].unique_rect = _lib.Rect(*uniq)
    check(net.L.anh_infer_device(net.h, d_image_ptr, height, width, _ptr(g), C.byref(tp) if tp is not None else None, arr, n, d_labels_ptr, d_blended_ptr))


def argmax_device(net, d_blended_ptr, height, width, row0, row1, d_labels_ptr, gains=None):
    """find_label (annonet_infer.cpp:170-185) over rows [row0, row1) of device-resident blended planes."""
    g = np.ascontiguousarray(gains, dtype=np.float64) if gains is not None else None
    check(net.L.anh_argmax_device(net.h, d_blended_ptr, height, width, row0, row1, _ptr(g), d_labels_ptr))


class TrainingNet(_Profiled):
    """NetPimpl::TrainingNet (annonet_train_main.cpp:396-410)."""
    _is_trainer = 1

    def __init__(self, levels=2, in_channels=3, precision=ANH_BF16, seed=0):
        self.L = _lib.lib()
        self.h = C.c_void_p()
        check(self.L.anh_trainer_create(C.byref(self.h)))
        check(self.L.anh_trainer_set_levels(self.h, levels))
        check(self.L.anh_trainer_set_input_channels(self.h, in_channels))
        check(self.L.anh_trainer_set_precision(self.h, precision))
        check(self.L.anh_trainer_set_seed(self.h, seed))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.anh_trainer_destroy(self.h)
            self.h = None

    @property
    def cfg(self):
        c = NetConfig()
        check(self.L.anh_trainer_config(self.h, C.byref(c)))
        return c

    @property
    def n_params(self):
        c = self.cfg
        return self.L.anh_net_param_count(C.byref(c))

    @property
    def n_running(self):
        c = self.cfg
        return self.L.anh_net_running_count(C.byref(c))

    def GetRequiredInputDimension(self):
        c = self.cfg
        return self.L.anh_required_input_dim(C.byref(c))

    def Initialize(self):
        check(self.L.anh_trainer_initialize(self.h))

    def SetNetWidth(self, scaler, min_filter_count):
        check(self.L.anh_trainer_set_net_width(self.h, scaler, min_filter_count))

    def SetClassCount(self, n):
        check(self.L.anh_trainer_set_class_count(self.h, n))

    def SetLearningRate(self, lr):
        check(self.L.anh_trainer_set_learning_rate(self.h, lr))

    def SetLearningRateShrinkFactor(self, f):
        check(self.L.anh_trainer_set_learning_rate_shrink_factor(self.h, f))

    def SetIterationsWithoutProgressThreshold(self, n):
        check(self.L.anh_trainer_set_iterations_without_progress_threshold(self.h, n))

    def SetPreviousLossValuesDumpAmount(self, n):
        check(self.L.anh_trainer_set_previous_loss_values_dump_amount(self.h, n))

    def SetAllBatchNormalizationRunningStatsWindowSizes(self, n):
        check(self.L.anh_trainer_set_all_bn_running_stats_window_sizes(self.h, n))

    def SetSynchronizationFile(self, path, seconds):
        check(self.L.anh_trainer_set_synchronization_file(self.h, path.encode(), float(seconds)))

    def BeVerbose(self):
        check(self.L.anh_trainer_be_verbose(self.h))

    def set_sgd(self, weight_decay=0.0005, momentum=0.9):
        check(self.L.anh_trainer_set_sgd(self.h, weight_decay, momentum))

    def GetLearningRate(self):
        return self.L.anh_trainer_get_learning_rate(self.h)

    def get_last_loss(self):
        return self.L.anh_trainer_get_last_loss(self.h)

    def step_count(self):
        return self.L.anh_trainer_get_step_count(self.h)

    def StartTraining(self, samples, labels):
        """samples: list of u8 [H,W,C]; labels: list of structured (label, weight) arrays from set_weights()."""
        n = len(samples)
        if n == 0 or n != len(labels):
            raise AnnonetHipError(1, "samples and labels must be non-empty and of equal length")
        imgs = [np.ascontiguousarray(s, dtype=np.uint8) for s in samples]
        labs = [np.ascontiguousarray(l) for l in labels]
        h, w = imgs[0].shape[:2]
        for im, lb in zip(imgs, labs):
            if im.shape[:2] != (h, w) or lb.shape != (h, w) or lb.itemsize != C.sizeof(_lib.WLabel):
                raise AnnonetHipError(1, "all samples and label images of a mini-batch must share one size")
        ip = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        lp = (C.c_void_p * n)(*[lb.ctypes.data for lb in labs])
        check(self.L.anh_trainer_step(self.h, ip, lp, n, h, w))

    # ---- device-resident, data-parallel-friendly form ----
    def StartTrainingOnCrops(self, dataset, specs, dim, class_weight=0.5, image_weight=0.5):
        """StartTraining on crops cut on the device from `dataset` (no host mini-batch)."""
        check(self.L.anh_trainer_step_crops(self.h, dataset.h, Dataset._specs(specs), len(specs), dim, class_weight, image_weight))

    def forward_backward_device(self, d_images, d_labels, d_weights, n, h, w, loss_scale_n):
        check(self.L.anh_trainer_forward_backward_device(self.h, d_images, d_labels, d_weights, n, h, w, float(loss_scale_n)))

    def apply_update(self, grad_scale=1.0):
        check(self.L.anh_trainer_apply_update(self.h, float(grad_scale)))

    def grad_buffer(self):
        p, n = C.c_void_p(), C.c_int64()
        check(self.L.anh_trainer_grad_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def get_params(self):
        p = np.empty(self.n_params, np.float32)
        r = np.empty(self.n_running, np.float32)
        check(self.L.anh_trainer_get_params(self.h, _ptr(p), p.size, _ptr(r), r.size))
        return p, r

    def set_params(self, params, running):
        p = np.ascontiguousarray(params, dtype=np.float32)
        r = np.ascontiguousarray(running, dtype=np.float32)
        check(self.L.anh_trainer_set_params(self.h, _ptr(p), p.size, _ptr(r), r.size))

    def replicas(self):
        return self.L.anh_handle_replicas(self.h, 1)

    def exchange_stats(self):
        """anh_trainer_exchange_stats as a dict: host time per StartTraining, sampled all-reduce times on replica 0, worker wake-ups."""
        st = _lib.ExchangeStats()
        check(self.L.anh_trainer_exchange_stats(self.h, C.byref(st)))
        return {name: getattr(st, name) for name, _ in _lib.ExchangeStats._fields_}

    def reset_exchange_stats(self):
        self.L.anh_trainer_reset_exchange_stats(self.h)

    def replica_params(self, replica):
        p = np.empty(self.n_params, np.float32)
        check(self.L.anh_trainer_replica_params(self.h, replica, _ptr(p), p.size))
        return p

    def get_grads(self):
        g = np.empty(self.n_params, np.float32)
        check(self.L.anh_trainer_get_grads(self.h, _ptr(g), g.size))
        return g

    def get_momentum(self):
        m = np.empty(self.n_params, np.float32)
        check(self.L.anh_trainer_get_momentum(self.h, _ptr(m), m.size))
        return m

    def set_momentum(self, m):
        m = np.ascontiguousarray(m, dtype=np.float32)
        check(self.L.anh_trainer_set_momentum(self.h, _ptr(m), m.size))

    def GetRuntimeNet(self, precision=None):
        h = C.c_void_p()
        check(self.L.anh_trainer_snapshot_runtime(self.h, self.cfg.precision if precision is None else precision, C.byref(h)))
        return RuntimeNet(_handle=h)

    def save_state(self, path):
        check(self.L.anh_trainer_save_state(self.h, path.encode()))

    def load_state(self, path):
        check(self.L.anh_trainer_load_state(self.h, path.encode()))

    def synchronize(self):
        check(self.L.anh_trainer_synchronize(self.h))

    def set_stream(self, stream_ptr):
        """As RuntimeNet.set_stream: 0 / None = a stream of the handle's own, never the legacy default stream."""
        check(self.L.anh_trainer_set_stream(self.h, stream_ptr or None))

    def stream_ptr(self):
        s = C.c_void_p()
        check(self.L.anh_trainer_get_stream(self.h, C.byref(s)))
        return s.value or 0

    def early_grads(self):
        """first element of the gradient bucket's EARLY part: [first, n_params + 1) is final before backward has finished
        (anh_trainer_early_grads); first > n_params means there is none"""
        first = C.c_int64()
        check(self.L.anh_trainer_early_grads(self.h, C.byref(first)))
        return int(first.value)

    def step_graph_stats(self):
        """(captures, launches) of the captured backward graph (ANH_STEP_GRAPH=1); (0, 0) when off"""
        c, l = C.c_int64(0), C.c_int64(0)
        check(self.L.anh_trainer_step_graph_stats(self.h, C.byref(c), C.byref(l)))
        return int(c.value), int(l.value)

    def wait_early_grads(self, stream_ptr):
        """makes the caller's HIP stream wait until the early part of the bucket is final"""
        check(self.L.anh_trainer_wait_early_grads(self.h, stream_ptr))

    def layer_tensor(self, layer, which=0):
        dims = (C.c_int * 4)()
        check(self.L.anh_trainer_layer_tensor(self.h, layer, which, None, 0, dims))
        out = np.empty(tuple(dims), np.float32)
        check(self.L.anh_trainer_layer_tensor(self.h, layer, which, _ptr(out), out.size, dims))
        return out


# ---- single-layer ops (include/annonet_hip.h "single-layer ops"): kernel-level parity tests and micro-benchmarks ----
def _op_input(x, scale, shift, keep):
    if x is None:
        return None
    x = np.ascontiguousarray(x, dtype=np.float32)
    s = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    t = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    keep.extend([x, s, t])
    return _lib.OpInput(x.ctypes.data, s.ctypes.data if s is not None else None, t.ctypes.data if t is not None else None)


def _out_dim(desc, n):
    type_, k, stride, pad = desc[:4]
    if stride <= 0:
        return 0  # the library rejects the descriptor
    return (n + 2 * pad - k) // stride + 1 if type_ == 0 else stride * (n - 1) + k - 2 * pad


def op_conv_forward(precision, desc, xa, sa=None, ta=None, xb=None, sb=None, tb=None, filters=None, bias=None):
    """desc = (type, k, stride, pad, cin, cout).  Returns (y [N,Ho,Wo,Cout] fp32, used_mfma)."""
    keep = []
    a = _op_input(xa, sa, ta, keep)
    b = _op_input(xb, sb, tb, keep)
    n, h, w, _ = keep[0].shape
    d = _lib.ConvDesc(*desc)
    f = np.ascontiguousarray(filters, dtype=np.float32)
    bi = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    y = np.empty((n, _out_dim(desc, h), _out_dim(desc, w), desc[5]), np.float32)
    used = C.c_int(0)
    check(_lib.lib().anh_op_conv_forward(precision, C.byref(d), n, h, w, C.byref(a), C.byref(b) if b is not None else None,
                                         _ptr(f), _ptr(bi), _ptr(y), C.byref(used)))
    return y, bool(used.value)


def op_conv_backward_data(precision, desc, dy, filters, in_hw):
    d = _lib.ConvDesc(*desc)
    g = np.ascontiguousarray(dy, dtype=np.float32)
    f = np.ascontiguousarray(filters, dtype=np.float32)
    n = g.shape[0]
    dx = np.empty((n, in_hw[0], in_hw[1], desc[4]), np.float32)
    used = C.c_int(0)
    check(_lib.lib().anh_op_conv_backward_data(precision, C.byref(d), n, in_hw[0], in_hw[1], _ptr(g), _ptr(f), _ptr(dx), C.byref(used)))
    return dx, bool(used.value)


def op_conv_forward_stats(precision, desc, xa, sa=None, ta=None, xb=None, sb=None, tb=None, filters=None):
    """Forward conv with the fused bn-statistics epilogue.  Returns (y, sums [cout, 2] = (sum y, sum y^2), fused)."""
    keep = []
    a = _op_input(xa, sa, ta, keep)
    b = _op_input(xb, sb, tb, keep)
    n, h, w, _ = keep[0].shape
    d = _lib.ConvDesc(*desc)
    f = np.ascontiguousarray(filters, dtype=np.float32)
    y = np.empty((n, _out_dim(desc, h), _out_dim(desc, w), desc[5]), np.float32)
    sums = np.empty((desc[5], 2), np.float64)
    fused = C.c_int(0)
    check(_lib.lib().anh_op_conv_forward_stats(precision, C.byref(d), n, h, w, C.byref(a), C.byref(b) if b is not None else None,
                                               _ptr(f), _ptr(y), _ptr(sums), C.byref(fused)))
    return y, sums, bool(fused.value)


def op_conv_backward_data_bn(precision, desc, dy, filters, in_hw, y_prev, scale, shift, mean, invstd, dx_init=None):
    """Backward-data conv (+= dx_init) with the fused bn + relu backward reduction of the layer that receives dx.
    Returns (dx, sums [cin, 2] = (sum dz*xhat, sum dz), fused)."""
    d = _lib.ConvDesc(*desc)
    g = np.ascontiguousarray(dy, dtype=np.float32)
    f = np.ascontiguousarray(filters, dtype=np.float32)
    n = g.shape[0]
    arrs = [np.ascontiguousarray(v, dtype=np.float32) for v in (y_prev, scale, shift, mean, invstd)]
    init = None if dx_init is None else np.ascontiguousarray(dx_init, dtype=np.float32)
    dx = np.empty((n, in_hw[0], in_hw[1], desc[4]), np.float32)
    sums = np.empty((desc[4], 2), np.float64)
    fused = C.c_int(0)
    check(_lib.lib().anh_op_conv_backward_data_bn(precision, C.byref(d), n, in_hw[0], in_hw[1], _ptr(g), _ptr(f), _ptr(init),
                                                  *[_ptr(v) for v in arrs], _ptr(dx), _ptr(sums), C.byref(fused)))
    return dx, sums, bool(fused.value)


def op_conv_backward_filter_bn(precision, desc, image_u8, da, y, scale, shift, mean, invstd, coef):
    """Stem filter gradient with dy = bn + relu backward of (da, y) computed while staging.  Returns (dw, in_kernel)."""
    d = _lib.ConvDesc(*desc)
    img = np.ascontiguousarray(image_u8, dtype=np.uint8)
    n, h, w, _ = img.shape
    arrs = [np.ascontiguousarray(v, dtype=np.float32) for v in (da, y, scale, shift, mean, invstd, coef)]
    bn = _lib.OpBnDy(*[v.ctypes.data for v in arrs])
    dw = np.empty(desc[1] * desc[1] * desc[4] * desc[5], np.float32)
    flag = C.c_int(0)
    check(_lib.lib().anh_op_conv_backward_filter_bn(precision, C.byref(d), n, h, w, _ptr(img), C.byref(bn), _ptr(dw), C.byref(flag)))
    return dw, bool(flag.value)


def op_conv_backward_filter(precision, desc, xa, sa=None, ta=None, xb=None, sb=None, tb=None, dy=None):
    keep = []
    a = _op_input(xa, sa, ta, keep)
    b = _op_input(xb, sb, tb, keep)
    n, h, w, _ = keep[0].shape
    d = _lib.ConvDesc(*desc)
    g = np.ascontiguousarray(dy, dtype=np.float32)
    dw = np.empty(desc[1] * desc[1] * desc[4] * desc[5], np.float32)
    used = C.c_int(0)
    check(_lib.lib().anh_op_conv_backward_filter(precision, C.byref(d), n, h, w, C.byref(a), C.byref(b) if b is not None else None,
                                                 _ptr(g), _ptr(dw), C.byref(used)))
    return dw, bool(used.value)
