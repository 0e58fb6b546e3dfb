"""ctypes view of libcubeslam_hip.so (include/cubeslam_hip.h) for the tests and bench.py.

This is plumbing, not a second implementation: every function here forwards to the C ABI, and
loading fails loudly when the HIP library has not been built (run __graft_entry__.build()).
"""
from __future__ import annotations

import os as _os
# (five streams per detector: with the ROCm runtime's default of four hardware queues they share queues and serialise -- INTEGRATION.md; only effective
# when nothing has initialised HIP yet, and a value already in the environment wins)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcubeslam_hip.so")


class CsDetectParams(C.Structure):
    _fields_ = [
        ("consider_config_1", C.c_int), ("consider_config_2", C.c_int),
        ("whether_sample_cam_roll_pitch", C.c_int), ("whether_sample_bbox_height", C.c_int),
        ("max_cuboid_num", C.c_int), ("nominal_skew_ratio", C.c_double), ("max_cut_skew", C.c_double),
        ("yaw_range_deg", C.c_double), ("yaw_step_deg", C.c_double),
        ("vp12_edge_angle_thre", C.c_double), ("vp3_edge_angle_thre", C.c_double), ("shorted_edge_thre", C.c_double),
        ("weight_vp_angle", C.c_double), ("weight_skew_error", C.c_double),
        ("pre_merge_dist_thre", C.c_double), ("pre_merge_angle_thre", C.c_double), ("edge_length_threshold", C.c_double),
        ("host_threads", C.c_int),
    ]


class CsCuboid(C.Structure):
    _fields_ = [
        ("pos", C.c_double * 3), ("scale", C.c_double * 3), ("rotY", C.c_double),
        ("box_config_type", C.c_double * 2), ("box_corners_2d", C.c_int32 * 16),
        ("box_corners_3d_world", C.c_double * 24), ("rect_detect_2d", C.c_double * 4),
        ("edge_distance_error", C.c_double), ("edge_angle_error", C.c_double),
        ("normalized_error", C.c_double), ("skew_ratio", C.c_double), ("down_expand_height", C.c_double),
        ("camera_roll_delta", C.c_double), ("camera_pitch_delta", C.c_double),
    ]


class CsRoi(C.Structure):
    _fields_ = [("left", C.c_int), ("top", C.c_int), ("width", C.c_int), ("height", C.c_int), ("down_expand", C.c_int)]


class CsFrameDesc(C.Structure):
    _fields_ = [
        ("K", C.POINTER(C.c_double)), ("T_wc", C.POINTER(C.c_double)), ("img_w", C.c_int), ("img_h", C.c_int),
        ("boxes", C.POINTER(C.c_double)), ("n_boxes", C.c_int), ("lines", C.POINTER(C.c_double)), ("n_lines", C.c_int),
        ("dist_maps", C.POINTER(C.POINTER(C.c_float))),
    ]


class CsDetectTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("setup_host_ms", "h2d_ms", "vp_kernel_ms", "cand_kernel_ms", "compact_ms", "d2h_ms",
                                          "rank_host_ms", "finalize_ms", "total_ms")] + \
               [("n_jobs", C.c_longlong), ("n_slots", C.c_longlong), ("n_valid", C.c_longlong),
                ("cand_kernel_bytes", C.c_longlong), ("cand_kernel_launches", C.c_int), ("n_fallback_boxes", C.c_int), ("n_redo_frames", C.c_int),
                ("rank_kernel_ms", C.c_double), ("line_setup_ms", C.c_double), ("score_kernel_ms", C.c_double), ("score_kernel_bytes", C.c_longlong)]


# every symbol include/cubeslam_hip.h declares (tests/test_capi_symbols.py checks the export table)
DECLARED_SYMBOLS = [
    "cs_last_error", "cs_device_count", "cs_diag_build", "cs_detect_default_params", "cs_box_rois", "cs_cam_euler_zyx", "cs_detector_create",
    "cs_detector_destroy", "cs_detect_cuboids", "cs_batch_create", "cs_batch_max_boxes", "cs_batch_run", "cs_batch_submit", "cs_batch_collect",
    "cs_bgr_to_gray", "cs_edge_distance_maps", "cs_edge_distance_maps_multi", "cs_detect_cuboids_gray", "cs_batch_create_gray", "cs_batch_refill_gray", "cs_batch_refill_wait", "cs_batch_destroy", "cs_batch_last_timing", "cs_batch_debug_candidates", "cs_batch_debug_kept", "cs_batch_set_debug", "cs_batch_set_pipeline_chunks", "cs_detect_lines_gray", "cs_detect_lines_batch", "cs_detect_lines_last_timing", "cs_detect_lsd_gray", "cs_detect_lsd_batch", "cs_detect_lsd_last_timing",
]

_lib = None


def lib():
    """Load libcubeslam_hip.so; raises if it is missing (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libcubeslam_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # One HIP runtime per process: the PyTorch wheel ships its own libamdhip64, and whichever copy is loaded
        # first serves both (same SONAME).  Loading ours first leaves torch unable to see the GPU, so when torch is
        # installed it goes first.  (A C++ host, e.g. the reference's ROS nodes, has no torch and no such issue.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.cs_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def last_error():
    return lib().cs_last_error().decode()


def default_params(**kw):
    p = CsDetectParams()
    lib().cs_detect_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def box_rois(box5, img_w, img_h, sample_height=False):
    out = (CsRoi * 3)()
    b = (C.c_double * 5)(*[float(x) for x in box5])
    n = lib().cs_box_rois(b, int(img_w), int(img_h), int(bool(sample_height)), out)
    return [((out[k].left, out[k].top, out[k].width, out[k].height), out[k].down_expand) for k in range(n)]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def cuboid_to_dict(c):
    return dict(
        pos=np.array(c.pos[:]), scale=np.array(c.scale[:]), rotY=c.rotY, box_config_type=np.array(c.box_config_type[:]),
        box_corners_2d=np.array(c.box_corners_2d[:], dtype=np.int32).reshape(2, 8),
        box_corners_3d_world=np.array(c.box_corners_3d_world[:]).reshape(3, 8),
        rect_detect_2d=np.array(c.rect_detect_2d[:]), edge_distance_error=c.edge_distance_error,
        edge_angle_error=c.edge_angle_error, normalized_error=c.normalized_error, skew_ratio=c.skew_ratio,
        down_expand_height=c.down_expand_height, camera_roll_delta=c.camera_roll_delta, camera_pitch_delta=c.camera_pitch_delta)


class Detector:
    """cs_detector handle."""

    def __init__(self, params=None, device=0):
        self.params = params if params is not None else default_params()
        self.h = C.c_void_p()
        rc = lib().cs_detector_create(C.byref(self.params), int(device), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("cs_detector_create failed (%d): %s" % (rc, last_error()))

    def edge_distance_maps(self, gray, rois):
        """Canny(80, 200) + 3x3 L2 distance transform of each ROI (left, top, width, height) of a uint8 gray image, on the
        device (cs_edge_distance_maps).  Returns one float32 (h, w) array per ROI."""
        gray = np.ascontiguousarray(gray, np.uint8)
        H, W = gray.shape
        n = len(rois)
        rr = (CsRoi * max(1, n))()
        outs, ptrs = [], (C.POINTER(C.c_float) * max(1, n))()
        for k, (l, t, w, h) in enumerate(rois):
            rr[k].left, rr[k].top, rr[k].width, rr[k].height, rr[k].down_expand = int(l), int(t), int(w), int(h), 0
            o = np.zeros((int(h), int(w)), np.float32)
            outs.append(o)
            ptrs[k] = o.ctypes.data_as(C.POINTER(C.c_float))
        rc = lib().cs_edge_distance_maps(self.h, gray.ctypes.data_as(C.POINTER(C.c_ubyte)), W, H, rr, n, ptrs)
        if rc != 0:
            raise RuntimeError("cs_edge_distance_maps failed (%d): %s" % (rc, last_error()))
        return outs

    def edge_distance_maps_time(self, grays, rois, roi_image):
        """Device time (ms) of Canny + distance transform over ROIs of several equally sized images; maps are discarded."""
        grays = [np.ascontiguousarray(g, np.uint8) for g in grays]
        H, W = grays[0].shape
        gp = (C.POINTER(C.c_ubyte) * len(grays))(*[g.ctypes.data_as(C.POINTER(C.c_ubyte)) for g in grays])
        n = len(rois)
        rr = (CsRoi * max(1, n))()
        for k, (l, t, w, h) in enumerate(rois):
            rr[k].left, rr[k].top, rr[k].width, rr[k].height, rr[k].down_expand = int(l), int(t), int(w), int(h), 0
        ri = np.ascontiguousarray(roi_image, np.int32)
        ms = C.c_double()
        rc = lib().cs_edge_distance_maps_multi(self.h, gp, len(grays), W, H, rr, ri.ctypes.data_as(C.POINTER(C.c_int)), n, None, C.byref(ms))
        if rc != 0:
            raise RuntimeError("cs_edge_distance_maps_multi failed (%d): %s" % (rc, last_error()))
        return ms.value

    def detect_lines(self, gray, length_thres=15.0, cap=20000, use_lsd=False):
        """cs_detect_lines_gray / cs_detect_lsd_gray: line_lbd_detect::detect_filter_lines (EDLines, or LSD with the reference's
        use_LSD flag; one octave) -> (n, 4) float32 x1 y1 x2 y2."""
        gray = np.ascontiguousarray(gray, np.uint8)
        out = np.zeros((cap, 4), np.float32)
        n = C.c_int()
        rc = (lib().cs_detect_lsd_gray if use_lsd else lib().cs_detect_lines_gray)(self.h, gray.ctypes.data_as(C.POINTER(C.c_ubyte)), int(gray.shape[1]), int(gray.shape[0]), C.c_double(length_thres),
                                        out.ctypes.data_as(C.POINTER(C.c_float)), int(cap), C.byref(n))
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % ("cs_detect_lsd_gray" if use_lsd else "cs_detect_lines_gray", rc, last_error()))
        return out[:n.value].copy()

    def detect_lines_batch(self, grays, length_thres=15.0, cap=20000, use_lsd=False):
        """cs_detect_lines_batch / cs_detect_lsd_batch: images of one size -> list of (n_i, 4) float32 arrays.
        (The marshalling is kept light -- uninitialised result rows, plain addresses instead of a typed pointer object per image: it was
        3.4 ms per 64-image call, against 5 ms for the library's own work.)"""
        grays = [g if (isinstance(g, np.ndarray) and g.dtype == np.uint8 and g.flags.c_contiguous) else np.ascontiguousarray(g, np.uint8) for g in grays]
        n = len(grays)
        H, W = grays[0].shape
        if any(g.shape != (H, W) for g in grays):
            raise ValueError("detect_lines_batch: the images of a batch have one size")
        out = np.empty((n, cap, 4), np.float32)
        cnt = np.zeros(n, np.int32)
        gp = (C.c_void_p * n)(*[g.ctypes.data for g in grays])
        base, stride = out.ctypes.data, out.strides[0]
        op = (C.c_void_p * n)(*[base + i * stride for i in range(n)])
        rc = (lib().cs_detect_lsd_batch if use_lsd else lib().cs_detect_lines_batch)(self.h, gp, n, int(W), int(H), C.c_double(length_thres), op, int(cap), cnt.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % ("cs_detect_lsd_batch" if use_lsd else "cs_detect_lines_batch", rc, last_error()))
        return [out[i, :cnt[i]].copy() for i in range(n)]

    def lines_timing(self, use_lsd=False):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        rc = (lib().cs_detect_lsd_last_timing if use_lsd else lib().cs_detect_lines_last_timing)(self.h, C.byref(a), C.byref(b), C.byref(c))
        if rc != 0:
            raise RuntimeError("cs_detect_lines_last_timing failed (%d)" % rc)
        return {"device_ms": a.value, "host_ms": b.value, "total_ms": c.value}

    def frame_call(self, frame, gray=None):
        """A reusable zero-argument callable for cs_detect_cuboids (or cs_detect_cuboids_gray when `gray` is given) on one frame: the
        ctypes descriptor is built once, so repeated calls time the library, not the marshalling.  call() -> list of lists of dicts."""
        K = np.ascontiguousarray(frame["K"], np.float64).reshape(9)
        T = np.ascontiguousarray(frame["T_wc"], np.float64).reshape(16)
        boxes = np.ascontiguousarray(frame["boxes"], np.float64).reshape(-1, 5)
        lines = np.ascontiguousarray(frame["lines"], np.float64).reshape(-1, 4)
        nb = boxes.shape[0]
        arr = (C.POINTER(C.c_float) * max(1, 3 * nb))()
        keep = [K, T, boxes, lines]
        if gray is None:
            for i in range(nb):
                for k, m in enumerate(frame["maps"][i]):
                    m = np.ascontiguousarray(m, np.float32); keep.append(m)
                    arr[3 * i + k] = m.ctypes.data_as(C.POINTER(C.c_float))
        else:
            gray = np.ascontiguousarray(gray, np.uint8); keep.append(gray)
        d = CsFrameDesc()
        d.K, d.T_wc, d.img_w, d.img_h = _dp(K), _dp(T), int(frame["img_w"]), int(frame["img_h"])
        d.boxes, d.n_boxes, d.lines, d.n_lines, d.dist_maps = _dp(boxes), nb, _dp(lines), lines.shape[0], arr
        kmax = self.params.max_cuboid_num
        out = (CsCuboid * max(1, nb * kmax))()
        counts = np.zeros(max(1, nb), np.int32)
        cp = counts.ctypes.data_as(C.POINTER(C.c_int))
        L, h = lib(), self.h
        gp = gray.ctypes.data_as(C.POINTER(C.c_ubyte)) if gray is not None else None

        def call(parse=False):
            rc = L.cs_detect_cuboids(h, C.byref(d), out, cp) if gp is None else L.cs_detect_cuboids_gray(h, C.byref(d), gp, out, cp)
            if rc != 0:
                raise RuntimeError("cs_detect_cuboids failed (%d): %s" % (rc, last_error()))
            if parse:
                return [[cuboid_to_dict(out[i * kmax + k]) for k in range(int(counts[i]))] for i in range(nb)]
        call._keep = keep
        return call

    def detect_gray(self, frame, gray):
        """cs_detect_cuboids_gray: image in, cuboids out (the frame's 'maps' are not used)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        K = np.ascontiguousarray(frame["K"], np.float64).reshape(9)
        T = np.ascontiguousarray(frame["T_wc"], np.float64).reshape(16)
        boxes = np.ascontiguousarray(frame["boxes"], np.float64).reshape(-1, 5)
        lines = np.ascontiguousarray(frame["lines"], np.float64).reshape(-1, 4)
        d = CsFrameDesc()
        d.K, d.T_wc, d.img_w, d.img_h = _dp(K), _dp(T), int(frame["img_w"]), int(frame["img_h"])
        d.boxes, d.n_boxes, d.lines, d.n_lines, d.dist_maps = _dp(boxes), boxes.shape[0], _dp(lines), lines.shape[0], None
        kmax = self.params.max_cuboid_num
        out = (CsCuboid * max(1, boxes.shape[0] * kmax))()
        counts = np.zeros(max(1, boxes.shape[0]), np.int32)
        rc = lib().cs_detect_cuboids_gray(self.h, C.byref(d), gray.ctypes.data_as(C.POINTER(C.c_ubyte)), out, counts.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError("cs_detect_cuboids_gray failed (%d): %s" % (rc, last_error()))
        return [[cuboid_to_dict(out[i * kmax + k]) for k in range(int(counts[i]))] for i in range(boxes.shape[0])]

    def close(self):
        if self.h:
            lib().cs_detector_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """cs_batch handle over a list of frame dicts (cube_slam_wu_amd.synth.make_frame layout)."""

    def __init__(self, det: Detector, frames, debug=False, force_host_rank=False, force_host_setup=False, force_no_pipeline=False, pipeline_chunks=1, grays=None):
        self.det = det
        self.n_frames = len(frames)
        self._keep = []
        descs = (CsFrameDesc * max(1, self.n_frames))()
        for f, fr in enumerate(frames):
            K = np.ascontiguousarray(fr["K"], np.float64).reshape(9)
            T = np.ascontiguousarray(fr["T_wc"], np.float64).reshape(16)
            boxes = np.ascontiguousarray(fr["boxes"], np.float64).reshape(-1, 5)
            lines = np.ascontiguousarray(fr["lines"], np.float64).reshape(-1, 4)
            n = boxes.shape[0]
            arr = (C.POINTER(C.c_float) * max(1, 3 * n))()
            for i in range(n if grays is None else 0):
                for k, m in enumerate(fr["maps"][i]):
                    m = np.ascontiguousarray(m, np.float32)
                    self._keep.append(m)
                    arr[3 * i + k] = m.ctypes.data_as(C.POINTER(C.c_float))
            self._keep += [K, T, boxes, lines, arr]
            d = descs[f]
            d.K, d.T_wc, d.img_w, d.img_h = _dp(K), _dp(T), int(fr["img_w"]), int(fr["img_h"])
            d.boxes, d.n_boxes, d.lines, d.n_lines, d.dist_maps = _dp(boxes), n, _dp(lines), lines.shape[0], arr
        self.h = C.c_void_p()
        if grays is None:
            rc = lib().cs_batch_create(det.h, descs, self.n_frames, C.byref(self.h))
        else:   # image input: the maps are produced on the device from the gray images
            gs = [np.ascontiguousarray(g, np.uint8) for g in grays]
            self._gray_px = int(gs[0].size) if gs else 0
            gp = (C.POINTER(C.c_ubyte) * max(1, len(gs)))(*[g.ctypes.data_as(C.POINTER(C.c_ubyte)) for g in gs])
            rc = lib().cs_batch_create_gray(det.h, descs, gp, self.n_frames, C.byref(self.h))
        if rc != 0:
            raise RuntimeError("cs_batch_create failed (%d): %s" % (rc, last_error()))
        self._keep = []  # inputs were copied by the library
        self.max_boxes = lib().cs_batch_max_boxes(self.h)
        self.kmax = det.params.max_cuboid_num
        self._out = (CsCuboid * max(1, self.n_frames * max(1, self.max_boxes) * self.kmax))()
        self._counts = np.zeros(max(1, self.n_frames * max(1, self.max_boxes)), np.int32)
        if debug or force_host_rank or force_host_setup or force_no_pipeline:
            lib().cs_batch_set_debug(self.h, (1 if debug else 0) | (2 if force_host_rank else 0) | (4 if force_host_setup else 0) | (8 if force_no_pipeline else 0))

        if pipeline_chunks != 1:
            if lib().cs_batch_set_pipeline_chunks(self.h, int(pipeline_chunks)) != 0:
                raise RuntimeError("cs_batch_set_pipeline_chunks failed")

    def refill_gray(self, grays=None, base_ptr=None):
        """cs_batch_refill_gray: new gray images for the same frames (asynchronous).  grays: a list of uint8 arrays, or base_ptr: the address of
        n_frames images contiguous in (ideally pinned) host memory."""
        px = None
        if base_ptr is not None:
            t = lib().cs_batch_max_boxes(self.h)      # (any call: keeps the handle checked)
            px = self._gray_px
            ptrs = (C.POINTER(C.c_ubyte) * self.n_frames)(*[C.cast(base_ptr + f * px, C.POINTER(C.c_ubyte)) for f in range(self.n_frames)])
        else:
            self._refill_keep = [np.ascontiguousarray(g, np.uint8) for g in grays]
            ptrs = (C.POINTER(C.c_ubyte) * self.n_frames)(*[g.ctypes.data_as(C.POINTER(C.c_ubyte)) for g in self._refill_keep])
        rc = lib().cs_batch_refill_gray(self.det.h, self.h, ptrs)
        if rc != 0:
            raise RuntimeError("cs_batch_refill_gray failed (%d): %s" % (rc, last_error()))

    def refill_wait(self):
        rc = lib().cs_batch_refill_wait(self.h)
        if rc != 0:
            raise RuntimeError("cs_batch_refill_wait failed (%d): %s" % (rc, last_error()))

    def run(self):
        rc = lib().cs_batch_run(self.det.h, self.h, self._out, self._counts.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError("cs_batch_run failed (%d): %s" % (rc, last_error()))

    def submit(self):
        """First half of run(): pack + queue the sweep, do not wait (cs_batch_submit)."""
        rc = lib().cs_batch_submit(self.det.h, self.h, self._out, self._counts.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError("cs_batch_submit failed (%d): %s" % (rc, last_error()))

    def collect(self):
        """Second half of run(): wait for the sweep, write the records (cs_batch_collect)."""
        rc = lib().cs_batch_collect(self.det.h, self.h)
        if rc != 0:
            raise RuntimeError("cs_batch_collect failed (%d): %s" % (rc, last_error()))

    def cuboids(self, frame):
        res = []
        for i in range(self.max_boxes):
            cnt = int(self._counts[frame * self.max_boxes + i])
            res.append([cuboid_to_dict(self._out[(frame * self.max_boxes + i) * self.kmax + k]) for k in range(cnt)])
        return res

    def raw_out_bytes(self):
        return bytes(self._out)

    def counts_bytes(self):
        return self._counts.tobytes()

    def timing(self):
        t = CsDetectTiming()
        rc = lib().cs_batch_last_timing(self.h, C.byref(t))
        if rc != 0:
            raise RuntimeError("cs_batch_last_timing failed (%d)" % rc)
        return {n: getattr(t, n) for n, _ in CsDetectTiming._fields_}

    def debug_candidates(self, frame, box, k=0, with_corners=True):
        L = lib()
        n = L.cs_batch_debug_candidates(self.h, frame, box, k, 0, None, None)
        if n < 0:
            raise RuntimeError("cs_batch_debug_candidates failed (%d): %s" % (n, last_error()))
        rows = np.zeros((n, 9))
        corners = np.zeros((n, 16))
        if n:
            rc = L.cs_batch_debug_candidates(self.h, frame, box, k, n, _dp(rows), _dp(corners) if with_corners else None)
            if rc < 0:
                raise RuntimeError("cs_batch_debug_candidates failed (%d): %s" % (rc, last_error()))
        return rows, corners

    def debug_kept(self, frame, box, k=0):
        L = lib()
        n = L.cs_batch_debug_kept(self.h, frame, box, k, 0, None, None)
        if n < 0:
            raise RuntimeError("cs_batch_debug_kept failed (%d)" % n)
        ids = np.zeros(n, np.int32)
        sc = np.zeros(n)
        if n:
            L.cs_batch_debug_kept(self.h, frame, box, k, n, ids.ctypes.data_as(C.POINTER(C.c_int)), _dp(sc))
        return ids, sc

    def close(self):
        if self.h:
            lib().cs_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ Path B: bundle adjustment ----------
class CsBaTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("errors_ms", "linearize_ms", "reduce_ms", "schur_ms", "factor_ms", "backsub_ms", "update_ms", "total_ms")] + \
               [("n_linearizations", C.c_longlong), ("n_solves", C.c_longlong), ("linearize_bytes", C.c_longlong), ("schur_entries", C.c_longlong)]


DECLARED_SYMBOLS += [
    "cs_ba_create", "cs_ba_destroy", "cs_ba_set_vertices", "cs_ba_set_estimates", "cs_ba_set_edges_proj", "cs_ba_set_edges_cuboid", "cs_ba_set_edges_cuboid_proj", "cs_ba_set_edges_odom",
    "cs_ba_compute_errors", "cs_ba_build_system", "cs_ba_solve", "cs_ba_update", "cs_ba_push", "cs_ba_pop", "cs_ba_optimize",
    "cs_ba_get_state", "cs_ba_sizes", "cs_ba_solver_layout", "cs_ba_get_system", "cs_ba_last_timing", "cs_ba_set_shard", "cs_ba_optimize_sharded",
    "cs_ba_shard_landmark_owners", "cs_ba_get_landmark_owners", "cs_ba_shard_info", "cs_ba_shard_timing", "cs_ba_append_vertices", "cs_ba_append_edges_proj", "cs_ba_append_edges_cuboid", "cs_ba_append_edges_cuboid_proj", "cs_ba_append_edges_odom", "cs_ba_get_vertex_hessians", "cs_ba_schur_layout", "cs_ba_structure_digest", "cs_ba_reduced_size", "cs_ba_solver_path", "cs_ba_band_order", "cs_ba_comm_unique_id", "cs_ba_comm_init", "cs_ba_set_robust_kernels",
    "cs_ba_set_external_edges", "cs_ba_set_external_terms", "cs_ba_set_external_chi2", "cs_ba_set_external_callback", "cs_ba_check_finite", "cs_ba_dump", "cs_ba_load", "cs_ba_get_reduced_system", "cs_ba_set_stage_timing", "cs_ba_set_lm_params", "cs_ba_pose_marginals",
]


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def _f64(a, shape):
    return np.ascontiguousarray(np.asarray(a, np.float64).reshape(shape))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, np.int32).ravel())


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, last_error()))


class BaProblem:
    """cs_ba handle; the method names mirror the g2o calls each one replaces."""

    def __init__(self, cams, cam_fixed, cuboids=None, cub_fixed=None, points=None, pt_fixed=None, cuboids_first=False, device=0):
        L = lib()
        self.h = C.c_void_p()
        _chk(L.cs_ba_create(int(device), C.byref(self.h)), "cs_ba_create")
        cams = _f64(cams, (-1, 7)); self.nc = len(cams)
        cub = _f64(cuboids if cuboids is not None else np.zeros((0, 10)), (-1, 10)); self.no = len(cub)
        pts = _f64(points if points is not None else np.zeros((0, 3)), (-1, 3)); self.np_ = len(pts)
        cf = _i32(cam_fixed); of = _i32(cub_fixed if cub_fixed is not None else np.zeros(self.no)); pf = _i32(pt_fixed if pt_fixed is not None else np.zeros(self.np_))
        _chk(L.cs_ba_set_vertices(self.h, _dp(cams), _ip(cf), self.nc, _dp(cub), _ip(of), self.no, _dp(pts), _ip(pf), self.np_, int(cuboids_first)), "cs_ba_set_vertices")
        self.n_proj = self.n_cub = self.n_odom = 0

    def set_edges_proj(self, pt, cam, uv, info4, intr4, huber=None):
        pt, cam = _i32(pt), _i32(cam); self.n_proj = len(pt)
        hb = _f64(huber, (-1,)) if huber is not None else None
        _chk(lib().cs_ba_set_edges_proj(self.h, self.n_proj, _ip(pt), _ip(cam), _dp(_f64(uv, (-1, 2))), _dp(_f64(info4, (-1, 4))), _dp(_f64(intr4, (-1, 4))),
                                        _dp(hb) if hb is not None else None), "cs_ba_set_edges_proj")

    def set_edges_cuboid(self, cam, cub, meas10, info81):
        cam, cub = _i32(cam), _i32(cub); self.n_cub = len(cam)
        _chk(lib().cs_ba_set_edges_cuboid(self.h, self.n_cub, _ip(cam), _ip(cub), _dp(_f64(meas10, (-1, 10))), _dp(_f64(info81, (-1, 81)))), "cs_ba_set_edges_cuboid")

    def set_edges_cuboid_proj(self, cam, cub, meas4, info16, K9):
        """EdgeSE3CuboidProj: bounding-box (cx, cy, w, h) error of the projected cuboid."""
        cam, cub = _i32(cam), _i32(cub); self.n_cproj = len(cam)
        _chk(lib().cs_ba_set_edges_cuboid_proj(self.h, self.n_cproj, _ip(cam), _ip(cub), _dp(_f64(meas4, (-1, 4))), _dp(_f64(info16, (-1, 16))), _dp(_f64(K9, (-1, 9)))),
             "cs_ba_set_edges_cuboid_proj")

    def set_edges_odom(self, ci, cj, meas7, info36):
        ci, cj = _i32(ci), _i32(cj); self.n_odom = len(ci)
        _chk(lib().cs_ba_set_edges_odom(self.h, self.n_odom, _ip(ci), _ip(cj), _dp(_f64(meas7, (-1, 7))), _dp(_f64(info36, (-1, 36)))), "cs_ba_set_edges_odom")

    def set_robust_kernels(self, edge_class, kind, delta):
        """cs_ba_set_robust_kernels: edge_class 0 projection / 1 EdgeSE3Cuboid / 2 EdgeSE3CuboidProj / 3 EdgeSE3Expmap; kind per edge
        (RK_* below), delta = RobustKernel::delta().  kind None removes the class's kernels."""
        if kind is None:        # (the library counts the class's edges itself: no shadow count here, a handle from Problem.load() has none)
            _chk(lib().cs_ba_set_robust_kernels(self.h, int(edge_class), 0, None, None), "cs_ba_set_robust_kernels")
            return
        kind, delta = _i32(kind), _f64(delta, (-1,))
        _chk(lib().cs_ba_set_robust_kernels(self.h, int(edge_class), len(kind), _ip(kind), _dp(delta)), "cs_ba_set_robust_kernels")

    # ---- external (host-evaluated) edges
    def set_external_edges(self, class_i, idx_i, class_j, idx_j):
        ci, ii, cj, ij = _i32(class_i), _i32(idx_i), _i32(class_j), _i32(idx_j)
        self.n_ext = len(ci)
        _chk(lib().cs_ba_set_external_edges(self.h, len(ci), _ip(ci), _ip(ii), _ip(cj), _ip(ij)), "cs_ba_set_external_edges")

    def set_external_terms(self, cam36=None, cam6=None, cub81=None, cub9=None, pt9=None, pt3=None, Hij81=None, chi2=0.0):
        a = [(_f64(x, (-1,)) if x is not None else None) for x in (cam36, cam6, cub81, cub9, pt9, pt3, Hij81)]
        _chk(lib().cs_ba_set_external_terms(self.h, *[(_dp(x) if x is not None else None) for x in a], C.c_double(chi2)), "cs_ba_set_external_terms")

    def set_external_chi2(self, chi2):
        _chk(lib().cs_ba_set_external_chi2(self.h, C.c_double(chi2)), "cs_ba_set_external_chi2")

    def set_external_callback(self, fn):
        """fn(want_system) -> None: re-evaluates the external edges at self.state() (see cs_ba_set_external_callback)."""
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)

        def _cb(ctx, ba, want_system):
            try:
                fn(int(want_system))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._ext_cb = CB(_cb) if fn is not None else None
        _chk(lib().cs_ba_set_external_callback(self.h, self._ext_cb, None), "cs_ba_set_external_callback")

    # ---- growing graphs: new vertices / edges behind the existing ones, device-side estimates kept
    def append_vertices(self, cams=None, cam_fixed=None, cuboids=None, cub_fixed=None, points=None, pt_fixed=None):
        c = _f64(cams if cams is not None else np.zeros((0, 7)), (-1, 7)); o = _f64(cuboids if cuboids is not None else np.zeros((0, 10)), (-1, 10))
        q = _f64(points if points is not None else np.zeros((0, 3)), (-1, 3))
        cf = _i32(cam_fixed if cam_fixed is not None else np.zeros(len(c))); of = _i32(cub_fixed if cub_fixed is not None else np.zeros(len(o))); pf = _i32(pt_fixed if pt_fixed is not None else np.zeros(len(q)))
        _chk(lib().cs_ba_append_vertices(self.h, _dp(c), _ip(cf), len(c), _dp(o), _ip(of), len(o), _dp(q), _ip(pf), len(q)), "cs_ba_append_vertices")
        self.nc += len(c); self.no += len(o); self.np_ += len(q)

    def append_edges_proj(self, pt, cam, uv, info4, intr4, huber=None):
        pt, cam = _i32(pt), _i32(cam)
        hb = _f64(huber, (-1,)) if huber is not None else None
        _chk(lib().cs_ba_append_edges_proj(self.h, len(pt), _ip(pt), _ip(cam), _dp(_f64(uv, (-1, 2))), _dp(_f64(info4, (-1, 4))), _dp(_f64(intr4, (-1, 4))), _dp(hb) if hb is not None else None), "cs_ba_append_edges_proj")
        self.n_proj = getattr(self, "n_proj", 0) + len(pt)

    def append_edges_cuboid(self, cam, cub, meas10, info81):
        cam, cub = _i32(cam), _i32(cub)
        _chk(lib().cs_ba_append_edges_cuboid(self.h, len(cam), _ip(cam), _ip(cub), _dp(_f64(meas10, (-1, 10))), _dp(_f64(info81, (-1, 81)))), "cs_ba_append_edges_cuboid")

    def append_edges_odom(self, ci, cj, meas7, info36):
        ci, cj = _i32(ci), _i32(cj)
        _chk(lib().cs_ba_append_edges_odom(self.h, len(ci), _ip(ci), _ip(cj), _dp(_f64(meas7, (-1, 7))), _dp(_f64(info36, (-1, 36)))), "cs_ba_append_edges_odom")

    def compute_errors(self):
        chi = C.c_double()
        _chk(lib().cs_ba_compute_errors(self.h, C.byref(chi)), "cs_ba_compute_errors")
        return chi.value

    def sizes(self):
        a, b = C.c_int(), C.c_int()
        _chk(lib().cs_ba_sizes(self.h, C.byref(a), C.byref(b)), "cs_ba_sizes")
        return a.value, b.value

    def build_system(self, dense_hpp=True):
        _chk(lib().cs_ba_build_system(self.h), "cs_ba_build_system")
        n, nl = self.sizes()
        Hpp, Hll, Hpl, b = (np.zeros((n, n)) if dense_hpp else None), np.zeros((nl // 3, 9)), np.zeros((self.n_proj, 18)), np.zeros(n + nl)
        _chk(lib().cs_ba_get_system(self.h, _dp(Hpp) if dense_hpp else None, _dp(Hll), _dp(Hpl), _dp(b), None), "cs_ba_get_system")
        return Hpp, Hll, Hpl, b

    def pose_marginals(self, pairs):
        """cs_ba_pose_marginals (Solver::computeMarginals): pairs = [((class_i, idx_i), (class_j, idx_j)), ...] with class 0 = camera (6), 1 = cuboid (9);
        -> (list of d_i x d_j blocks of inv(H_pp), positive_definite)."""
        n = len(pairs)
        ci = np.array([p[0][0] for p in pairs], np.int32); ii = np.array([p[0][1] for p in pairs], np.int32)
        cj = np.array([p[1][0] for p in pairs], np.int32); ij = np.array([p[1][1] for p in pairs], np.int32)
        dims = [({0: 6, 1: 9}.get(int(a), 3), {0: 6, 1: 9}.get(int(b), 3)) for a, b in zip(ci, cj)]      # (a point: the library refuses it)
        out = np.zeros(max(1, sum(a * b for a, b in dims)))
        pd = C.c_int(1)
        _chk(lib().cs_ba_pose_marginals(self.h, n, _ip(ci), _ip(ii), _ip(cj), _ip(ij), _dp(out), C.byref(pd)), "cs_ba_pose_marginals")
        blocks, o = [], 0
        for a, b in dims:
            blocks.append(out[o:o + a * b].reshape(a, b).copy()); o += a * b
        return blocks, bool(pd.value)

    def reduced_system(self, lam):
        """(S dense n_red x n_red, rhs, camera columns, cuboid columns) in solver order -- cs_ba_get_reduced_system."""
        n = self.reduced_size()[0]
        S, rhs, cc, oc = np.zeros((n, n)), np.zeros(n), np.zeros(max(1, self.nc), np.int32), np.zeros(max(1, self.no), np.int32)
        _chk(lib().cs_ba_get_reduced_system(self.h, C.c_double(lam), _dp(S), _dp(rhs), _ip(cc), _ip(oc)), "cs_ba_get_reduced_system")
        return S, rhs, cc[:self.nc], oc[:self.no]

    def vertex_hessians(self):
        """A_ii of every vertex (what g2o maps into BaseVertex::_hessian): (nc, 6, 6), (no, 9, 9), (np, 3, 3)."""
        hc, ho, hp = np.zeros((self.nc, 36)), np.zeros((self.no, 81)), np.zeros((self.np_, 9))
        _chk(lib().cs_ba_get_vertex_hessians(self.h, _dp(hc), _dp(ho), _dp(hp)), "cs_ba_get_vertex_hessians")
        return hc.reshape(-1, 6, 6), ho.reshape(-1, 9, 9), hp.reshape(-1, 3, 3)

    def set_estimates(self, cams=None, cuboids=None, points=None):
        """New estimates for the same graph (after a g2o-side update()/pop()): no structure rebuild."""
        c = _f64(cams, (-1, 7)) if cams is not None else None
        o = _f64(cuboids, (-1, 10)) if cuboids is not None else None
        q = _f64(points, (-1, 3)) if points is not None else None
        _chk(lib().cs_ba_set_estimates(self.h, _dp(c) if c is not None else None, _dp(o) if o is not None else None, _dp(q) if q is not None else None), "cs_ba_set_estimates")

    def update(self):
        _chk(lib().cs_ba_update(self.h), "cs_ba_update")

    def push(self):
        _chk(lib().cs_ba_push(self.h), "cs_ba_push")

    def pop(self):
        _chk(lib().cs_ba_pop(self.h), "cs_ba_pop")

    def system_vectors(self):
        """(b, x) of the last build / solve in g2o's order [poses..., landmarks...] (Solver::b(), Solver::x())."""
        n, nl = self.sizes()
        b, x = np.zeros(n + nl), np.zeros(n + nl)
        _chk(lib().cs_ba_get_system(self.h, None, None, None, _dp(b), _dp(x)), "cs_ba_get_system")
        return b, x

    def reduced_size(self):
        """(unknowns of the factorised system, cuboids eliminated?) -- cs_ba_reduced_size."""
        a, b = C.c_int(), C.c_int()
        _chk(lib().cs_ba_reduced_size(self.h, C.byref(a), C.byref(b)), "cs_ba_reduced_size")
        return a.value, bool(b.value)

    def band_order(self):
        """(block cyclic reduction?, levels) of a banded reduced system -- cs_ba_band_order."""
        a, b = C.c_int(0), C.c_int(0)
        _chk(lib().cs_ba_band_order(self.h, C.byref(a), C.byref(b)), "cs_ba_band_order")
        return bool(a.value), b.value

    def solver_path(self, detail=False):
        """'band' / 'sparse' / 'dense': the factorisation the reduced system takes (cs_ba_solver_path); detail: (path, bandwidth, sparse fill)."""
        a, b, f = C.c_int(), C.c_int(), C.c_double()
        _chk(lib().cs_ba_solver_path(self.h, C.byref(a), C.byref(b), C.byref(f)), "cs_ba_solver_path")
        name = {0: "dense", 1: "band", 2: "sparse"}[a.value]
        return (name, b.value, f.value) if detail else name

    def structure_digest(self):
        """One 64-bit hash per index table of the structure phase (cs_ba_structure_digest)."""
        n = C.c_int(0)
        _chk(lib().cs_ba_structure_digest(self.h, None, 0, C.byref(n)), "cs_ba_structure_digest")
        out = (C.c_ulonglong * n.value)()
        _chk(lib().cs_ba_structure_digest(self.h, out, n.value, C.byref(n)), "cs_ba_structure_digest")
        return [int(v) for v in out]

    def schur_layout(self):
        """(fused, segments, partial blocks, destination blocks) of the Schur-complement build (cs_ba_schur_layout)."""
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _chk(lib().cs_ba_schur_layout(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "cs_ba_schur_layout")
        return bool(a.value), b.value, c.value, d.value

    def solver_layout(self):
        a, b = C.c_int(), C.c_int()
        _chk(lib().cs_ba_solver_layout(self.h, C.byref(a), C.byref(b)), "cs_ba_solver_layout")
        return a.value, b.value

    def solve(self, lam):
        pd = C.c_int()
        _chk(lib().cs_ba_solve(self.h, C.c_double(lam), C.byref(pd)), "cs_ba_solve")
        n, nl = self.sizes()
        x = np.zeros(n + nl)
        if pd.value:
            _chk(lib().cs_ba_get_system(self.h, None, None, None, None, _dp(x)), "cs_ba_get_system")
        return bool(pd.value), x

    def optimize(self, iters, cap=64):
        done = C.c_int()
        self._chi, self._lam, self._tr = np.zeros(cap), np.zeros(cap), np.zeros(cap, np.int32)
        _chk(lib().cs_ba_optimize(self.h, int(iters), C.byref(done), _dp(self._chi), _dp(self._lam), _ip(self._tr), cap), "cs_ba_optimize")
        self._done = done.value
        return done.value

    def history(self):
        n = self._done
        return self._chi[:n], self._lam[:n], self._tr[:n]

    # ---- sharded BA -------------------------------------------------------------------------------
    def set_shard(self, rank, n_ranks):
        _chk(lib().cs_ba_set_shard(self.h, int(rank), int(n_ranks)), "cs_ba_set_shard")
        self.shard = (int(rank), int(n_ranks))

    def comm_init(self, rank, n_ranks, unique_id):
        """RCCL communicator for the sharded BA (cs_ba_comm_init): unique_id = the 128 bytes rank 0 got from comm_unique_id()."""
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _chk(lib().cs_ba_comm_init(self.h, int(rank), int(n_ranks), buf), "cs_ba_comm_init")
        self.shard = (int(rank), int(n_ranks))

    def landmark_owners(self):
        """Rank owning each landmark under the rule in force for this handle (cs_ba_get_landmark_owners)."""
        out = np.zeros(max(1, self.np_), np.int32)
        _chk(lib().cs_ba_get_landmark_owners(self.h, _ip(out)), "cs_ba_get_landmark_owners")
        return out[:self.np_]

    def shard_info(self):
        """dict(sep_mode, n_sep, w_max, bytes_per_trial, bytes_per_trial_allreduce, interior_n) -- cs_ba_shard_info."""
        a, b, c, f = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        d, e = C.c_longlong(), C.c_longlong()
        _chk(lib().cs_ba_shard_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e), C.byref(f)), "cs_ba_shard_info")
        return dict(sep_mode=a.value, n_sep=b.value, w_max=c.value, bytes_per_trial=d.value, bytes_per_trial_allreduce=e.value, interior_n=f.value)

    def shard_timing(self):
        """Accumulated ms of the separator-mode solve stages (cs_ba_shard_timing)."""
        out = (C.c_double * 5)()
        _chk(lib().cs_ba_shard_timing(self.h, out), "cs_ba_shard_timing")
        return dict(zip(("interior_factor_ms", "separator_message_ms", "gather_ms", "separator_solve_ms", "interior_backsolve_ms"), list(out)))

    def optimize_sharded(self, iters, allreduce=None, cap=64):
        """allreduce(ptr, n_doubles, on_device, op) -> 0: in-place all-reduce (op 0 = SUM, 1 = MAX) of n doubles; None = the
        library's own RCCL communicator (comm_init)."""
        if allreduce is None:
            done = C.c_int()
            self._chi, self._lam, self._tr = np.zeros(cap), np.zeros(cap), np.zeros(cap, np.int32)
            _chk(lib().cs_ba_optimize_sharded(self.h, int(iters), None, None, C.byref(done), _dp(self._chi), _dp(self._lam), _ip(self._tr), cap), "cs_ba_optimize_sharded")
            self._done = done.value
            return done.value
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)

        def _cb(ctx, data, n, on_device, op):
            try:
                return int(allreduce(data, int(n), int(on_device), int(op)) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = CB(_cb)
        done = C.c_int()
        self._chi, self._lam, self._tr = np.zeros(cap), np.zeros(cap), np.zeros(cap, np.int32)
        _chk(lib().cs_ba_optimize_sharded(self.h, int(iters), cb, None, C.byref(done), _dp(self._chi), _dp(self._lam), _ip(self._tr), cap), "cs_ba_optimize_sharded")
        self._done = done.value
        return done.value

    def check_finite(self):
        """(number of non-finite values, report) -- cs_ba_check_finite."""
        n, buf = C.c_int(), C.create_string_buffer(4096)
        _chk(lib().cs_ba_check_finite(self.h, C.byref(n), buf, 4096), "cs_ba_check_finite")
        return n.value, buf.value.decode()

    def dump(self, path):
        _chk(lib().cs_ba_dump(self.h, str(path).encode()), "cs_ba_dump")

    @classmethod
    def load(cls, path, sizes, device=0):
        """cs_ba_load; sizes = (n_cams, n_cuboids, n_points, n_proj) of the dumped problem (the Python wrapper sizes its host arrays with them)."""
        P = cls.__new__(cls)
        P.h = C.c_void_p()
        _chk(lib().cs_ba_load(str(path).encode(), int(device), C.byref(P.h)), "cs_ba_load")
        P.nc, P.no, P.np_, P.n_proj = [int(x) for x in sizes]
        P.n_cub = P.n_odom = 0
        return P

    def state(self):
        cams, cubs, pts = np.zeros((self.nc, 7)), np.zeros((self.no, 10)), np.zeros((self.np_, 3))
        _chk(lib().cs_ba_get_state(self.h, _dp(cams), _dp(cubs), _dp(pts)), "cs_ba_get_state")
        return cams, cubs, pts

    def stage_timing(self, on=True):
        """g2o's setComputeBatchStatistics: turn the per-stage split of timing() on (off by default: its phase marks cost ~6 us each on the stream)."""
        _chk(lib().cs_ba_set_stage_timing(self.h, 1 if on else 0), "cs_ba_set_stage_timing")
        return self

    def set_lm_params(self, user_lambda_init=0.0, max_trials_after_failure=10):
        """OptimizationAlgorithmLevenberg::setUserLambdaInit / setMaxTrialsAfterFailure (optimization_algorithm_levenberg.cpp:191-199)."""
        lib().cs_ba_set_lm_params.argtypes = [C.c_void_p, C.c_double, C.c_int]
        _chk(lib().cs_ba_set_lm_params(self.h, float(user_lambda_init), int(max_trials_after_failure)), "cs_ba_set_lm_params")
        return self

    def timing(self):
        t = CsBaTiming()
        _chk(lib().cs_ba_last_timing(self.h, C.byref(t)), "cs_ba_last_timing")
        return {n: getattr(t, n) for n, _ in CsBaTiming._fields_}

    def close(self):
        if self.h:
            lib().cs_ba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


RK_NONE, RK_HUBER, RK_PSEUDO_HUBER, RK_CAUCHY, RK_SATURATED, RK_DCS, RK_TUKEY = range(7)      # enum cs_robust_kernel
EDGE_PROJ, EDGE_CUBOID, EDGE_CUBOID_PROJ, EDGE_ODOM = range(4)                                 # enum cs_edge_class


def ba_from_dict(pr, device=0, cuboids_first=False):
    """BaProblem from a cube_slam_wu_amd.synth_ba.make_problem() dict."""
    P = BaProblem(pr["cams"], pr["cam_fixed"], pr["cuboids"], pr["cub_fixed"], pr["points"], pr["pt_fixed"], cuboids_first=cuboids_first, device=device)
    if len(pr["e_pt"]):
        P.set_edges_proj(pr["e_pt"], pr["e_cam"], pr["e_uv"], pr["e_info"], pr["e_intr"], pr["e_huber"])
    if len(pr["ce_cam"]):
        P.set_edges_cuboid(pr["ce_cam"], pr["ce_cub"], pr["ce_meas"], pr["ce_info"])
    if len(pr.get("pe_cam", [])):
        P.set_edges_cuboid_proj(pr["pe_cam"], pr["pe_cub"], pr["pe_meas"], pr["pe_info"], pr["pe_K"])
    if len(pr["oe_i"]):
        P.set_edges_odom(pr["oe_i"], pr["oe_j"], pr["oe_meas"], pr["oe_info"])
    for cls, (kind, delta) in pr.get("robust", {}).items():       # optional: {edge class: (kinds, deltas)}
        P.set_robust_kernels(cls, kind, delta)
    return P


def comm_unique_id():
    """128-byte RCCL unique id (ncclGetUniqueId); call on rank 0 and broadcast."""
    buf = (C.c_ubyte * 128)()
    _chk(lib().cs_ba_comm_unique_id(buf), "cs_ba_comm_unique_id")
    return bytes(buf)


def landmark_owners(n_ranks, n_cams, n_points, e_pt, e_cam):
    """Host-only: rank owning each landmark in the sharded BA (cs_ba_shard_landmark_owners)."""
    e_pt, e_cam = _i32(e_pt), _i32(e_cam)
    out = np.zeros(int(n_points), np.int32)
    _chk(lib().cs_ba_shard_landmark_owners(int(n_ranks), int(n_cams), int(n_points), len(e_pt), _ip(e_pt), _ip(e_cam), _ip(out)), "cs_ba_shard_landmark_owners")
    return out


class DeviceDoubles:
    """__cuda_array_interface__ view of n doubles at a raw device pointer (for torch.as_tensor(..., device='cuda'))."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def torch_allreduce(dist, device):
    """all-reduce callback for BaProblem.optimize_sharded backed by torch.distributed (NCCL = RCCL on ROCm)."""
    import torch

    def fn(ptr, n, on_device, op):
        rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
        if on_device:
            t = torch.as_tensor(DeviceDoubles(ptr, n), device=device)
            if dist.get_backend() == "gloo":      # CPU collective: stage through the host
                h = t.cpu()
                dist.all_reduce(h, op=rop)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=rop)
            torch.cuda.synchronize(device)
        else:
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,))
            t = torch.from_numpy(a)
            if dist.get_backend() == "gloo":
                dist.all_reduce(t, op=rop)
            else:
                d = t.to(device)
                dist.all_reduce(d, op=rop)
                t.copy_(d.cpu())
        return 0
    return fn


def cam_rank(cam, n_cams, n_ranks):
    """Rank owning a camera in the sharded BA: contiguous subsequences [r*Nc/R, (r+1)*Nc/R) (ba_host.cpp cam_rank)."""
    return (np.asarray(cam, np.int64) * int(n_ranks)) // max(1, int(n_cams))


def shard_frames(n_frames, rank, world):
    """Path A multi-GPU: frames are independent units, dealt round-robin; returns this rank's frame indices."""
    return list(range(int(rank), int(n_frames), int(world)))
