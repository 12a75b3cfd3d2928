"""ctypes binding of oracle/libekf_oracle.so -- TEST INFRASTRUCTURE ONLY (the checker, never the product path)."""
import ctypes as C
import os
import subprocess

import numpy as np

from openekfmonoslam_amd.ekftypes import (DESC_BYTES, KEYPOINT_DTYPE, MATCH_DTYPE, PREDICTION_DTYPE, EkfCamera,
                                       EkfParams, EkfStepInfo)

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
LITERAL, ALGORITHMIC = 0, 1

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libekf_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, i32, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.POINTER(EkfCamera), C.POINTER(EkfParams), i32]
        L.orc_destroy.argtypes = [vp]
        L.orc_set_descriptor_format.argtypes = [vp, i32]
        L.orc_descriptor_bytes.argtypes = [vp]
        L.orc_reset.argtypes = [vp]
        L.orc_add_feature.argtypes = [vp, vp, vp]
        L.orc_set_state.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        L.orc_state_dim.argtypes = [vp]
        L.orc_num_features.argtypes = [vp]
        for name, rt in [("orc_x13", dp), ("orc_rotation", dp), ("orc_feature_pos", dp), ("orc_P", dp),
                         ("orc_feature_type", C.POINTER(C.c_int32)), ("orc_feature_covpos", C.POINTER(C.c_int32)),
                         ("orc_feature_desc", C.POINTER(C.c_uint8)),
                         ("orc_feature_times_predicted", C.POINTER(C.c_uint32)),
                         ("orc_feature_times_matched", C.POINTER(C.c_uint32))]:
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [vp]
        L.orc_remove_features.argtypes = [vp, vp, i32]
        L.orc_remove_bad_features.argtypes = [vp]
        L.orc_linearity_index.restype = C.c_double
        L.orc_linearity_index.argtypes = [vp, i32]
        L.orc_convert_to_depth.argtypes = [vp, i32]
        L.orc_convert_inverse_depth_to_depth.argtypes = [vp]
        L.orc_predict.argtypes = [vp, vp, vp]
        L.orc_predict_measurement_state.argtypes = [vp, vp, vp, vp, vp, i32, vp]
        L.orc_predict_measurements.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        L.orc_ellipse.argtypes = [vp, vp, vp]
        L.orc_point_in_ellipse.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, i32, i32, C.c_double]
        L.orc_match.argtypes = [vp, vp, i32, vp, vp, i32, vp]
        L.orc_ransac.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
        L.orc_update.argtypes = [vp, vp, vp, vp, vp, i32, i32]
        L.orc_rescue.argtypes = [vp, vp, vp, i32, vp]
        L.orc_step.argtypes = [vp, vp, vp, i32, i32, vp]
        L.orc_image_set.argtypes = [vp, i32, i32, i32, i32]
        L.orc_image_level.restype = C.POINTER(C.c_uint8)
        L.orc_image_level.argtypes = [i32, C.POINTER(i32), C.POINTER(i32)]
        L.orc_capture_templates.argtypes = [vp, vp, vp, i32]
        L.orc_templates.restype = C.POINTER(C.c_uint8)
        L.orc_match_ncc.argtypes = [vp, vp, i32, vp]
        L.orc_step_image.argtypes = [vp, i32, vp]
        L.orc_detect_new_features.argtypes = [vp, i32, i32, i32, C.c_double, C.c_double, vp]
        L.orc_time_literal_rows.restype = C.c_double
        L.orc_time_literal_rows.argtypes = [i32, i32, i32, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """Thin object wrapper; array outputs are numpy copies."""

    def __init__(self, cam, par, max_features, descriptor_cols_f32=0):
        self.L = lib()
        self.cam, self.par = cam, par
        self.h = self.L.orc_create(C.byref(cam), C.byref(par), int(max_features))
        assert self.h
        self.cap = int(max_features)
        if descriptor_cols_f32:  # CV_32F descriptors, L2 distance (Matching.cpp:60-73)
            assert self.L.orc_set_descriptor_format(self.h, 1 | (int(descriptor_cols_f32) << 8)) == 0
        self.desc_bytes = self.L.orc_descriptor_bytes(self.h)
        self.desc_dtype = np.float32 if descriptor_cols_f32 else np.uint8

    def _desc(self, desc):
        if desc is None:
            return None
        return np.ascontiguousarray(desc, dtype=self.desc_dtype).view(np.uint8).reshape(-1, self.desc_bytes)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    # -- state access
    @property
    def n(self):
        return self.L.orc_state_dim(self.h)

    @property
    def N(self):
        return self.L.orc_num_features(self.h)

    def x13(self):
        return np.ctypeslib.as_array(self.L.orc_x13(self.h), (13,)).copy()

    def rotation(self):
        return np.ctypeslib.as_array(self.L.orc_rotation(self.h), (9,)).copy().reshape(3, 3)

    def P(self):
        n = self.n
        return np.ctypeslib.as_array(self.L.orc_P(self.h), (n * n,)).copy().reshape(n, n)

    def feature_pos(self):
        return np.ctypeslib.as_array(self.L.orc_feature_pos(self.h), (self.N * 6,)).copy().reshape(-1, 6)

    def map_features(self):
        N = self.N
        d = np.ctypeslib.as_array(self.L.orc_feature_desc(self.h), (N * self.desc_bytes,)).copy().reshape(N, self.desc_bytes).view(self.desc_dtype)
        tp = np.ctypeslib.as_array(self.L.orc_feature_times_predicted(self.h), (N,)).copy()
        tm = np.ctypeslib.as_array(self.L.orc_feature_times_matched(self.h), (N,)).copy()
        return d, tp, tm

    def feature_type(self):
        return np.ctypeslib.as_array(self.L.orc_feature_type(self.h), (self.N,)).copy()

    def remove_features(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        assert self.L.orc_remove_features(self.h, _p(idx), len(idx)) == 0

    def remove_bad_features(self):
        return self.L.orc_remove_bad_features(self.h)

    def linearity_index(self, fi):
        return self.L.orc_linearity_index(self.h, int(fi))

    def convert_to_depth(self, fi):
        assert self.L.orc_convert_to_depth(self.h, int(fi)) == 0

    def convert_inverse_depth_to_depth(self):
        return self.L.orc_convert_inverse_depth_to_depth(self.h)

    def feature_covpos(self):
        return np.ctypeslib.as_array(self.L.orc_feature_covpos(self.h), (self.N,)).copy()

    def reset(self):
        self.L.orc_reset(self.h)

    def add_feature(self, uv, desc=None):
        uv = np.ascontiguousarray(uv, dtype=np.float64)
        d = self._desc(desc)
        return self.L.orc_add_feature(self.h, _p(uv), _p(d))

    def set_state(self, x13, feature_pos, feature_type, desc, P):
        x13 = np.ascontiguousarray(x13, dtype=np.float64)
        fp = np.ascontiguousarray(feature_pos, dtype=np.float64).reshape(-1, 6)
        ft = None if feature_type is None else np.ascontiguousarray(feature_type, dtype=np.int32)
        d = self._desc(desc)
        Pm = np.ascontiguousarray(P, dtype=np.float64)
        rc = self.L.orc_set_state(self.h, _p(x13), len(fp), _p(fp), _p(ft), _p(d), _p(Pm))
        assert rc == 0, rc

    # -- stages
    def predict(self, want_F=False):
        if want_F:
            F = np.zeros((13, 13))
            Q = np.zeros((13, 13))
            self.L.orc_predict(self.h, _p(F), _p(Q))
            return F, Q
        self.L.orc_predict(self.h, None, None)

    def predict_measurements(self, feat_idx=None, want_HP=False):
        N = self.N
        idx = None if feat_idx is None else np.ascontiguousarray(feat_idx, dtype=np.int32)
        cnt = 0 if idx is None else len(idx)
        cap = N if idx is None else max(cnt, 1)
        preds = np.zeros(cap, dtype=PREDICTION_DTYPE)
        Hs = np.zeros((cap, 2, 13))
        Hf = np.zeros((cap, 2, 6))
        HP = np.zeros((cap, 2, self.n)) if want_HP else None
        k = self.L.orc_predict_measurements(self.h, _p(idx), cnt, _p(preds), _p(Hs), _p(Hf), _p(HP))
        out = (preds[:k].copy(), Hs[:k].copy(), Hf[:k].copy())
        return out + (HP[:k].copy(),) if want_HP else out

    def predict_measurement_state(self, x13, R, feature_pos):
        x13 = np.ascontiguousarray(x13, dtype=np.float64)
        R = np.ascontiguousarray(R, dtype=np.float64)
        fp = np.ascontiguousarray(feature_pos, dtype=np.float64)
        preds = np.zeros(max(self.N, 1), dtype=PREDICTION_DTYPE)
        k = self.L.orc_predict_measurement_state(self.h, _p(x13), _p(R), _p(fp), None, 0, _p(preds))
        return preds[:k].copy()

    def ellipse(self, S):
        S = np.ascontiguousarray(S, dtype=np.float64).reshape(4)
        ax = np.zeros(2, dtype=np.float32)
        ang = C.c_double(0)
        self.L.orc_ellipse(_p(S), _p(ax), C.byref(ang))
        return ax, ang.value

    def point_in_ellipse(self, px, py, cx, cy, aw, ah, angle):
        return bool(self.L.orc_point_in_ellipse(px, py, cx, cy, int(aw), int(ah), angle))

    def match(self, preds, kps, desc):
        preds = np.ascontiguousarray(preds, dtype=PREDICTION_DTYPE)
        kps = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        desc = self._desc(desc)
        out = np.zeros(max(len(preds), 1), dtype=MATCH_DTYPE)
        k = self.L.orc_match(self.h, _p(preds), len(preds), _p(kps), _p(desc), len(kps), _p(out))
        return out[:k].copy()

    def ransac(self, preds, Hs, Hf, matches):
        M = len(matches)
        mask = np.zeros(max(M, 1), dtype=np.uint8)
        counts = np.full(max(M, 1), -1, dtype=np.int32)
        preds = np.ascontiguousarray(preds)
        Hs = np.ascontiguousarray(Hs)
        Hf = np.ascontiguousarray(Hf)
        matches = np.ascontiguousarray(matches)
        nh = self.L.orc_ransac(self.h, _p(preds), _p(Hs), _p(Hf), _p(matches), M, _p(mask), _p(counts))
        return mask[:M].astype(bool), counts[:nh].copy()

    def update(self, matches, preds, Hs, Hf, variant=LITERAL):
        matches = np.ascontiguousarray(matches)
        preds = np.ascontiguousarray(preds)
        Hs = np.ascontiguousarray(Hs)
        Hf = np.ascontiguousarray(Hf)
        return self.L.orc_update(self.h, _p(matches), _p(preds), _p(Hs), _p(Hf), len(matches), variant)

    def rescue(self, matches, preds):
        M = len(matches)
        mask = np.zeros(max(M, 1), dtype=np.uint8)
        matches = np.ascontiguousarray(matches)
        preds = np.ascontiguousarray(preds)
        self.L.orc_rescue(self.h, _p(matches), _p(preds), M, _p(mask))
        return mask[:M].astype(bool)

    def step(self, kps, desc, variant=LITERAL):
        kps = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        desc = self._desc(desc)
        info = EkfStepInfo()
        self.L.orc_step(self.h, _p(kps), _p(desc), len(kps), variant, C.byref(info))
        return info


    # -- matcher mode B (one current image per process)
    def set_image(self, image):
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if img.ndim == 2:
            h, w = img.shape
            ch = 1
        else:
            h, w, ch = img.shape
        assert self.L.orc_image_set(_p(img), w, h, w * ch, ch) == 0

    def image_level(self, level):
        w, h = C.c_int(0), C.c_int(0)
        ptr = self.L.orc_image_level(int(level), C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(ptr, (h.value * w.value,)).copy().reshape(h.value, w.value)

    def capture_templates(self, feat_idx, uv):
        idx = np.ascontiguousarray(feat_idx, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        assert self.L.orc_capture_templates(self.h, _p(idx), _p(uv), len(idx)) == 0

    def templates(self):
        return np.ctypeslib.as_array(self.L.orc_templates(), (self.N * 363,)).copy().reshape(self.N, 3, 11, 11)

    def match_ncc(self, preds):
        preds = np.ascontiguousarray(preds, dtype=PREDICTION_DTYPE)
        out = np.zeros(max(len(preds), 1), dtype=MATCH_DTYPE)
        k = self.L.orc_match_ncc(self.h, _p(preds), len(preds), _p(out))
        return out[:k].copy()

    def detect_new_features(self, preds, max_new, divide_times=2, mask_ellipse_size=10.0, min_response=1e9):
        preds = np.ascontiguousarray(preds, dtype=PREDICTION_DTYPE)
        out = np.zeros((max(int(max_new), 1), 2))
        k = self.L.orc_detect_new_features(_p(preds), len(preds), int(max_new), int(divide_times),
                                           float(mask_ellipse_size), float(min_response), _p(out))
        return out[:k].copy()

    def step_image(self, image, variant=LITERAL):
        self.set_image(image)
        info = EkfStepInfo()
        self.L.orc_step_image(self.h, variant, C.byref(info))
        return info


def align_to_matches(preds, Hs, Hf, matches):
    """EKF/EKF.cpp:368-392: predictions / Jacobians re-ordered to match order."""
    lut = {int(p["featureIndex"]): k for k, p in enumerate(preds)}
    sel = np.array([lut[int(m["featureIndex"])] for m in matches], dtype=np.int64)
    return preds[sel], Hs[sel], Hf[sel]


# ---- one oracle run shared by the tests that step the SAME frames (the CPU oracle needs ~25 s per N = 2000 frame: the GPU suite
# spent a third of its time repeating identical oracle runs).  One entry is kept: the tests that share a run are adjacent.
_ORACLE_RUNS = {}


def cached_oracle_run(key, build_fn):
    """build_fn() -> any; the result of the first call with `key` serves the following ones (read-only use)."""
    if key not in _ORACLE_RUNS:
        _ORACLE_RUNS.clear()
        _ORACLE_RUNS[key] = build_fn()
    return _ORACLE_RUNS[key]


def oracle_frames(seq, frames, P0, key, variant=ALGORITHMIC):
    """[(step info, x13, feature_pos, P) after every frame] of `frames` oracle steps of `seq` started from covariance P0, cached
    under `key`."""
    def run():
        n = seq.n_features
        o = Oracle(seq.cam, seq.par, n + 8)
        o.set_state(seq.x13, seq.feature_pos, seq.feature_type, seq.feature_desc, P0)
        out = []
        for t in range(frames):
            info = o.step(*seq.frames[t], variant)
            out.append((info, np.array(o.x13(), copy=True), np.array(o.feature_pos(), copy=True), np.array(o.P(), copy=True)))
        return out
    return cached_oracle_run(key, run)
