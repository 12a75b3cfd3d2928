"""ctypes binding of libekf_engine.so (the HIP engine).  There is no CPU fallback: if the library is missing or
no MI355X is visible, construction fails loudly.

The names mirror the reference's stage functions (include/ekf_engine.h cites each reference file:line):
``predict`` = stateAndCovariancePrediction, ``predict_measurements`` = predictCameraMeasurements,
``match`` = matchPredictedFeatures (downstream of the detector), ``ransac``, ``update``, ``rescue``,
``step`` = EKF::step.
"""
import ctypes as C
import os

import numpy as np

from .ekftypes import (DESC_BYTES, KEYPOINT_DTYPE, MATCH_DTYPE, PREDICTION_DTYPE, STATUS_NAMES, EkfCamera, EkfParams,
                       EkfStepInfo)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libekf_engine.so")
PRECISION_F64, PRECISION_F32, PRECISION_F32_EXACT, PRECISION_F64_EXACT, PRECISION_AUTO = 0, 1, 2, 3, 4


class EkfEngineConfig(C.Structure):
    _fields_ = [
        ("cam", EkfCamera),
        ("par", EkfParams),
        ("max_features", C.c_int32),
        ("max_keypoints", C.c_int32),
        ("precision", C.c_int32),
        ("device", C.c_int32),
        ("ransac_batch", C.c_int32),
        ("flags", C.c_int32),
    ]


class EkfStageTimes(C.Structure):
    _fields_ = [
        ("prediction_ms", C.c_double),
        ("matching_ms", C.c_double),
        ("ransac_ms", C.c_double),
        ("update_li_ms", C.c_double),
        ("rescue_ms", C.c_double),
        ("update_hi_ms", C.c_double),
        ("p_update_kernel_ms", C.c_double),
        ("p_update_launches", C.c_int64),
        ("p_update_flops", C.c_double),
        ("p_update_bytes", C.c_double),
        ("steps", C.c_int64),
    ]


# every symbol include/ekf_engine.h declares: name -> (restype, argtypes)
_vp, _i = C.c_void_p, C.c_int
ABI = {
    "ekf_engine_create": (_i, [C.POINTER(EkfEngineConfig), C.POINTER(_vp)]),
    "ekf_engine_destroy": (None, [_vp]),
    "ekf_last_error": (C.c_char_p, [_vp]),
    "ekf_abi_version": (_i, []),
    "ekf_device_count": (_i, []),
    "ekf_set_state": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "ekf_get_state": (_i, [_vp, _vp, _vp, _vp]),
    "ekf_get_map_features": (_i, [_vp, _vp, _vp, _vp]),
    "ekf_reset": (_i, [_vp]),
    "ekf_add_features": (_i, [_vp, _vp, _vp, _i]),
    "ekf_remove_features": (_i, [_vp, _vp, _i]),
    "ekf_remove_bad_features": (_i, [_vp, C.POINTER(_i)]),
    "ekf_convert_inverse_depth_to_depth": (_i, [_vp, C.POINTER(_i)]),
    "ekf_get_feature_layout": (_i, [_vp, _vp, _vp]),
    "ekf_keep_step_predictions": (_i, [_vp, _i]),
    "ekf_get_step_predictions": (_i, [_vp, _vp, C.POINTER(_i)]),
    "ekf_get_camera_covariance": (_i, [_vp, _vp]),
    "ekf_get_unseen_features": (_i, [_vp, _vp, C.POINTER(_i)]),
    "ekf_state_dim": (_i, [_vp]),
    "ekf_descriptor_bytes": (_i, [_vp]),
    "ekf_num_features": (_i, [_vp]),
    "ekf_predict": (_i, [_vp]),
    "ekf_predict_measurements": (_i, [_vp, _vp, _i, _vp, C.POINTER(_i), _vp, _vp]),
    "ekf_predict_measurement_state": (_i, [_vp, _vp, C.POINTER(_i)]),
    "ekf_match": (_i, [_vp, _vp, _vp, _i, _vp, C.POINTER(_i)]),
    "ekf_ransac": (_i, [_vp, _vp, _i, _vp, C.POINTER(_i)]),
    "ekf_update": (_i, [_vp, _vp, _i]),
    "ekf_update_only_state": (_i, [_vp, _vp, _i]),
    "ekf_rescue": (_i, [_vp, _vp, _i, _vp]),
    "ekf_step": (_i, [_vp, _vp, _vp, _i, C.POINTER(EkfStepInfo)]),
    "ekf_frames_upload": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ekf_step_frame": (_i, [_vp, _i, C.POINTER(EkfStepInfo)]),
    "ekf_set_async_errors": (_i, [_vp, _i]),
    "ekf_set_update_path": (_i, [_vp, _i]),
    "ekf_set_sweep_mode": (_i, [_vp, _i]),
    "ekf_get_precision": (_i, [_vp]),
    "ekf_image_upload": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "ekf_get_image_level": (_i, [_vp, _i, _vp, C.POINTER(_i), C.POINTER(_i)]),
    "ekf_capture_templates": (_i, [_vp, _vp, _vp, _i]),
    "ekf_match_ncc": (_i, [_vp, _vp, C.POINTER(_i)]),
    "ekf_step_image": (_i, [_vp, _vp, _i, _i, _i, _i, C.POINTER(EkfStepInfo)]),
    "ekf_detect_new_features": (_i, [_vp, _i, _i, C.c_double, C.c_double, _vp, C.POINTER(_i)]),
    "ekf_images_upload": (_i, [_vp, _i, _vp, _i, _i, _i, _i]),
    "ekf_select_staged_image": (_i, [_vp, _i]),
    "ekf_step_staged_image": (_i, [_vp, _i, C.POINTER(EkfStepInfo)]),
    "ekf_timing_enable": (_i, [_vp, _i]),
    "ekf_timing_reset": (_i, [_vp]),
    "ekf_timing_get": (_i, [_vp, C.POINTER(EkfStageTimes)]),
    "ekf_synchronize": (_i, [_vp]),
    "ekf_timing_p_update_launches": (_i, [_vp, _i, _vp, _vp, C.POINTER(_i)]),
    "ekf_timing_sweep": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                              C.POINTER(C.c_double)]),
    "ekf_timing_sweep_launches": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "ekf_round_covariance_to_f32": (_i, [_vp]),
    "ekf_get_sweep_retries": (_i, [_vp]),
    "ekf_shard_rows": (_i, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "ekf_engine_create_sharded": (_i, [C.POINTER(EkfEngineConfig), _i, _i, C.POINTER(_vp)]),
    "ekf_set_exchange": (_i, [_vp, _vp, _vp]),
    "ekf_comm_unique_id": (_i, [_vp]),
    "ekf_comm_init": (_i, [_vp, _vp]),
    "ekf_shard_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "ekf_shard_counters": (_i, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ekf_device_copy": (_i, [_vp, _vp, _vp, C.c_size_t]),
}

# include/ekf_test_hooks.h: fault injection for the tests, not part of the boundary
TEST_HOOKS = {
    "ekf_debug_stall_next_sweep": (_i, [_vp]),
    "ekf_debug_stall_sweep_after": (_i, [_vp, _i]),
    "ekf_debug_plane0_pieces": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "ekf_debug_dense_products": (_i, [_vp, _i]),
}

# int fn(void *user, int what, void *device_base, size_t row_bytes, const int32_t *row_begin, int world, int rank)
EXCHANGE_FN = C.CFUNCTYPE(_i, _vp, _i, _vp, C.c_size_t, C.POINTER(C.c_int32), _i, _i)
XCHG_HP, XCHG_PRED_S = 0, 1

_lib = None


def load_library():
    """Loads libekf_engine.so.  torch (if installed) is imported first so that one HIP runtime serves both."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m openekfmonoslam_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback."
        )
    try:
        import torch  # noqa: F401  (pulls in torch's libamdhip64 before ours resolves the same SONAME)
    except Exception:
        pass
    override = os.environ.get("EKF_ENGINE_LIB")  # A/B timing of another build of the engine (scripts/build_variant.sh)
    L = C.CDLL(os.path.abspath(override) if override else LIB_PATH)
    for name, (rt, at) in list(ABI.items()) + list(TEST_HOOKS.items()):
        if override and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        fn.restype = rt
        fn.argtypes = at
    _lib = L
    return L


class EkfError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def shard_rows(n_features, world, rank):
    lo, hi = _i(0), _i(0)
    rc = load_library().ekf_shard_rows(n_features, world, rank, C.byref(lo), C.byref(hi))
    if rc:
        raise EkfError(rc)
    return lo.value, hi.value


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 of a sharded filter creates it, the host distributes it to every rank)."""
    buf = np.zeros(128, dtype=np.uint8)
    rc = load_library().ekf_comm_unique_id(_p(buf))
    if rc:
        raise EkfError(rc, "ekf_comm_unique_id (librccl.so.1 not loadable?)")
    return buf


class EkfEngine:
    """One device-resident filter (state, covariance, map, per-frame tables) on one MI355X."""

    def __init__(self, cam, par, max_features, max_keypoints=0, precision=PRECISION_F64, device=-1, ransac_batch=0,
                 shard=None, descriptor_cols_f32=0):
        """shard = (rank, world) creates one rank of a row-sharded filter; install the exchange with
        set_exchange before the first prediction.  descriptor_cols_f32 > 0: CV_32F descriptors of that many floats
        matched by L2 distance (EKF_DESCRIPTOR_F32_L2) instead of 32-byte binary descriptors / Hamming."""
        self.L = load_library()
        cfg = EkfEngineConfig()
        cfg.cam, cfg.par = cam, par
        cfg.max_features, cfg.max_keypoints = int(max_features), int(max_keypoints)
        cfg.precision, cfg.device, cfg.ransac_batch = int(precision), int(device), int(ransac_batch)
        cfg.flags = (1 | (int(descriptor_cols_f32) << 8)) if descriptor_cols_f32 else 0
        h = _vp()
        if shard is None:
            rc = self.L.ekf_engine_create(C.byref(cfg), C.byref(h))
        else:
            rc = self.L.ekf_engine_create_sharded(C.byref(cfg), int(shard[0]), int(shard[1]), C.byref(h))
        self._xchg_ref = None
        if rc:
            raise EkfError(rc, "ekf_engine_create failed (no MI355X visible?)")
        self.h = h
        self.cap = int(max_features)
        self.precision = int(self.L.ekf_get_precision(self.h))  # (what PRECISION_AUTO resolved to)
        self.desc_bytes = self.L.ekf_descriptor_bytes(self.h)
        self.desc_dtype = np.float32 if descriptor_cols_f32 else np.uint8

    def _desc(self, desc):
        """descriptor matrix -> contiguous bytes, one row of desc_bytes per descriptor"""
        if desc is None:
            return None
        d = np.ascontiguousarray(desc, dtype=self.desc_dtype)
        return d.view(np.uint8).reshape(-1, self.desc_bytes)

    def set_exchange(self, fn):
        """fn(what, device_base, row_bytes, row_begin[world+1], world, rank) -> 0 on success; called by the engine
        (from C, with its stream idle) whenever a replicated per-feature table has to be completed across ranks."""
        def tramp(_user, what, base, row_bytes, rb, world, rank):
            try:
                return int(fn(int(what), int(base), int(row_bytes), [int(rb[i]) for i in range(world + 1)], int(world),
                              int(rank)) or 0)
            except Exception:  # an exception must not unwind through the C frames
                import traceback

                traceback.print_exc()
                return 1

        self._xchg_ref = EXCHANGE_FN(tramp)
        self._chk(self.L.ekf_set_exchange(self.h, C.cast(self._xchg_ref, _vp), None))

    def comm_init(self, unique_id):
        """collective over the ranks of a sharded filter: the engine's own RCCL communicator (in-stream exchange)"""
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert uid.size == 128
        self._chk(self.L.ekf_comm_init(self.h, _p(uid)))

    def shard_info(self):
        r, w, lo, hi = _i(0), _i(1), _i(0), _i(0)
        self._chk(self.L.ekf_shard_info(self.h, C.byref(r), C.byref(w), C.byref(lo), C.byref(hi)))
        return r.value, w.value, lo.value, hi.value

    def shard_counters(self):
        """(bytes of digit planes received so far, first own column, one past the last own column) -- exact configuration"""
        b, c0, c1 = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        self._chk(self.L.ekf_shard_counters(self.h, C.byref(b), C.byref(c0), C.byref(c1)))
        return b.value, c0.value, c1.value

    def device_copy(self, dst, src, nbytes):
        self._chk(self.L.ekf_device_copy(self.h, _vp(dst), _vp(src), nbytes))

    def close(self):
        if getattr(self, "h", None):
            self.L.ekf_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc, allow=()):
        if rc and rc not in allow:
            raise EkfError(rc, (self.L.ekf_last_error(self.h) or b"").decode())
        return rc

    # ---- state
    @property
    def n(self):
        return self.L.ekf_state_dim(self.h)

    @property
    def N(self):
        return self.L.ekf_num_features(self.h)

    def set_state(self, x13, feature_pos, feature_type, desc, P):
        x13 = np.ascontiguousarray(x13, dtype=np.float64)
        fp = np.ascontiguousarray(feature_pos, dtype=np.float64).reshape(-1, 6)
        ft = None if feature_type is None else np.ascontiguousarray(feature_type, dtype=np.int32)
        d = self._desc(desc)
        Pm = None if P is None else np.ascontiguousarray(P, dtype=np.float64)
        self._chk(self.L.ekf_set_state(self.h, _p(x13), len(fp), _p(fp), _p(ft), _p(d), _p(Pm)))

    def get_state(self, want_P=True, P_out=None):
        """P_out: an existing [n, n] float64 array to fill (a sharded engine writes only the rows it holds, so the
        ranks of a group can assemble the whole matrix in one buffer)."""
        x = np.zeros(13)
        fp = np.zeros((max(self.N, 1), 6))
        P = P_out if P_out is not None else (np.zeros((self.n, self.n)) if want_P else None)
        self._chk(self.L.ekf_get_state(self.h, _p(x), _p(fp), _p(P)))
        return x, fp[: self.N], P

    def camera_covariance(self):
        P = np.zeros((13, 13))
        self._chk(self.L.ekf_get_camera_covariance(self.h, _p(P)))
        return P

    def unseen_features(self):
        idx = np.zeros(max(self.N, 1), dtype=np.int32)
        n = C.c_int(0)
        self._chk(self.L.ekf_get_unseen_features(self.h, _p(idx), C.byref(n)))
        return idx[: n.value].copy()

    def set_async_errors(self, on=True):
        """no read-back at the end of step(): a failed second update is reported by the next step instead"""
        if hasattr(self.L, "ekf_set_async_errors"):
            self._chk(self.L.ekf_set_async_errors(self.h, 1 if on else 0))

    @property
    def sweep_retries(self):
        """updates re-run on the launch-per-panel sweep because their persistent sweep timed out (ekf_get_sweep_retries)"""
        return int(self.L.ekf_get_sweep_retries(self.h))

    def stall_sweep(self, skip=0):
        """test hook (include/ekf_test_hooks.h): the persistent sweep after the next `skip` ones runs without its chain workgroup"""
        self._chk(self.L.ekf_debug_stall_sweep_after(self.h, int(skip)))

    def plane0_pieces(self):
        """test hook: (non-zero, all) 16 x 32 pieces of digit plane 0 of B in the last update (exact configurations)"""
        a, b = _i(0), _i(0)
        self._chk(self.L.ekf_debug_plane0_pieces(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dense_products(self, on=True):
        """test hook: the exact downdate / int8 GEMM multiply every digit product (the zero-piece tables are not consulted)"""
        self._chk(self.L.ekf_debug_dense_products(self.h, 1 if on else 0))

    def set_update_path(self, path):
        """0: by size, 1: B = inv(L) H P inside the Cholesky sweep, 2: explicit inverse + GEMM (ekf_engine.h)"""
        self._chk(self.L.ekf_set_update_path(self.h, int(path)))

    def set_sweep_mode(self, mode):
        """2: by size (default: ONE persistent launch per update on maps below 8192 state columns, launches per panel above),
        3: the persistent sweep wherever it is available, 4: launches per panel (1 / 0: one / two panels per launch); ekf_engine.h"""
        self._chk(self.L.ekf_set_sweep_mode(self.h, int(mode)))

    def keep_step_predictions(self, on=True):
        self._chk(self.L.ekf_keep_step_predictions(self.h, 1 if on else 0))

    def step_predictions(self):
        """predictions of the last step's full prediction, as they were before the updates (drawPrediction's input)"""
        from .ekftypes import PREDICTION_DTYPE

        n = C.c_int(0)
        self._chk(self.L.ekf_get_step_predictions(self.h, None, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=PREDICTION_DTYPE)
        self._chk(self.L.ekf_get_step_predictions(self.h, _p(out), C.byref(n)))
        return out[: n.value].copy()

    # ---- map management
    def reset(self):
        self._chk(self.L.ekf_reset(self.h))

    def add_features(self, uv, desc=None):
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        d = self._desc(desc)
        self._chk(self.L.ekf_add_features(self.h, _p(uv), _p(d), len(uv)))

    def remove_features(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self._chk(self.L.ekf_remove_features(self.h, _p(idx), len(idx)))

    def remove_bad_features(self):
        k = _i(0)
        self._chk(self.L.ekf_remove_bad_features(self.h, C.byref(k)))
        return k.value

    def convert_inverse_depth_to_depth(self):
        k = _i(-1)
        self._chk(self.L.ekf_convert_inverse_depth_to_depth(self.h, C.byref(k)))
        return k.value

    def feature_layout(self):
        t = np.zeros(max(self.N, 1), dtype=np.int32)
        c = np.zeros(max(self.N, 1), dtype=np.int32)
        self._chk(self.L.ekf_get_feature_layout(self.h, _p(t), _p(c)))
        return t[: self.N], c[: self.N]

    def get_map_features(self):
        """(descriptors [N,32] u8, timesPredicted [N] u32, timesMatched [N] u32)."""
        N = max(self.N, 1)
        d = np.zeros((N, self.desc_bytes), dtype=np.uint8)
        tp = np.zeros(N, dtype=np.uint32)
        tm = np.zeros(N, dtype=np.uint32)
        self._chk(self.L.ekf_get_map_features(self.h, _p(d), _p(tp), _p(tm)))
        return d[: self.N].view(self.desc_dtype), tp[: self.N], tm[: self.N]

    # ---- stages
    def predict(self):
        self._chk(self.L.ekf_predict(self.h))

    def predict_measurements(self, feat_idx=None):
        idx = None if feat_idx is None else np.ascontiguousarray(feat_idx, dtype=np.int32)
        cnt = 0 if idx is None else len(idx)
        cap = self.N if idx is None else max(cnt, 1)
        preds = np.zeros(max(cap, 1), dtype=PREDICTION_DTYPE)
        Hs = np.zeros((max(cap, 1), 2, 13))
        Hf = np.zeros((max(cap, 1), 2, 6))
        k = _i(0)
        self._chk(self.L.ekf_predict_measurements(self.h, _p(idx), cnt, _p(preds), C.byref(k), _p(Hs), _p(Hf)))
        return preds[: k.value].copy(), Hs[: k.value].copy(), Hf[: k.value].copy()

    def predict_measurement_state(self):
        preds = np.zeros(max(self.N, 1), dtype=PREDICTION_DTYPE)
        k = _i(0)
        self._chk(self.L.ekf_predict_measurement_state(self.h, _p(preds), C.byref(k)))
        return preds[: k.value].copy()

    def match(self, kps, desc):
        kps = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        desc = self._desc(desc)
        out = np.zeros(max(self.N, 1), dtype=MATCH_DTYPE)
        k = _i(0)
        self._chk(self.L.ekf_match(self.h, _p(kps), _p(desc), len(kps), _p(out), C.byref(k)))
        return out[: k.value].copy()

    def ransac(self, matches):
        matches = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        mask = np.zeros(max(len(matches), 1), dtype=np.uint8)
        nh = _i(0)
        self._chk(self.L.ekf_ransac(self.h, _p(matches), len(matches), _p(mask), C.byref(nh)))
        return mask[: len(matches)].astype(bool), nh.value

    def update(self, matches, allow_errors=()):
        matches = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        return self._chk(self.L.ekf_update(self.h, _p(matches), len(matches)), allow_errors)

    def update_only_state(self, matches):
        matches = np.ascontiguousarray(matches, dtype=MATCH_DTYPE)
        self._chk(self.L.ekf_update_only_state(self.h, _p(matches), len(matches)))

    def rescue(self, outliers):
        outliers = np.ascontiguousarray(outliers, dtype=MATCH_DTYPE)
        mask = np.zeros(max(len(outliers), 1), dtype=np.uint8)
        self._chk(self.L.ekf_rescue(self.h, _p(outliers), len(outliers), _p(mask)))
        return mask[: len(outliers)].astype(bool)

    def step(self, kps, desc):
        kps = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        desc = self._desc(desc)
        info = EkfStepInfo()
        self._chk(self.L.ekf_step(self.h, _p(kps), _p(desc), len(kps), C.byref(info)))
        return info

    # ---- staged sequences
    def upload_frames(self, frames):
        counts = np.array([len(k) for k, _ in frames], dtype=np.int32)
        kps = np.ascontiguousarray(np.concatenate([k for k, _ in frames]), dtype=KEYPOINT_DTYPE)
        desc = self._desc(np.concatenate([d for _, d in frames]))
        self._chk(self.L.ekf_frames_upload(self.h, len(frames), _p(counts), _p(kps), _p(desc)))

    def step_frame(self, i):
        info = EkfStepInfo()
        self._chk(self.L.ekf_step_frame(self.h, int(i), C.byref(info)))
        return info

    # ---- matcher mode B (image in)
    @staticmethod
    def _image_args(image):
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if img.ndim == 2:
            h, w = img.shape
            ch = 1
        else:
            h, w, ch = img.shape
        return img, w, h, w * ch, ch

    def upload_image(self, image):
        img, w, h, stride, ch = self._image_args(image)
        self._chk(self.L.ekf_image_upload(self.h, _p(img), w, h, stride, ch))

    def image_level(self, level):
        w, h = C.c_int(0), C.c_int(0)
        self._chk(self.L.ekf_get_image_level(self.h, int(level), None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), dtype=np.uint8)
        self._chk(self.L.ekf_get_image_level(self.h, int(level), _p(out), C.byref(w), C.byref(h)))
        return out

    def capture_templates(self, feat_idx, uv):
        idx = np.ascontiguousarray(feat_idx, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        assert len(idx) == len(uv)
        self._chk(self.L.ekf_capture_templates(self.h, _p(idx), _p(uv), len(idx)))

    def match_ncc(self):
        out = np.zeros(max(self.N, 1), dtype=MATCH_DTYPE)
        n = C.c_int(0)
        self._chk(self.L.ekf_match_ncc(self.h, _p(out), C.byref(n)))
        return out[: n.value]

    def step_image(self, image):
        img, w, h, stride, ch = self._image_args(image)
        info = EkfStepInfo()
        self._chk(self.L.ekf_step_image(self.h, _p(img), w, h, stride, ch, C.byref(info)))
        return info

    def detect_new_features(self, max_new, divide_times=2, mask_ellipse_size=10.0, min_response=1e9):
        """detectNewImageFeatures on the current image -> [k, 2] pixel positions (k <= max_new)."""
        out = np.zeros((max(int(max_new), 1), 2))
        n = C.c_int(0)
        self._chk(self.L.ekf_detect_new_features(self.h, int(max_new), int(divide_times), float(mask_ellipse_size),
                                                 float(min_response), _p(out), C.byref(n)))
        return out[: n.value].copy()

    def upload_images(self, images):
        arr = np.ascontiguousarray(np.stack(images), dtype=np.uint8)
        n = arr.shape[0]
        _, w, h, stride, ch = self._image_args(arr[0])
        self._chk(self.L.ekf_images_upload(self.h, n, _p(arr), w, h, stride, ch))

    def select_staged_image(self, i):
        self._chk(self.L.ekf_select_staged_image(self.h, int(i)))

    def step_staged_image(self, i):
        info = EkfStepInfo()
        self._chk(self.L.ekf_step_staged_image(self.h, int(i), C.byref(info)))
        return info

    # ---- instrumentation
    def timing(self, on=True):
        self._chk(self.L.ekf_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        self._chk(self.L.ekf_timing_reset(self.h))

    def timing_get(self):
        t = EkfStageTimes()
        self._chk(self.L.ekf_timing_get(self.h, C.byref(t)))
        return t

    def p_update_launches(self):
        """(m_rows[int32], ms[float32]) of every P-update launch since the last timing_reset."""
        k = _i(0)
        self._chk(self.L.ekf_timing_p_update_launches(self.h, 0, None, None, C.byref(k)))
        m = np.zeros(max(k.value, 1), dtype=np.int32)
        ms = np.zeros(max(k.value, 1), dtype=np.float32)
        self._chk(self.L.ekf_timing_p_update_launches(self.h, k.value, _p(m), _p(ms), C.byref(k)))
        return m[: k.value], ms[: k.value]

    def sweep_timing(self):
        """dict: HIP-event ms over the Cholesky sweep's launches, panels, updates, fp64 flops of the factorisations, flops of
        the rows of B formed in the same launches (since the last timing_reset)."""
        ms, fl64, flb = C.c_double(0), C.c_double(0), C.c_double(0)
        panels, updates = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.ekf_timing_sweep(self.h, C.byref(ms), C.byref(panels), C.byref(updates), C.byref(fl64), C.byref(flb)))
        launches, slice_ms = C.c_int64(0), C.c_double(0)
        self._chk(self.L.ekf_timing_sweep_launches(self.h, C.byref(launches), C.byref(slice_ms)))
        return {"ms": ms.value, "panels": panels.value, "updates": updates.value, "flops_fp64": fl64.value, "flops_b": flb.value,
                "launches": launches.value, "slice_ms": slice_ms.value}

    def round_covariance_to_f32(self):
        """fp64 engines: every entry of P to its nearest fp32 value, in place (storage-floor measurements)."""
        self._chk(self.L.ekf_round_covariance_to_f32(self.h))

    def synchronize(self):
        self._chk(self.L.ekf_synchronize(self.h))
