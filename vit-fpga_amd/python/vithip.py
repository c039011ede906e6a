"""ctypes binding of libvithip.so — the thin host-side mirror used by tests/, bench.py and
__graft_entry__.py.  Everything goes through the C ABI declared in include/vithip.h; there is
no Python compute path and no fallback: if the library is missing, import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.environ.get("VITHIP_LIB") or os.path.join(PKG_ROOT, "libvithip.so")   # VITHIP_LIB: A/B builds (tools/)

DTYPE_BF16, DTYPE_FP16, DTYPE_FP8 = 0, 1, 2
FLAG_LN_FOLD_OFF, FLAG_LN_FOLD_ON, FLAG_W8_E4M3, FLAG_CLS_TAIL = 1, 2, 4, 8   # vh_config.flags
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_BIAS_F32, EPI_PATCH, EPI_LNFOLD, EPI_LNFOLD_GELU, EPI_RESID_LN, EPI_RESID_SPLIT, EPI_PATCH_SPLIT = range(10)
ACT_IDENTITY, ACT_RELU2, ACT_RELU, ACT_HARDTANH, ACT_GELU = range(5)

STAGES = ["im2col", "patch_gemm", "cls_rows", "layernorm", "qkv_gemm", "attention", "proj_gemm",
          "fc1_gemm", "fc2_gemm", "final_layernorm", "head_gemm", "ln_stats"]


class VhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvithip error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("patch_size", C.c_int32), ("channels", C.c_int32),
                ("dim", C.c_int32), ("heads", C.c_int32), ("mlp_dim", C.c_int32),
                ("layers", C.c_int32), ("classes", C.c_int32), ("dtype", C.c_int32),
                ("max_batch", C.c_int32), ("ln_eps", C.c_float), ("flags", C.c_int32)]


# every exported symbol of include/vithip.h: name -> (restype, argtypes)
_vp, _i, _i64, _u64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_size_t, C.c_float
_pi = C.POINTER(C.c_int)
SYMBOLS = {
    "vh_abi_version": (_i, []),
    "vh_device_count": (_i, [_pi]),
    "vh_last_error": (C.c_char_p, [_vp]),
    "vh_malloc": (_i, [_i, _sz, C.POINTER(_vp)]),
    "vh_free": (_i, [_i, _vp]),
    "vh_memcpy_h2d": (_i, [_i, _vp, _vp, _sz]),
    "vh_memcpy_d2h": (_i, [_i, _vp, _vp, _sz]),
    "vh_device_synchronize": (_i, [_i]),
    "vh_create": (_i, [C.POINTER(Config), _i, C.POINTER(_vp)]),
    "vh_destroy": (_i, [_vp]),
    "vh_get_config": (_i, [_vp, C.POINTER(Config)]),
    "vh_get_ln_fold": (_i, [_vp, _pi]),
    "vh_get_ln_guard": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), _pi]),
    "vh_get_fp8_guard": (_i, [_vp, C.POINTER(_f), C.POINTER(_f)]),
    "vh_weight_blob_bytes": (_sz, [C.POINTER(Config)]),
    "vh_load_weights": (_i, [_vp, _vp, _sz]),
    "vh_load_weights_device": (_i, [_vp, _vp, _sz]),
    "vh_init_weights_seeded": (_i, [_vp, _u64]),
    "vh_export_weights": (_i, [_vp, _vp, _sz]),
    "vh_export_weights_device": (_i, [_vp, _vp, _sz]),
    "vh_forward": (_i, [_vp, _vp, _i, _vp]),
    "vh_forward_device": (_i, [_vp, _vp, _i, _vp]),
    "vh_forward_device_async": (_i, [_vp, _vp, _i, _vp, _i]),
    "vh_synchronize": (_i, [_vp]),
    "vh_fill_input_seeded": (_i, [_vp, _u64, _i, _vp]),
    "vh_last_forward_us": (_i, [_vp, C.POINTER(_i64)]),
    "vh_last_kernel_ms": (_i, [_vp, C.POINTER(C.c_double)]),
    "vh_profile_forward": (_i, [_vp, _vp, _i, _vp, C.POINTER(C.c_double), _i, _pi]),
    "vh_stage_name": (C.c_char_p, [_i]),
    "vh_filter_create": (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_void_p)]),
    "vh_filter_destroy": (_i, [_vp]),
    "vh_filter_free_slots": (_i, [_vp, _pi]),
    "vh_filter_submit": (_i, [_vp, _vp]),
    "vh_filter_collect": (_i, [_vp, _vp]),
    "vh_filter_last_error": (C.c_char_p, [_vp]),
    "vh_device_mem_info": (_i, [_i, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "vh_blob_file_config": (_i, [C.c_char_p, C.POINTER(Config)]),
    "vh_blob_file_read": (_i, [C.c_char_p, _vp, _sz]),
    "vh_save_weights_file": (_i, [_vp, C.c_char_p]),
    "vh_load_weights_file": (_i, [_vp, C.c_char_p]),
    "vh_ring_create": (_i, [_vp, _i, _i]),
    "vh_ring_destroy": (_i, [_vp]),
    "vh_ring_free_slots": (_i, [_vp, _pi]),
    "vh_ring_input": (_i, [_vp, C.POINTER(C.POINTER(C.c_float))]),
    "vh_ring_submit": (_i, [_vp, _vp, _i]),
    "vh_ring_collect": (_i, [_vp, _vp, _pi]),
    "vh_set_graph": (_i, [_vp, _i]),
    "vh_get_graph": (_i, [_vp, _pi, _pi]),
    "vh_set_streams": (_i, [_vp, _i]),
    "vh_get_streams": (_i, [_vp, _pi]),
    "vh_set_stage_timing": (_i, [_vp, _i]),
    "vh_get_stage_timing": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), _pi]),
    "vh_set_step_timing": (_i, [_vp, _i]),
    "vh_get_step_timing": (_i, [_vp, C.POINTER(C.c_double), _i, _pi]),
    "vh_debug_read": (_i, [_vp, _i, _vp, _sz]),
    "vh_debug_set_layers": (_i, [_vp, _i]),
    "vh_op_gemm": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "vh_op_gemm_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "vh_op_gemm_fp8_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "vh_op_quantize_rows": (_i, [_vp, _i, _i, C.c_float, _vp, _vp, _vp]),
    "vh_op_gemm_ex": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "vh_op_rowstats_cast": (_i, [_vp, _i64, _i, _f, _vp, _vp, _i, _vp]),
    "vh_op_finalize_stats": (_i, [_vp, _i, _i64, _i, _f, _vp, _vp]),
    "vh_op_rowstats_split": (_i, [_vp, _i64, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "vh_op_fold_ln": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "vh_op_layernorm": (_i, [_vp, _i64, _i, _i64, _vp, _vp, _f, _vp, _i, _vp]),
    "vh_op_attention": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "vh_op_im2col": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "vh_op_cast": (_i, [_vp, _vp, _i64, _i, _vp]),
    "vh_op_fill": (_i, [_vp, _i64, _u64, C.c_uint32, _i, _f, _vp]),
    "vh_bench_gemm": (_i, [_i, _i64, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_double)]),
    "vh_group_create": (_i, [C.POINTER(Config), _pi, _i, C.POINTER(_vp)]),
    "vh_group_destroy": (_i, [_vp]),
    "vh_group_size": (_i, [_vp, _pi]),
    "vh_group_member": (_i, [_vp, _i, C.POINTER(_vp), _pi]),
    "vh_group_last_error": (C.c_char_p, [_vp]),
    "vh_group_shard_bounds": (None, [_i, _i, _i, _pi, _pi]),
    "vh_group_load_weights": (_i, [_vp, _vp, _sz]),
    "vh_group_init_weights_seeded": (_i, [_vp, _u64]),
    "vh_group_broadcast_weights": (_i, [_vp]),
    "vh_group_forward": (_i, [_vp, _vp, _i, _vp]),
    "vh_group_fill_inputs_seeded": (_i, [_vp, _u64, _i]),
    "vh_group_forward_resident": (_i, [_vp, _i, _i]),
    "vh_group_read_logits": (_i, [_vp, _i, _vp]),
    "vh_mlp_create": (_i, [_i, _i, _i, _pi, _i, C.POINTER(_vp)]),
    "vh_mlp_load_params": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "vh_mlp_forward": (_i, [_vp, _vp, _i, _vp]),
    "vh_mlp_last_forward_us": (_i, [_vp, C.POINTER(_i64)]),
    "vh_mlp_init_gradient": (_i, [_vp, _vp, _vp, _i]),
    "vh_mlp_launch_gradient": (_i, [_vp, _i, _f, _f, _vp]),
    "vh_mlp_read_params": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "vh_mlp_last_gradient_us": (_i, [_vp, C.POINTER(_i64)]),
    "vh_mlp_last_error": (C.c_char_p, [_vp]),
    "vh_mlp_destroy": (_i, [_vp]),
}

_lib = None


def lib():
    """Load libvithip.so (in-tree build).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: run `make -C vit-fpga_amd` (or "
                              "__graft_entry__.build()); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, ctx=None):
    if rc != 0:
        msg = lib().vh_last_error(ctx)
        raise VhError(rc, msg.decode() if msg else "?")


def device_count():
    n = C.c_int(0)
    rc = lib().vh_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def blob_file_config(path):
    """Model shape stored in a VHBLOB1 file (host only) -> dict usable as `cfg`."""
    c = Config()
    _check(lib().vh_blob_file_config(os.fsencode(path), C.byref(c)))
    return {k: getattr(c, k) for k in ("image_size", "patch_size", "channels", "dim", "heads", "mlp_dim", "layers", "classes")}, c.ln_eps


def device_free_bytes(device=0):
    free, total = C.c_size_t(0), C.c_size_t(0)
    _check(lib().vh_device_mem_info(device, C.byref(free), C.byref(total)))
    return free.value


def make_config(cfg, dtype=DTYPE_BF16, max_batch=1, ln_eps=1e-6, flags=0):
    return Config(cfg["image_size"], cfg["patch_size"], cfg["channels"], cfg["dim"], cfg["heads"],
                  cfg["mlp_dim"], cfg["layers"], cfg["classes"], dtype, max_batch, ln_eps, flags)


# ---- 16-bit helpers (host side, for building operator inputs / reading operator outputs) -------
def to_bf16_bits(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    return (r >> np.uint32(16)).astype(np.uint16)


def from_bf16_bits(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def to16(a, dtype):
    return to_bf16_bits(a) if dtype == DTYPE_BF16 else np.ascontiguousarray(a, dtype=np.float16).view(np.uint16)


def from16(b, dtype):
    return from_bf16_bits(b) if dtype == DTYPE_BF16 else np.ascontiguousarray(b, dtype=np.uint16).view(np.float16).astype(np.float32)


class DeviceBuffer:
    """A raw HBM allocation owned through vh_malloc/vh_free."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        _check(lib().vh_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, device)
        _check(lib().vh_memcpy_h2d(device, b.ptr, a.ctypes.data, a.nbytes))
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(lib().vh_memcpy_d2h(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().vh_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


FILTER_BLUR3, FILTER_SOBEL3 = 0, 1


class FilterPipeline:
    """vh_filter wrapper: the filter_image / get_filtered_image ring."""

    def __init__(self, height, width, slots=24, kind=FILTER_BLUR3, device=0):
        self.shape = (height, width)
        h = C.c_void_p()
        _check(lib().vh_filter_create(device, height, width, slots, kind, C.byref(h)))
        self.h = h.value

    def _chk(self, rc):
        if rc != 0:
            raise VhError(rc, lib().vh_filter_last_error(self.h).decode())

    def free_slots(self):
        n = C.c_int(0)
        self._chk(lib().vh_filter_free_slots(self.h, C.byref(n)))
        return n.value

    def submit(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == self.shape
        self._chk(lib().vh_filter_submit(self.h, frame.ctypes.data))

    def collect(self):
        out = np.empty(self.shape, dtype=np.uint8)
        self._chk(lib().vh_filter_collect(self.h, out.ctypes.data))
        return out

    def close(self):
        if self.h:
            lib().vh_filter_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VitContext:
    """vh_ctx wrapper: create / load weights / forward, mirroring hip::net_hip's ViT mode."""

    def __init__(self, cfg, dtype=DTYPE_BF16, max_batch=1, device=0, ln_eps=1e-6, flags=0):
        self.cfg, self.dtype, self.device = dict(cfg), dtype, device
        self.c = make_config(cfg, dtype, max_batch, ln_eps, flags)
        h = C.c_void_p()
        _check(lib().vh_create(C.byref(self.c), device, C.byref(h)))
        self.h = h.value
        self._ring_batch = 0

    def close(self):
        if self.h:
            lib().vh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def blob_bytes(self):
        return lib().vh_weight_blob_bytes(C.byref(self.c))

    def ln_fold(self):
        """True when this context folds its LayerNorms into the GEMMs (vh_config.flags, model shape, dtype)."""
        on = C.c_int(0)
        _check(lib().vh_get_ln_fold(self.h, C.byref(on)), self.h)
        return bool(on.value)

    def ln_guard(self):
        """(max |row mean| / sigma seen since the weights were loaded, threshold, tripped) -- the fold's run-time guard."""
        r, t, trip = C.c_float(0), C.c_float(0), C.c_int(0)
        _check(lib().vh_get_ln_guard(self.h, C.byref(r), C.byref(t), C.byref(trip)), self.h)
        return r.value, t.value, bool(trip.value)

    def fp8_guard(self):
        """(largest |x| bound seen on the raw residual rows of an fp8 context, e4m3's limit 448)."""
        a, lim = C.c_float(0), C.c_float(0)
        _check(lib().vh_get_fp8_guard(self.h, C.byref(a), C.byref(lim)), self.h)
        return a.value, lim.value

    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        _check(lib().vh_load_weights(self.h, blob.ctypes.data, blob.nbytes), self.h)

    def load_weights_device(self, ptr, nbytes):
        _check(lib().vh_load_weights_device(self.h, ptr, nbytes), self.h)

    def init_weights_seeded(self, seed):
        _check(lib().vh_init_weights_seeded(self.h, seed), self.h)

    def export_weights(self):
        out = np.empty(self.blob_bytes, dtype=np.uint8)
        _check(lib().vh_export_weights(self.h, out.ctypes.data, out.nbytes), self.h)
        return out

    def export_weights_device(self, ptr, nbytes):
        _check(lib().vh_export_weights_device(self.h, ptr, nbytes), self.h)

    def save_weights_file(self, path):
        _check(lib().vh_save_weights_file(self.h, os.fsencode(path)), self.h)

    def load_weights_file(self, path):
        _check(lib().vh_load_weights_file(self.h, os.fsencode(path)), self.h)

    def forward(self, images):
        """images: [B, H, W, C] fp32 (host).  Returns [B, classes] fp32 logits."""
        images = np.ascontiguousarray(images, dtype=np.float32)
        b = images.shape[0]
        out = np.empty((b, self.cfg["classes"]), dtype=np.float32)
        _check(lib().vh_forward(self.h, images.ctypes.data, b, out.ctypes.data), self.h)
        return out

    def forward_device(self, in_ptr, batch, out_ptr):
        _check(lib().vh_forward_device(self.h, in_ptr, batch, out_ptr), self.h)

    def forward_device_async(self, in_ptr, batch, out_ptr, steps=1):
        _check(lib().vh_forward_device_async(self.h, in_ptr, batch, out_ptr, steps), self.h)

    def synchronize(self):
        _check(lib().vh_synchronize(self.h), self.h)

    def fill_input_seeded(self, seed, batch, in_ptr):
        _check(lib().vh_fill_input_seeded(self.h, seed, batch, in_ptr), self.h)

    def last_forward_us(self):
        v = C.c_int64(0)
        _check(lib().vh_last_forward_us(self.h, C.byref(v)), self.h)
        return v.value

    def last_kernel_ms(self):
        v = C.c_double(0)
        _check(lib().vh_last_kernel_ms(self.h, C.byref(v)), self.h)
        return v.value

    def profile_forward(self, in_ptr, batch, out_ptr):
        n = len(STAGES)
        arr = (C.c_double * (2 * n))()
        nw = C.c_int(0)
        _check(lib().vh_profile_forward(self.h, in_ptr, batch, out_ptr, arr, 2 * n, C.byref(nw)), self.h)
        return {STAGES[i]: (arr[i], int(arr[n + i])) for i in range(n)}

    # ---- pipelined host path (ring of in-flight batches) ----
    def ring_create(self, slots, batch_per_slot):
        _check(lib().vh_ring_create(self.h, slots, batch_per_slot), self.h)
        self._ring_batch = batch_per_slot

    def ring_free_slots(self):
        n = C.c_int(0)
        _check(lib().vh_ring_free_slots(self.h, C.byref(n)), self.h)
        return n.value

    def ring_input(self, batch):
        """numpy view of the pinned staging buffer the next submit will use."""
        p = C.POINTER(C.c_float)()
        _check(lib().vh_ring_input(self.h, C.byref(p)), self.h)
        n = batch * self.cfg["image_size"] ** 2 * self.cfg["channels"]
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(batch, self.cfg["image_size"], self.cfg["image_size"], self.cfg["channels"])

    def ring_submit(self, images=None, batch=None):
        if images is None:
            _check(lib().vh_ring_submit(self.h, None, batch), self.h)
        else:
            images = np.ascontiguousarray(images, dtype=np.float32)
            _check(lib().vh_ring_submit(self.h, images.ctypes.data, images.shape[0]), self.h)

    def ring_collect(self):
        out = np.empty((max(self._ring_batch, 1), self.cfg["classes"]), dtype=np.float32)
        nb = C.c_int(0)
        _check(lib().vh_ring_collect(self.h, out.ctypes.data, C.byref(nb)), self.h)
        return out[:nb.value]

    def set_graph(self, enable=True):
        _check(lib().vh_set_graph(self.h, 1 if enable else 0), self.h)

    def get_graph(self):
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().vh_get_graph(self.h, C.byref(a), C.byref(b)), self.h)
        return bool(a.value), b.value

    def set_streams(self, n):
        _check(lib().vh_set_streams(self.h, n), self.h)

    def get_streams(self):
        n = C.c_int(0)
        _check(lib().vh_get_streams(self.h, C.byref(n)), self.h)
        return n.value

    def set_stage_timing(self, stage_name):
        _check(lib().vh_set_stage_timing(self.h, STAGES.index(stage_name) if stage_name else -1), self.h)

    def get_stage_timing(self):
        avg, mn, n = C.c_double(0), C.c_double(0), C.c_int(0)
        _check(lib().vh_get_stage_timing(self.h, C.byref(avg), C.byref(mn), C.byref(n)), self.h)
        return avg.value, mn.value, n.value

    def set_step_timing(self, on):
        _check(lib().vh_set_step_timing(self.h, 1 if on else 0), self.h)

    def get_step_timing(self, max_steps=4096):
        """Device time in ms of every step of the last forward_device_async call (step timing enabled)."""
        buf = (C.c_double * max_steps)()
        n = C.c_int(0)
        _check(lib().vh_get_step_timing(self.h, buf, max_steps, C.byref(n)), self.h)
        return [buf[i] for i in range(min(n.value, max_steps))]

    def debug_read(self, what, n_floats):
        out = np.empty(n_floats, dtype=np.float32)
        _check(lib().vh_debug_read(self.h, what, out.ctypes.data, n_floats), self.h)
        return out

    def debug_set_layers(self, n):
        _check(lib().vh_debug_set_layers(self.h, n), self.h)


def group_shard_bounds(batch, n, r):
    lo, hi = C.c_int(0), C.c_int(0)
    lib().vh_group_shard_bounds(batch, n, r, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class VitGroup:
    """vh_group wrapper: the N GPUs of one node from one process (one context + host thread per device inside
    libvithip, one RCCL broadcast of the weight blob, contiguous image shards)."""

    def __init__(self, cfg, devices, dtype=DTYPE_BF16, max_batch_per_device=1, ln_eps=1e-6, flags=0):
        self.cfg, self.devices = dict(cfg), list(devices)
        self.c = make_config(cfg, dtype, max_batch_per_device, ln_eps, flags)
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _check(lib().vh_group_create(C.byref(self.c), arr, len(self.devices), C.byref(h)))
        self.h = h.value

    def _chk(self, rc):
        if rc != 0:
            msg = lib().vh_group_last_error(self.h)
            raise VhError(rc, msg.decode() if msg else "?")

    def close(self):
        if self.h:
            lib().vh_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        n = C.c_int(0)
        self._chk(lib().vh_group_size(self.h, C.byref(n)))
        return n.value

    def fp8_guard(self):
        """(largest |x| bound seen on the raw residual rows of an fp8 context, e4m3's limit 448)."""
        a, lim = C.c_float(0), C.c_float(0)
        _check(lib().vh_get_fp8_guard(self.h, C.byref(a), C.byref(lim)), self.h)
        return a.value, lim.value

    def load_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._chk(lib().vh_group_load_weights(self.h, blob.ctypes.data, blob.nbytes))

    def init_weights_seeded(self, seed):
        self._chk(lib().vh_group_init_weights_seeded(self.h, seed))

    def forward(self, images):
        images = np.ascontiguousarray(images, dtype=np.float32)
        out = np.empty((images.shape[0], self.cfg["classes"]), dtype=np.float32)
        self._chk(lib().vh_group_forward(self.h, images.ctypes.data, images.shape[0], out.ctypes.data))
        return out

    def fill_inputs_seeded(self, seed, batch_per_device):
        self._chk(lib().vh_group_fill_inputs_seeded(self.h, seed, batch_per_device))

    def forward_resident(self, batch_per_device, steps=1):
        self._chk(lib().vh_group_forward_resident(self.h, batch_per_device, steps))

    def read_logits(self, batch_per_device):
        out = np.empty((len(self.devices) * batch_per_device, self.cfg["classes"]), dtype=np.float32)
        self._chk(lib().vh_group_read_logits(self.h, batch_per_device, out.ctypes.data))
        return out


class MlpContext:
    """vh_mlp wrapper — the reference's real launch_forward semantics (dense-layer chain)."""

    def __init__(self, n_ins, n_p_l, activation=ACT_RELU2, device=0):
        self.n_ins, self.n_p_l = n_ins, list(n_p_l)
        arr = (C.c_int * len(n_p_l))(*n_p_l)
        h = C.c_void_p()
        _check(lib().vh_mlp_create(device, n_ins, len(n_p_l), arr, activation, C.byref(h)))
        self.h = h.value

    def _chk(self, rc):
        if rc != 0:
            msg = lib().vh_mlp_last_error(self.h)
            raise VhError(rc, msg.decode() if msg else "?")

    def load_params(self, params, bias):
        p = np.ascontiguousarray(params, dtype=np.float32)
        b = np.ascontiguousarray(bias, dtype=np.float32)
        self._chk(lib().vh_mlp_load_params(self.h, p.ctypes.data, p.size, b.ctypes.data, b.size))

    def forward(self, inputs):
        x = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, self.n_ins)
        out = np.empty((x.shape[0], self.n_p_l[-1]), dtype=np.float32)
        self._chk(lib().vh_mlp_forward(self.h, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def init_gradient(self, set_ins, set_outs):
        si = np.ascontiguousarray(set_ins, dtype=np.float32).reshape(-1, self.n_ins)
        so = np.ascontiguousarray(set_outs, dtype=np.float32).reshape(si.shape[0], self.n_p_l[-1])
        self._chk(lib().vh_mlp_init_gradient(self.h, si.ctypes.data, so.ctypes.data, si.shape[0]))

    def launch_gradient(self, iterations, error_threshold, multiplier):
        err = np.zeros(max(iterations, 1), dtype=np.float32)
        self._chk(lib().vh_mlp_launch_gradient(self.h, iterations, error_threshold, multiplier, err.ctypes.data))
        return err[:iterations]

    def read_params(self):
        fan, n_params = self.n_ins, 0
        for n in self.n_p_l:
            n_params += n * fan
            fan = n
        p, b = np.empty(n_params, dtype=np.float32), np.empty(sum(self.n_p_l), dtype=np.float32)
        self._chk(lib().vh_mlp_read_params(self.h, p.ctypes.data, p.size, b.ctypes.data, b.size))
        return p, b

    def last_gradient_us(self):
        us = C.c_int64(0)
        self._chk(lib().vh_mlp_last_gradient_us(self.h, C.byref(us)))
        return us.value

    def close(self):
        if self.h:
            lib().vh_mlp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- operator-level wrappers (device pointers in, nothing hidden) ------------------------------------
def op_gemm(a_ptr, w_ptr, bias_ptr, out_ptr, M, N, K, epilogue, dtype, aux_ptr=None, aux_i=0, variant=0):
    _check(lib().vh_op_gemm(a_ptr, w_ptr, bias_ptr, out_ptr, M, N, K, epilogue, aux_ptr, aux_i, dtype, variant, None))


def op_gemm_fp8(a8_ptr, w8_ptr, scale_ptr, bias_ptr, out_ptr, M, N, K, epilogue, variant=0):
    _check(lib().vh_op_gemm_fp8(a8_ptr, w8_ptr, scale_ptr, bias_ptr, out_ptr, M, N, K, epilogue, variant, None))


def op_quantize_rows(w_ptr, rows, cols, post_scale, w8_ptr, scale_ptr):
    _check(lib().vh_op_quantize_rows(w_ptr, rows, cols, post_scale, w8_ptr, scale_ptr, None))


def op_gemm_fp8_ex(a8_ptr, w8_ptr, scale_ptr, bias_ptr, out_ptr, M, N, K, epilogue, c_ptr=None, stats_ptr=None, out16_ptr=None,
                   partials_ptr=None, variant=0):
    _check(lib().vh_op_gemm_fp8_ex(a8_ptr, w8_ptr, scale_ptr, bias_ptr, out_ptr, M, N, K, epilogue, c_ptr, stats_ptr, out16_ptr,
                                   partials_ptr, variant, None))


# ---- OCP e4m3fn on the host (table-driven; used to build / read fp8 operator operands in tests) ----
def e4m3_table():
    b = np.arange(256, dtype=np.uint32)
    e, m = (b >> 3) & 15, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (1 + m / 8.0) * 2.0 ** (e.astype(np.float64) - 7))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return np.where(b >> 7 == 1, -v, v).astype(np.float32)


def from_e4m3(b):
    return e4m3_table()[np.ascontiguousarray(b, dtype=np.uint8)]


def to_e4m3(x):
    """fp32 -> OCP e4m3fn bytes: round to nearest even, saturating at +-448 (what pack4_e4m3 does on the device)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    pos = e4m3_table()[:127].astype(np.float64)          # codes 0x00 .. 0x7E, ascending (0x7F is NaN)
    a = np.minimum(np.abs(x).astype(np.float64), 448.0)
    hi = np.clip(np.searchsorted(pos, a, side="left"), 1, 126)
    lo = hi - 1
    dl, dh = a - pos[lo], pos[hi] - a
    code = np.where(dl < dh, lo, np.where(dh < dl, hi, np.where(lo % 2 == 0, lo, hi)))
    return (code | np.where(np.signbit(x), 0x80, 0)).astype(np.uint8)


# ---- the lo plane of the split residual: one e4m3 byte per element, scaled (csrc/vh_common.h Lo8) ----
LO8_SCALE = {DTYPE_BF16: 32.0, DTYPE_FP16: 256.0}


def to_lo8(residue, dtype):
    return to_e4m3(np.asarray(residue, dtype=np.float32) * np.float32(LO8_SCALE[dtype]))


def from_lo8(b, dtype):
    return from_e4m3(b) / np.float32(LO8_SCALE[dtype])


def op_gemm_ex(a_ptr, w_ptr, bias_ptr, out_ptr, M, N, K, epilogue, dtype, aux_ptr=None, aux_i=0, stats_ptr=None,
               out16_ptr=None, partials_ptr=None, variant=0):
    _check(lib().vh_op_gemm_ex(a_ptr, w_ptr, bias_ptr, out_ptr, M, N, K, epilogue, aux_ptr, aux_i, stats_ptr, out16_ptr,
                               partials_ptr, dtype, variant, None))


def op_rowstats_cast(x_ptr, rows, dim, eps, x16_ptr, stats_ptr, dtype):
    _check(lib().vh_op_rowstats_cast(x_ptr, rows, dim, eps, x16_ptr, stats_ptr, dtype, None))


def op_rowstats_split(x_ptr, rows, dim, eps, hi_ptr, lo_ptr, stats_ptr, dtype):
    _check(lib().vh_op_rowstats_split(x_ptr, rows, dim, eps, hi_ptr, lo_ptr, stats_ptr, dtype, None))


def op_finalize_stats(partials_ptr, nblk, rows, dim, eps, stats_ptr):
    _check(lib().vh_op_finalize_stats(partials_ptr, nblk, rows, dim, eps, stats_ptr, None))


def op_fold_ln(w_ptr, b_ptr, gamma_ptr, beta_ptr, rows, dim, scale, w16_ptr, c_ptr, d_ptr, dtype):
    _check(lib().vh_op_fold_ln(w_ptr, b_ptr, gamma_ptr, beta_ptr, rows, dim, scale, w16_ptr, c_ptr, d_ptr, dtype, None))


def bench_gemm(M, N, K, epilogue, dtype=DTYPE_BF16, variant=0, iters=20, device=0):
    """Average launch time in ms of one GEMM shape (synthetic operands generated in HBM)."""
    ms = C.c_double(0)
    _check(lib().vh_bench_gemm(device, M, N, K, epilogue, dtype, variant, iters, C.byref(ms)))
    return ms.value


def op_layernorm(x_ptr, rows, dim, row_stride, gamma_ptr, beta_ptr, eps, out_ptr, dtype):
    _check(lib().vh_op_layernorm(x_ptr, rows, dim, row_stride, gamma_ptr, beta_ptr, eps, out_ptr, dtype, None))


def op_attention(qkv_ptr, batch, tokens, heads, out_ptr, dtype):
    _check(lib().vh_op_attention(qkv_ptr, batch, tokens, heads, out_ptr, dtype, None))


def op_im2col(in_ptr, batch, image, patch, channels, out_ptr, dtype):
    _check(lib().vh_op_im2col(in_ptr, batch, image, patch, channels, out_ptr, dtype, None))


def op_cast(in_ptr, out_ptr, n, dtype):
    _check(lib().vh_op_cast(in_ptr, out_ptr, n, dtype, None))


def op_fill(out_ptr, n, seed, tensor_id, kind, sigma=0.0):
    _check(lib().vh_op_fill(out_ptr, n, seed, tensor_id, kind, sigma, None))
