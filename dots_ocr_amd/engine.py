"""ctypes binding of include/dots_ocr_hip.h — the only way Python reaches the HIP kernels.

No torch types cross this boundary: numpy arrays for host buffers, integers for device pointers
(torch CUDA tensors are passed by ``data_ptr()``; torch and this library share one HIP runtime).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib
from .config import DotsConfig

EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, EPI_GELU, EPI_F32 = range(5)
DTYPE_BF16, DTYPE_F32, DTYPE_F16 = 0, 1, 2


class CDotsConfig(C.Structure):
    _fields_ = [
        ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("intermediate_size", C.c_int32), ("vocab_size", C.c_int32),
        ("rope_theta", C.c_float), ("rms_norm_eps", C.c_float),
        ("attention_bias", C.c_int32), ("image_token_id", C.c_int32),
        ("v_embed_dim", C.c_int32), ("v_layers", C.c_int32), ("v_heads", C.c_int32), ("v_intermediate", C.c_int32),
        ("v_patch", C.c_int32), ("v_merge", C.c_int32), ("v_channels", C.c_int32), ("v_temporal_patch", C.c_int32),
        ("v_rms_eps", C.c_float), ("v_ln_eps", C.c_float),
        ("v_use_bias", C.c_int32), ("v_post_norm", C.c_int32),
        ("max_batch", C.c_int32), ("max_seq_len", C.c_int32),
        ("max_patches", C.c_int64), ("max_prefill_tokens", C.c_int64), ("kv_pool_tokens", C.c_int64),
        ("fp8_weights", C.c_int32), ("_reserved", C.c_int32),
    ]


class CDotsStats(C.Structure):
    _fields_ = [
        ("vit_ms", C.c_float), ("prefill_ms", C.c_float), ("decode_ms", C.c_float), ("total_ms", C.c_float),
        ("vit_attn_ms", C.c_float), ("vit_attn_launches", C.c_int32),
        ("vit_gemm_ms", C.c_float), ("decode_steps", C.c_int32),
        ("vit_patches", C.c_int64), ("prefill_tokens", C.c_int64), ("new_tokens", C.c_int64),
        ("vit_attn_flops", C.c_double), ("vit_flops", C.c_double), ("prefill_flops", C.c_double),
        ("decode_bytes", C.c_double),
    ]


class DotsEngineError(RuntimeError):
    pass


_SIGNATURES_SET = False


def _prototypes(lib):
    global _SIGNATURES_SET
    if _SIGNATURES_SET:
        return
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    P = C.POINTER
    sig = {
        "dots_create": (i32, [P(CDotsConfig), i32, P(vp)]),
        "dots_destroy": (None, [vp]),
        "dots_last_error": (C.c_char_p, [vp]),
        "dots_stream": (vp, [vp]),
        "dots_load_weight": (i32, [vp, C.c_char_p, vp, i32, P(i64), i32]),
        "dots_finalize_weights": (i32, [vp]),
        "dots_vit_forward": (i32, [vp, vp, i32, i64, P(i64), i32, vp]),
        "dots_vit_prefetch": (i32, [vp, vp, i32, i64, P(i64), i32, i32]),
        "dots_vit_take_prefetched": (i32, [vp]),
        "dots_vit_prefetch_ready": (i32, [vp, P(i32)]),
        "dots_prefill": (i32, [vp, P(i32), P(i32), i32]),
        "dots_decode_step": (i32, [vp]),
        "dots_generate": (i32, [vp, P(i32), P(i32), i32, vp, i32, i64, P(i64), i32, i32, P(i32), i32, P(i32), P(i32)]),
        "dots_preprocess_image": (i32, [vp, vp, i32, i32, i32, i32, i32, P(i32), P(i32), i32, P(i32), P(i32), i32, P(f32), P(f32), f32, vp]),
        "dots_set_sampling": (i32, [vp, f32, f32, C.c_uint64]),
        "dots_set_decode_plan": (i32, [vp, i32]),
        "dots_set_gemm_plan": (i32, [vp, i32]),
        "dots_tower_tail": (i32, [vp, i32, P(i32)]),
        "dots_slot_capacity": (i32, [vp, i32, P(i32), P(i32)]),
        "dots_slots_reset": (i32, [vp]),
        "dots_set_eos": (i32, [vp, P(i32), i32]),
        "dots_slots_prefill": (i32, [vp, P(i32), i32, P(i32), P(i32), P(i32)]),
        "dots_slots_decode": (i32, [vp, i32]),
        "dots_slots_poll": (i32, [vp, P(i32), P(i32)]),
        "dots_slot_read": (i32, [vp, i32, P(i32), i32, P(i32)]),
        "dots_slot_release": (i32, [vp, i32]),
        "dots_kv_pool_info": (i32, [vp, P(i32), P(i32)]),
        "dots_get_logits": (i32, [vp, P(f32)]),
        "dots_set_next_tokens": (i32, [vp, P(i32), i32]),
        "dots_get_last_tokens": (i32, [vp, P(i32)]),
        "dots_get_stats": (i32, [vp, P(CDotsStats)]),
        "dots_synchronize": (i32, [vp]),
        "dots_debug_capture_hidden": (i32, [vp, i64]),
        "dots_debug_read_hidden": (i32, [vp, i32, i32, vp, P(i64)]),
        "dots_dev_alloc": (i32, [vp, i64, P(vp)]),
        "dots_dev_free": (i32, [vp, vp]),
        "dots_memcpy_h2d": (i32, [vp, vp, vp, i64]),
        "dots_memcpy_d2h": (i32, [vp, vp, vp, i64]),
        "dots_op_rmsnorm": (i32, [vp, vp, vp, vp, i64, i32, f32]),
        "dots_op_layernorm": (i32, [vp, vp, vp, vp, vp, i64, i32, f32]),
        "dots_op_gemm": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
        "dots_op_quant_fp8": (i32, [vp, vp, vp, i64, i32]),
        "dots_op_gemm_fp8": (i32, [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32]),
        "dots_op_flash_attn": (i32, [vp, vp, vp, vp, vp, P(i32), i32, i32, i32, i32, f32]),
        "dots_plan_flash_xcd": (i32, [P(i32), i32, i32, P(i32), P(i32), P(i64)]),
        "dots_op_qkv_rope_split": (i32, [vp, vp, vp, vp, vp, P(i32), i32, P(i32), i32, i32, i32, f32]),
        "dots_op_qkv_proj_rope": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, P(i32), i32, P(i32), i32, i32, i32, i32, f32, i32]),
        "dots_op_dec_qkv": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, f32, f32, i32]),
        "dots_op_decode_attn": (i32, [vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32]),
        "dots_op_dec_proj": (i32, [vp, vp, vp, vp, i32, i32, i32, i32]),
        "dots_op_dec_gateup": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32]),
        "dots_op_dec_lmhead": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32]),
        "dots_probe_mfma": (i32, [i32, vp, vp, vp, vp]),
        "dots_probe_grid_barrier": (i32, [i32, i32, i32, i32, i32, P(f32), P(i32)]),
        "dots_probe_cu_mask": (i32, [P(C.c_uint32), i32, i32, i32, i32, P(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _SIGNATURES_SET = True


EXPORTED_SYMBOLS = [
    "dots_create", "dots_destroy", "dots_last_error", "dots_stream", "dots_load_weight", "dots_finalize_weights",
    "dots_vit_forward", "dots_vit_prefetch", "dots_vit_take_prefetched", "dots_vit_prefetch_ready", "dots_preprocess_image", "dots_prefill", "dots_decode_step", "dots_generate", "dots_set_sampling", "dots_set_decode_plan", "dots_set_gemm_plan", "dots_tower_tail", "dots_get_logits",
    "dots_set_eos", "dots_slots_prefill", "dots_slots_decode", "dots_slots_poll", "dots_slot_read", "dots_slot_release", "dots_kv_pool_info", "dots_slot_capacity", "dots_slots_reset",
    "dots_set_next_tokens", "dots_get_last_tokens", "dots_get_stats", "dots_synchronize", "dots_debug_capture_hidden",
    "dots_debug_read_hidden", "dots_dev_alloc",
    "dots_dev_free", "dots_memcpy_h2d", "dots_memcpy_d2h", "dots_op_rmsnorm", "dots_op_layernorm", "dots_op_gemm", "dots_op_quant_fp8", "dots_op_gemm_fp8",
    "dots_op_flash_attn", "dots_plan_flash_xcd", "dots_op_qkv_rope_split", "dots_op_qkv_proj_rope", "dots_op_dec_qkv", "dots_op_decode_attn", "dots_op_dec_proj", "dots_op_dec_gateup",
    "dots_op_dec_lmhead", "dots_probe_mfma", "dots_probe_grid_barrier", "dots_probe_cu_mask",
]


def c_config(cfg: DotsConfig, max_batch: int, max_seq_len: int, max_patches: int, max_prefill_tokens: int, kv_pool_tokens: int = 0,
             fp8_weights: bool = False) -> CDotsConfig:
    v = cfg.vision
    return CDotsConfig(
        hidden_size=cfg.hidden_size, num_layers=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads,
        num_kv_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
        vocab_size=cfg.vocab_size, rope_theta=cfg.rope_theta, rms_norm_eps=cfg.rms_norm_eps,
        attention_bias=int(cfg.attention_bias), image_token_id=cfg.image_token_id,
        v_embed_dim=v.embed_dim, v_layers=v.num_hidden_layers, v_heads=v.num_attention_heads,
        v_intermediate=v.intermediate_size, v_patch=v.patch_size, v_merge=v.spatial_merge_size,
        v_channels=v.num_channels, v_temporal_patch=v.temporal_patch_size, v_rms_eps=v.rms_norm_eps,
        v_ln_eps=v.merger_ln_eps, v_use_bias=int(v.use_bias), v_post_norm=int(v.post_norm),
        max_batch=max_batch, max_seq_len=max_seq_len, max_patches=max_patches,
        max_prefill_tokens=max_prefill_tokens, kv_pool_tokens=kv_pool_tokens, fp8_weights=int(bool(fp8_weights)))


def plan_flash_xcd(lens: Sequence[int], heads: int):
    """Host-only: (base[8], cnt[8], cost[8], n_items) — how the flash-attention work list of a packed batch of sequences of `lens` patches
    is cut across the XCDs (dots_plan_flash_xcd: equal KV-tile cost per XCD).  Needs the library, not a GPU."""
    lib = _lib.load()
    _prototypes(lib)
    L = np.ascontiguousarray(lens, dtype=np.int32)
    base, cnt, cost = np.zeros(8, np.int32), np.zeros(8, np.int32), np.zeros(8, np.int64)
    n = lib.dots_plan_flash_xcd(_i32p(L), int(L.shape[0]), int(heads), _i32p(base), _i32p(cnt), _i64p(cost))
    if n < 0:
        raise DotsEngineError(f"dots_plan_flash_xcd failed ({n})")
    return base, cnt, cost, int(n)


def _i32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _i64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class Engine:
    """One GPU, one HIP stream, one model replica."""

    def __init__(self, cfg: DotsConfig, device: int = 0, max_batch: int = 8, max_seq_len: int = 8192,
                 max_patches: int = 8 * 19824 + 64, max_prefill_tokens: Optional[int] = None, kv_pool_tokens: int = 0,
                 fp8_weights: bool = False):
        self.lib = _lib.load()
        _prototypes(self.lib)
        self.cfg = cfg
        self.device = device
        self.max_batch = max_batch
        self.max_seq_len = max_seq_len
        self.max_patches = max_patches
        if max_prefill_tokens is None:
            max_prefill_tokens = max_batch * max_seq_len
        self.max_prefill_tokens = max_prefill_tokens
        self.kv_pool_tokens = kv_pool_tokens
        self.fp8_weights = bool(fp8_weights)
        self._cc = c_config(cfg, max_batch, max_seq_len, max_patches, max_prefill_tokens, kv_pool_tokens, fp8_weights)
        h = C.c_void_p()
        rc = self.lib.dots_create(C.byref(self._cc), device, C.byref(h))
        if rc != 0:
            raise DotsEngineError(f"dots_create failed ({rc}): {self.lib.dots_last_error(None).decode()}")
        self.h = h

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc: int, what: str):
        if rc != 0:
            raise DotsEngineError(f"{what} failed ({rc}): {self.lib.dots_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.dots_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return int(self.lib.dots_stream(self.h) or 0)

    def synchronize(self):
        self._ck(self.lib.dots_synchronize(self.h), "dots_synchronize")

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, "torch.Tensor"]):  # noqa: F821 (torch only used by the caller)
        import torch
        dt = {torch.bfloat16: DTYPE_BF16, torch.float32: DTYPE_F32, torch.float16: DTYPE_F16}
        for name, t in sd.items():
            t = t.detach().cpu().contiguous()
            if t.dtype not in dt:
                t = t.float()
            shape = (C.c_int64 * t.dim())(*t.shape)
            self._ck(self.lib.dots_load_weight(self.h, name.encode(), C.c_void_p(t.data_ptr()), dt[t.dtype], shape, t.dim()),
                     f"dots_load_weight({name})")
        self._ck(self.lib.dots_finalize_weights(self.h), "dots_finalize_weights")

    # ------------------------------------------------------------------ device memory helpers
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._ck(self.lib.dots_dev_alloc(self.h, nbytes, C.byref(p)), "dots_dev_alloc")
        return int(p.value)

    def dev_free(self, ptr: int):
        self._ck(self.lib.dots_dev_free(self.h, C.c_void_p(ptr)), "dots_dev_free")

    def to_device(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        p = self.dev_alloc(a.nbytes)
        self._ck(self.lib.dots_memcpy_h2d(self.h, C.c_void_p(p), a.ctypes.data_as(C.c_void_p), a.nbytes), "dots_memcpy_h2d")
        return p

    def copy_to_device(self, ptr: int, a: np.ndarray):
        """host array -> an existing device buffer (blocking)"""
        a = np.ascontiguousarray(a)
        self._ck(self.lib.dots_memcpy_h2d(self.h, C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes), "dots_memcpy_h2d")

    def to_host(self, ptr: int, shape: Sequence[int], dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._ck(self.lib.dots_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes), "dots_memcpy_d2h")
        return out

    # ------------------------------------------------------------------ hot path
    def vit_forward(self, pixel_values, grid_thw: np.ndarray, on_device: bool = False, out_dev: int = 0) -> int:
        """pixel_values: np.float32 [N, patch_dim] (host) or a device pointer when on_device."""
        grid = np.ascontiguousarray(grid_thw, dtype=np.int64)
        n = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
        if on_device:
            ptr = C.c_void_p(int(pixel_values))
        else:
            pv = np.ascontiguousarray(pixel_values, dtype=np.float32)
            assert pv.shape[0] == n, (pv.shape, n)
            ptr = pv.ctypes.data_as(C.c_void_p)
        self._ck(self.lib.dots_vit_forward(self.h, ptr, int(on_device), n, _i64p(grid), grid.shape[0],
                                           C.c_void_p(out_dev) if out_dev else None), "dots_vit_forward")
        return n // (self.cfg.vision.spatial_merge_size ** 2)

    def vit_prefetch(self, pixel_values, grid_thw: np.ndarray, on_device: bool = False, after_prefill: bool = False) -> int:
        """The tower of the NEXT page batch, asynchronously on the CU-masked side stream (include/dots_ocr_hip.h "Software
        pipelining"); pixel_values must stay alive until vit_take().  after_prefill: launch it behind the next prefill."""
        grid = np.ascontiguousarray(grid_thw, dtype=np.int64)
        n = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
        if on_device:
            ptr = C.c_void_p(int(pixel_values))
        else:
            self._pref_keep = np.ascontiguousarray(pixel_values, dtype=np.float32)
            assert self._pref_keep.shape[0] == n, (self._pref_keep.shape, n)
            ptr = self._pref_keep.ctypes.data_as(C.c_void_p)
        self._ck(self.lib.dots_vit_prefetch(self.h, ptr, int(on_device), n, _i64p(grid), grid.shape[0], int(after_prefill)), "dots_vit_prefetch")
        return n // (self.cfg.vision.spatial_merge_size ** 2)

    def vit_take(self):
        """Put the prefetched vision rows in place for the next prefill / generate(..., vision_taken=True)."""
        self._ck(self.lib.dots_vit_take_prefetched(self.h), "dots_vit_take_prefetched")

    def vit_ready(self) -> bool:
        """True when the prefetched tower has finished: vit_take() then makes nothing wait (a serving loop keeps decoding until then)."""
        r = C.c_int32(0)
        self._ck(self.lib.dots_vit_prefetch_ready(self.h, C.byref(r)), "dots_vit_prefetch_ready")
        return bool(r.value)

    def preprocess_image(self, rgb, out_dev: int, min_pixels: Optional[int] = None, max_pixels: Optional[int] = None,
                         shape: Optional[Sequence[int]] = None):
        """uint8 [h, w, 3] image -> float32 patches written at device pointer `out_dev`; returns [t, gh, gw].
        `rgb` is a host numpy array, or (with shape=(h, w)) a device pointer to the uint8 pixels already in HBM.
        Bit-identical to image_utils.preprocess_image (Pillow BICUBIC + normalise + patchify), computed on the GPU."""
        from .image_utils import bicubic_resample_tables, smart_resize
        on_device = shape is not None
        if on_device:
            h, w = int(shape[0]), int(shape[1])
            rgb_ptr = C.c_void_p(int(rgb))
        else:
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            h, w, _ = rgb.shape
            rgb_ptr = rgb.ctypes.data_as(C.c_void_p)
        v = self.cfg.vision
        rh, rw = smart_resize(h, w, v.patch_size * v.spatial_merge_size, min_pixels or self.cfg.min_pixels, max_pixels or self.cfg.max_pixels)
        hc = hb = vc = vb = None
        hk = vk = 0
        if rw != w:
            hc, hb = bicubic_resample_tables(w, rw)
            hk = hc.shape[1]
        if rh != h:
            vc, vb = bicubic_resample_tables(h, rh)
            vk = vc.shape[1]
        mean = np.asarray(self.cfg.image_mean, np.float32)
        std = np.asarray(self.cfg.image_std, np.float32)
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        self._ck(self.lib.dots_preprocess_image(
            self.h, rgb_ptr, int(on_device), h, w, rh, rw,
            _i32p(hc) if hc is not None else None, _i32p(hb) if hb is not None else None, hk,
            _i32p(vc) if vc is not None else None, _i32p(vb) if vb is not None else None, vk,
            fp(mean), fp(std), float(np.float32(1.0 / 255.0)), C.c_void_p(out_dev)), "dots_preprocess_image")
        return [1, rh // v.patch_size, rw // v.patch_size]

    def prefill(self, input_ids: np.ndarray, prompt_lens: np.ndarray):
        ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        lens = np.ascontiguousarray(prompt_lens, dtype=np.int32)
        assert ids.shape[0] == int(lens.sum())
        self._ck(self.lib.dots_prefill(self.h, _i32p(ids), _i32p(lens), lens.shape[0]), "dots_prefill")
        self._B = int(lens.shape[0])

    def set_sampling(self, temperature: float = 0.0, top_p: float = 1.0, seed: int = 0):
        """temperature 0 = greedy; otherwise softmax(logits/T) restricted to the top_p nucleus, reproducible from seed."""
        self._ck(self.lib.dots_set_sampling(self.h, float(temperature), float(top_p), int(seed) & (2 ** 64 - 1)), "dots_set_sampling")

    def set_decode_plan(self, plan: int):
        """0 = launch plan by stream (whole chip / CU partition beside a prefetched tower), 1 = the partition plan (whole-tile projections,
        pair-walking gate|up) on every step; + 2 = streaming decode attention wherever legal, + 4 = per-split decode attention (default: by
        items per CU); bit-identical results."""
        self._ck(self.lib.dots_set_decode_plan(self.h, int(plan)), "dots_set_decode_plan")

    def set_gemm_plan(self, plan: int):
        """0 = 8-wave ping-pong GEMM, 1 = one wave per SIMD (round 5); process-wide, bit-identical results."""
        self._ck(self.lib.dots_set_gemm_plan(self.h, int(plan)), "dots_set_gemm_plan")

    def tower_tail(self, set: int = -2) -> int:
        """Blocks of a prefetched tower that run on the whole chip instead of the tower's CU partition: set >= 0 fixed (0 = off, the engine's
        DEFAULT), -1 adaptive (opt-in: sized from the previous launch's events so that the partition part ends with the decode loop — the rule
        reads "the host stopped issuing decode chunks" as "the decode loop drained", which holds for bench.py's closed a4 pipeline and NOT for a
        serving loop that keeps slots occupied: do not copy it into one), -2 = query only (this method's default argument).
        Returns the tail of the tower launched last."""
        now = C.c_int32(0)
        self._ck(self.lib.dots_tower_tail(self.h, int(set), C.byref(now)), "dots_tower_tail")
        return int(now.value)

    def decode_step(self):
        self._ck(self.lib.dots_decode_step(self.h), "dots_decode_step")

    # ------------------------------------------------------------------ continuous batching (sequence slots)
    def set_eos(self, eos_ids: Sequence[int]):
        eos = np.ascontiguousarray(list(eos_ids), dtype=np.int32)
        self._ck(self.lib.dots_set_eos(self.h, _i32p(eos) if len(eos) else None, len(eos)), "dots_set_eos")

    def slots_prefill(self, slots: Sequence[int], input_ids: np.ndarray, prompt_lens: Sequence[int], max_new_tokens: Sequence[int]):
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        lens = np.ascontiguousarray(prompt_lens, dtype=np.int32)
        cap = np.ascontiguousarray(max_new_tokens, dtype=np.int32)
        assert sl.shape == lens.shape == cap.shape and ids.shape[0] == int(lens.sum())
        self._ck(self.lib.dots_slots_prefill(self.h, _i32p(sl), sl.shape[0], _i32p(ids), _i32p(lens), _i32p(cap)), "dots_slots_prefill")

    def slots_reset(self):
        """Slot mode, every slot free, every KV page in the pool."""
        self._ck(self.lib.dots_slots_reset(self.h), "dots_slots_reset")

    def slots_decode(self, n_steps: int):
        self._ck(self.lib.dots_slots_decode(self.h, int(n_steps)), "dots_slots_decode")

    def slots_poll(self):
        """-> (finished [max_batch]: -1 free / 0 running / 1 finished, out_lens [max_batch])"""
        fin = np.empty((self.max_batch,), dtype=np.int32)
        lens = np.empty((self.max_batch,), dtype=np.int32)
        self._ck(self.lib.dots_slots_poll(self.h, _i32p(fin), _i32p(lens)), "dots_slots_poll")
        return fin, lens

    def slot_read(self, slot: int, capacity: int) -> np.ndarray:
        out = np.empty((max(1, capacity),), dtype=np.int32)
        n = C.c_int32(0)
        self._ck(self.lib.dots_slot_read(self.h, int(slot), _i32p(out), int(capacity), C.byref(n)), "dots_slot_read")
        return out[:min(int(n.value), capacity)].copy()

    def slot_release(self, slot: int):
        self._ck(self.lib.dots_slot_release(self.h, int(slot)), "dots_slot_release")

    def slot_capacity(self, slot: int):
        """(pages owned, limit on prompt + generated tokens) of an occupied slot; the limit is prompt + max_new_tokens unless the KV
        pool ran dry while the sequence was growing."""
        pg, lim = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.dots_slot_capacity(self.h, int(slot), C.byref(pg), C.byref(lim)), "dots_slot_capacity")
        return pg.value, lim.value

    def kv_pool_info(self):
        """(total, free) pages of 64 tokens in the paged KV pool."""
        tot, free = C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.dots_kv_pool_info(self.h, C.byref(tot), C.byref(free)), "dots_kv_pool_info")
        return int(tot.value), int(free.value)

    def get_logits(self) -> np.ndarray:
        out = np.empty((self._B, self.cfg.vocab_size), dtype=np.float32)
        self._ck(self.lib.dots_get_logits(self.h, out.ctypes.data_as(C.POINTER(C.c_float))), "dots_get_logits")
        return out

    def get_last_tokens(self) -> np.ndarray:
        out = np.empty((self._B,), dtype=np.int32)
        self._ck(self.lib.dots_get_last_tokens(self.h, _i32p(out)), "dots_get_last_tokens")
        return out

    def set_next_tokens(self, tokens: Sequence[int]):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        self._ck(self.lib.dots_set_next_tokens(self.h, _i32p(t), t.shape[0]), "dots_set_next_tokens")

    def generate(self, input_ids: np.ndarray, prompt_lens: np.ndarray, pixel_values=None,
                 grid_thw: Optional[np.ndarray] = None, max_new_tokens: int = 128, eos_ids: Sequence[int] = (),
                 pixel_on_device: bool = False, vision_taken: bool = False):
        """Packed prompts + packed patches -> (out_ids [B, max_new_tokens] int32, out_lens [B]).
        vision_taken: the vision rows were prefetched and taken (vit_prefetch / vit_take); pixel_values / grid_thw are ignored."""
        ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        lens = np.ascontiguousarray(prompt_lens, dtype=np.int32)
        B = int(lens.shape[0])
        assert ids.shape[0] == int(lens.sum())
        out_ids = np.zeros((B, max_new_tokens), dtype=np.int32)
        out_lens = np.zeros((B,), dtype=np.int32)
        eos = np.ascontiguousarray(list(eos_ids), dtype=np.int32)
        if vision_taken:
            ptr, n, gp, n_img = None, 0, None, -1
        elif grid_thw is not None and len(grid_thw):
            grid = np.ascontiguousarray(grid_thw, dtype=np.int64)
            n = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
            if pixel_on_device:
                ptr = C.c_void_p(int(pixel_values))
            else:
                pv = np.ascontiguousarray(pixel_values, dtype=np.float32)
                assert pv.shape[0] == n
                ptr = pv.ctypes.data_as(C.c_void_p)
            gp, n_img = _i64p(grid), grid.shape[0]
        else:
            ptr, n, gp, n_img = None, 0, None, 0
        self._ck(self.lib.dots_generate(self.h, _i32p(ids), _i32p(lens), B, ptr, int(pixel_on_device), n, gp, n_img,
                                        max_new_tokens, _i32p(eos) if len(eos) else None, len(eos),
                                        _i32p(out_ids), _i32p(out_lens)), "dots_generate")
        self._B = B
        return out_ids, out_lens

    def capture_hidden(self, capacity_elems: int):
        """Debug: keep the residual stream after every ViT block / LM prefill layer of the next calls (0 = off)."""
        self._ck(self.lib.dots_debug_capture_hidden(self.h, int(capacity_elems)), "dots_debug_capture_hidden")

    def read_hidden(self, which: str, layer: int) -> np.ndarray:
        """which = "vit" | "lm" -> uint16 (raw bf16) [rows, dim] of that layer's output."""
        dim = self.cfg.vision.embed_dim if which == "vit" else self.cfg.hidden_size
        cap = (self.max_patches if which == "vit" else self.max_prefill_tokens) * dim
        buf = np.empty(cap, dtype=np.uint16)
        rows = C.c_int64(0)
        self._ck(self.lib.dots_debug_read_hidden(self.h, 0 if which == "vit" else 1, int(layer), buf.ctypes.data_as(C.c_void_p), C.byref(rows)),
                 "dots_debug_read_hidden")
        return buf[: rows.value * dim].reshape(rows.value, dim).copy()

    def stats(self) -> dict:
        st = CDotsStats()
        self._ck(self.lib.dots_get_stats(self.h, C.byref(st)), "dots_get_stats")
        return {k: getattr(st, k) for k, _ in CDotsStats._fields_}

    # ------------------------------------------------------------------ single kernels (device pointers)
    def op_rmsnorm(self, x, w, y, rows, dim, eps):
        self._ck(self.lib.dots_op_rmsnorm(self.h, x, w, y, rows, dim, eps), "dots_op_rmsnorm")

    def op_layernorm(self, x, w, b, y, rows, dim, eps):
        self._ck(self.lib.dots_op_layernorm(self.h, x, w, b, y, rows, dim, eps), "dots_op_layernorm")

    def op_gemm(self, A, W, bias, residual, Cout, M, N, K, epilogue=EPI_NONE, colscale=None):
        self._ck(self.lib.dots_op_gemm(self.h, A, W, bias or None, residual or None, Cout, M, N, K, epilogue, colscale or None), "dots_op_gemm")

    def op_gemm_fp8(self, A, W, bias, residual, Cout, M, N, K, epilogue=EPI_NONE):
        self._ck(self.lib.dots_op_gemm_fp8(self.h, A, W, bias or None, residual or None, Cout, M, N, K, epilogue), "dots_op_gemm_fp8")

    def op_quant_fp8(self, w_inout, scale_out, N, K):
        self._ck(self.lib.dots_op_quant_fp8(self.h, w_inout, scale_out, N, K), "dots_op_quant_fp8")

    def op_flash_attn(self, q, k, vt, out, cu_seqlens, Hq, Hkv, causal, scale):
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        self._ck(self.lib.dots_op_flash_attn(self.h, q, k, vt, out, _i32p(cu), cu.shape[0] - 1, Hq, Hkv, int(causal), scale),
                 "dots_op_flash_attn")

    def op_qkv_rope_split(self, qkv, q, k, vt, cu_seqlens, pos, Hq, Hkv, rope2d, theta):
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        self._ck(self.lib.dots_op_qkv_rope_split(self.h, qkv, q, k, vt, _i32p(cu), cu.shape[0] - 1, _i32p(pos), Hq, Hkv,
                                                 int(rope2d), theta), "dots_op_qkv_rope_split")

    def op_qkv_proj_rope(self, x, w, bias, qkv_ws, q, k, vt, cu_seqlens, pos, K, Hq, Hkv, rope2d, theta, fused):
        cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        self._ck(self.lib.dots_op_qkv_proj_rope(self.h, x, w, bias, qkv_ws, q, k, vt, _i32p(cu), cu.shape[0] - 1, _i32p(pos), K, Hq, Hkv,
                                                int(rope2d), theta, int(fused)), "dots_op_qkv_proj_rope")

    # ---- single kernels of the decode step (row-major device tensors; packing happens inside the library)
    def op_dec_qkv(self, h, ln_w, wqkv, bias, ctx_len, block_table, max_pages, pool_layer, q_out, B, H, Hq, Hkv, eps, rope_theta, fp8=False):
        self._ck(self.lib.dots_op_dec_qkv(self.h, h, ln_w, wqkv, bias or None, ctx_len, block_table, max_pages, pool_layer, q_out,
                                          B, H, Hq, Hkv, eps, rope_theta, int(fp8)), "dots_op_dec_qkv")

    def op_decode_attn(self, q, pool_layer, ctx_len, block_table, max_pages, out, B, Hq, Hkv, max_seq_len):
        self._ck(self.lib.dots_op_decode_attn(self.h, q, pool_layer, ctx_len, block_table, max_pages, out, B, Hq, Hkv, max_seq_len),
                 "dots_op_decode_attn")

    def op_dec_proj(self, x, w, h_inout, B, N, K, fp8=False):
        self._ck(self.lib.dots_op_dec_proj(self.h, x, w, h_inout, B, N, K, int(fp8)), "dots_op_dec_proj")

    def op_dec_gateup(self, h, ln_w, gate_w, up_w, act_out, B, H, I, eps, fp8=False):
        self._ck(self.lib.dots_op_dec_gateup(self.h, h, ln_w, gate_w, up_w, act_out, B, H, I, eps, int(fp8)), "dots_op_dec_gateup")

    def op_dec_lmhead(self, h, ln_w, w, logits_out, B, H, V, eps, fp8=False):
        self._ck(self.lib.dots_op_dec_lmhead(self.h, h, ln_w, w, logits_out, B, H, V, eps, int(fp8)), "dots_op_dec_lmhead")
