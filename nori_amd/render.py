"""`Renderer`: Python handle on a libnori_hip context (one GPU).

Thin glue over the C ABI of include/nori_hip.h: numpy in / numpy out for the
operator-level twins, torch CUDA tensors (device memory, streams) for the
frame buffer of the hot path.  No compute happens in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _capi as capi
from ._capi import NoriError, ptr
from .scene import Bsdf, Scene


def _f32(a, shape_last=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape_last is not None:
        a = a.reshape(-1, shape_last)
    return a


class Renderer:
    """Owns a `nori_hip_ctx`.  Mirrors Scene/Accel/Integrator of the reference:
    `upload` = Scene::addChild + activate, `build_accel` = Accel::build,
    `intersect` = Accel::rayIntersect, `li` = Integrator::Li, `render` =
    render() of src/main.cpp:58-119."""

    def __init__(self, device: int = 0):
        self._lib = capi.load_hip()
        h = C.c_void_p()
        rc = self._lib.nori_hip_create(int(device), C.byref(h))
        if rc != 0:
            msg = self._lib.nori_hip_last_error(None)
            raise NoriError(f"nori_hip_create({device}) failed: {capi.STATUS.get(rc, rc)}: "
                            f"{msg.decode() if msg else ''}")
        self._h = h
        self.device = int(device)
        self.scene: Optional[Scene] = None

    # -------------------------------------------------------------- plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.nori_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.nori_hip_last_error(self._h)
            raise NoriError(f"{what}: {capi.STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def excursions(self, reset: bool = False):
        """(rcp, div, sqrt operands outside the verified domain, fallbacks taken) -- libnori_hip_count.so only."""
        out = (C.c_ulonglong * 4)()
        self._check(self._lib.nori_hip_debug_excursions(self._h, out, int(reset)), "debug_excursions")
        return tuple(int(v) for v in out)

    # ------------------------------------------------------------ load time
    def upload(self, scene: Scene, build: bool = True, builder: int = 0):
        """builder: nori_accel_builder.  The default here is 0, the host's SAH builder, whose trees the suite's fixed expectations (depths,
        costs, which traversal-stack variant runs) were written for; the library's own default -- NORI_ACCEL_AUTO = 2, what the C++ host,
        bench.py, smoke() and the tools pass -- is the device's builder (lbvh.hip)."""
        desc, keep = scene.c_desc()
        self._check(self._lib.nori_hip_upload_scene(self._h, C.byref(desc)), "upload_scene")
        del keep
        self.scene = scene
        if build:
            self.build_accel(builder)
        return self

    def build_accel(self, builder: int = 0):
        self._check(self._lib.nori_hip_build_accel(self._h, int(builder)), "build_accel")

    def set_option(self, key: str, value) -> None:
        self._check(self._lib.nori_hip_set_option(self._h, key.encode(), str(value).encode()), f"set_option({key})")

    def accel_info(self) -> dict:
        info = capi.AccelInfo()
        self._check(self._lib.nori_hip_accel_info(self._h, C.byref(info)), "accel_info")
        return info.as_dict()

    @property
    def border(self) -> int:
        b = self._lib.nori_hip_border_size(self._h)
        if b < 0:
            self._check(b, "border_size")
        return b

    def frame_shape(self):
        b = self.border
        c = self.scene.camera
        return (c.height + 2 * b, c.width + 2 * b, 4)

    # ---------------------------------------------------- operator-level twins
    def intersect(self, rays: np.ndarray, shadow: bool = False) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=capi.RAY_DTYPE)
        its = np.zeros(rays.shape[0], dtype=capi.ITS_DTYPE)
        self._check(self._lib.nori_hip_intersect(self._h, ptr(rays), ptr(its), rays.shape[0], int(shadow)), "intersect")
        return its

    def sample_rays(self, pixel_samples) -> np.ndarray:
        ps = _f32(pixel_samples, 2)
        rays = np.zeros(ps.shape[0], dtype=capi.RAY_DTYPE)
        self._check(self._lib.nori_hip_sample_rays(self._h, ptr(ps), ps.shape[0], ptr(rays)), "sample_rays")
        return rays

    def li(self, rays, seed_state, seed_seq) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=capi.RAY_DTYPE)
        ss = np.ascontiguousarray(seed_state, dtype=np.uint64)
        sq = np.ascontiguousarray(seed_seq, dtype=np.uint64)
        out = np.zeros((rays.shape[0], 3), dtype=np.float32)
        self._check(self._lib.nori_hip_li(self._h, ptr(rays), rays.shape[0], ptr(ss), ptr(sq), ptr(out)), "li")
        return out

    def bsdf_sample(self, bsdf: Bsdf, wi, sample):
        wi, sample = _f32(wi, 3), _f32(sample, 2)
        n = wi.shape[0]
        wo, w = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        eta, meas = np.zeros(n, np.float32), np.zeros(n, np.int32)
        d = bsdf.desc()
        self._check(self._lib.nori_hip_bsdf_sample(self._h, C.byref(d), ptr(wi), ptr(sample), n, ptr(wo), ptr(w),
                                                   ptr(eta), ptr(meas)), "bsdf_sample")
        return wo, w, eta, meas

    def bsdf_eval(self, bsdf: Bsdf, wi, wo):
        wi, wo = _f32(wi, 3), _f32(wo, 3)
        out = np.zeros((wi.shape[0], 3), np.float32)
        d = bsdf.desc()
        self._check(self._lib.nori_hip_bsdf_eval(self._h, C.byref(d), ptr(wi), ptr(wo), wi.shape[0], ptr(out)), "bsdf_eval")
        return out

    def bsdf_pdf(self, bsdf: Bsdf, wi, wo):
        wi, wo = _f32(wi, 3), _f32(wo, 3)
        out = np.zeros(wi.shape[0], np.float32)
        d = bsdf.desc()
        self._check(self._lib.nori_hip_bsdf_pdf(self._h, C.byref(d), ptr(wi), ptr(wo), wi.shape[0], ptr(out)), "bsdf_pdf")
        return out

    def warp(self, name: str, sample, param: float = 0.0):
        s = _f32(sample, 2)
        out = np.zeros((s.shape[0], 3), np.float32)
        self._check(self._lib.nori_hip_warp(self._h, capi.WARP_NAMES[name], float(param), ptr(s), s.shape[0], ptr(out)), "warp")
        return out

    def warp_pdf(self, name: str, points, param: float = 0.0):
        p = _f32(points, 3)
        out = np.zeros(p.shape[0], np.float32)
        self._check(self._lib.nori_hip_warp_pdf(self._h, capi.WARP_NAMES[name], float(param), ptr(p), p.shape[0], ptr(out)), "warp_pdf")
        return out

    def pcg32_floats(self, seed_state, seed_seq, count: int):
        ss = np.ascontiguousarray(seed_state, dtype=np.uint64)
        sq = np.ascontiguousarray(seed_seq, dtype=np.uint64)
        out = np.zeros((ss.shape[0], count), np.float32)
        self._check(self._lib.nori_hip_pcg32_floats(self._h, ptr(ss), ptr(sq), ss.shape[0], count, ptr(out)), "pcg32_floats")
        return out

    def splat(self, positions, values, rgbw: Optional[np.ndarray] = None):
        p, v = _f32(positions, 2), _f32(values, 3)
        if rgbw is None:
            rgbw = np.zeros(self.frame_shape(), np.float32)
        rgbw = np.ascontiguousarray(rgbw, dtype=np.float32)
        self._check(self._lib.nori_hip_splat(self._h, ptr(p), ptr(v), p.shape[0], ptr(rgbw)), "splat")
        return rgbw

    # ------------------------------------------------------------ hot path
    @staticmethod
    def _params(spp_begin, spp_count, tile_mod, tile_rem, count_traversal, stream, time_kernels=False, seed_mode=capi.SEED_PER_SAMPLE):
        p = capi.RenderParams()
        p.spp_begin, p.spp_count = int(spp_begin), int(spp_count)
        p.tile_mod, p.tile_rem = int(tile_mod), int(tile_rem)
        p.seed_mode = int(seed_mode)
        p.count_traversal = int(bool(count_traversal))
        p.time_kernels = int(bool(time_kernels))
        p.stream = stream
        return p

    def render_host(self, spp_count=None, spp_begin=0, tile_mod=1, tile_rem=0, count_traversal=False, seed_mode=capi.SEED_PER_SAMPLE):
        """Render into a fresh host RGBW frame; returns (rgbw, stats dict).  seed_mode: capi.SEED_PER_SAMPLE (default)
        or capi.SEED_NORI_BLOCK -- the reference's one-stream-per-32x32-block sampler (src/independent.cpp:36-41)."""
        spp = self.scene.sample_count if spp_count is None else spp_count
        p = self._params(spp_begin, spp, tile_mod, tile_rem, count_traversal, None, seed_mode=seed_mode)
        rgbw = np.zeros(self.frame_shape(), np.float32)
        st = capi.RenderStats()
        self._check(self._lib.nori_hip_render_host(self._h, C.byref(p), ptr(rgbw), C.byref(st)), "render_host")
        return rgbw, st.as_dict()

    def render_into(self, rgbw_tensor, spp_count=None, spp_begin=0, tile_mod=1, tile_rem=0,
                    count_traversal=False, stream=None, want_stats=True, time_kernels=False):
        """Accumulate into a torch CUDA float32 tensor of shape frame_shape().

        `stream`: a torch.cuda.Stream (its raw hipStream_t is handed to the
        library) or None for the legacy default stream."""
        assert rgbw_tensor.is_cuda and rgbw_tensor.is_contiguous() and tuple(rgbw_tensor.shape) == tuple(self.frame_shape())
        spp = self.scene.sample_count if spp_count is None else spp_count
        raw = C.c_void_p(stream.cuda_stream) if stream is not None else None
        p = self._params(spp_begin, spp, tile_mod, tile_rem, count_traversal, raw, time_kernels)
        st = capi.RenderStats() if want_stats else None
        self._check(self._lib.nori_hip_render(self._h, C.byref(p), C.c_void_p(rgbw_tensor.data_ptr()),
                                              C.byref(st) if st is not None else None), "render")
        return st.as_dict() if st is not None else None

    # -- film_order = reference shared out: rows of 32x32 blocks (include/nori_hip.h: nori_hip_render_block_rows) --
    def block_rows(self) -> int:
        """rows of 32x32 blocks (NORI_BLOCK_SIZE) in the frame"""
        return (self.scene.camera.height + 31) // 32

    def block_acc_floats(self) -> int:
        n = C.c_size_t(0)
        self._check(self._lib.nori_hip_block_acc_floats(self._h, C.byref(n)), "block_acc_floats")
        return int(n.value)

    def render_block_rows_into(self, acc_tensor, row_begin, row_count, spp_count=None, spp_begin=0, stream=None, count_traversal=False):
        """Block rows [row_begin, row_begin + row_count) in the reference's summation order: their blocks' accumulators are written
        into `acc_tensor` (CUDA float32, block_acc_floats() elements, zeroed by the caller; other blocks untouched)."""
        assert acc_tensor.is_cuda and acc_tensor.is_contiguous() and acc_tensor.numel() == self.block_acc_floats()
        spp = self.scene.sample_count if spp_count is None else spp_count
        raw = C.c_void_p(stream.cuda_stream) if stream is not None else None
        p = self._params(spp_begin, spp, 1, 0, count_traversal, raw, False)
        st = capi.RenderStats()
        self._check(self._lib.nori_hip_render_block_rows(self._h, C.byref(p), int(row_begin), int(row_count), C.c_void_p(acc_tensor.data_ptr()), C.byref(st)), "render_block_rows")
        return st.as_dict()

    def resolve_blocks(self, acc_tensor, rgbw_tensor, stream=None):
        """ImageBlock::put(ImageBlock&) of every block in BlockGenerator's order: adds into the RGBW frame tensor."""
        assert acc_tensor.is_cuda and acc_tensor.numel() == self.block_acc_floats()
        assert rgbw_tensor.is_cuda and rgbw_tensor.is_contiguous() and tuple(rgbw_tensor.shape) == tuple(self.frame_shape())
        raw = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._check(self._lib.nori_hip_resolve_blocks(self._h, C.c_void_p(acc_tensor.data_ptr()), C.c_void_p(rgbw_tensor.data_ptr()), raw), "resolve_blocks")
        return rgbw_tensor

    def develop(self, rgbw_tensor, stream=None):
        """ImageBlock::toBitmap on the device: (H, W, 3) torch tensor."""
        import torch
        c = self.scene.camera
        rgb = torch.empty((c.height, c.width, 3), dtype=torch.float32, device=rgbw_tensor.device)
        raw = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._check(self._lib.nori_hip_develop(self._h, C.c_void_p(rgbw_tensor.data_ptr()), C.c_void_p(rgb.data_ptr()), raw), "develop")
        return rgb


class DeviceGroup:
    """All GPUs of a node behind one call (nori_hip_group_*, include/nori_hip.h): the render loop of src/main.cpp:78-119
    with the image blocks shared out over devices instead of TBB workers and ImageBlock::put(ImageBlock&) as one merge."""
    SPLIT = {"tile": 0, "sample": 1}
    MERGE = {"reduce": 0, "gather": 1}

    def __init__(self, devices):
        self._lib = capi.load_hip()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._lib.nori_hip_group_create(devs, len(devices), C.byref(h))
        if rc != 0:
            msg = self._lib.nori_hip_group_last_error(None)
            raise NoriError(f"nori_hip_group_create({list(devices)}) failed: {capi.STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")
        self._h = h
        self.scene: Optional[Scene] = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.nori_hip_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.nori_hip_group_last_error(self._h)
            raise NoriError(f"{what}: {capi.STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")

    @property
    def size(self) -> int:
        return int(self._lib.nori_hip_group_size(self._h))

    @property
    def transport(self) -> str:
        return self._lib.nori_hip_group_transport(self._h).decode()

    @property
    def warning(self) -> str:
        """"" or why the merge runs over peer copies although RCCL was wanted."""
        return self._lib.nori_hip_group_warning(self._h).decode()

    def engines(self):
        """nori_render_stats::engine of every device's share of the last frame."""
        out = (C.c_uint32 * self.size)()
        n = self._lib.nori_hip_group_engines(self._h, out, self.size)
        return [int(out[k]) for k in range(max(n, 0))]

    def upload(self, scene: Scene, builder: int = 0):
        desc, keep = scene.c_desc()
        self._check(self._lib.nori_hip_group_upload_scene(self._h, C.byref(desc), int(builder)), "group_upload_scene")
        del keep
        self.scene = scene
        return self

    def set_option(self, key: str, value) -> None:
        for i in range(self.size):
            ctx = self._lib.nori_hip_group_ctx(self._h, i)
            if self._lib.nori_hip_set_option(ctx, key.encode(), str(value).encode()) != 0:
                raise NoriError(f"set_option({key}) on group member {i}: {self._lib.nori_hip_last_error(ctx).decode()}")

    def render_host(self, split="tile", merge="reduce", spp_count=None, spp_begin=0, count_traversal=False):
        """(rgbw, stats dict, merge ms): the merged frame of all devices."""
        c = self.scene.camera
        b = self._lib.nori_hip_border_size(self._lib.nori_hip_group_ctx(self._h, 0))
        spp = self.scene.sample_count if spp_count is None else spp_count
        p = Renderer._params(spp_begin, spp, 1, 0, count_traversal, None)
        rgbw = np.zeros((c.height + 2 * b, c.width + 2 * b, 4), np.float32)
        st = capi.RenderStats()
        ms = C.c_float(0.0)
        self._check(self._lib.nori_hip_group_render_host(self._h, C.byref(p), self.SPLIT[split], self.MERGE[merge], c.width, c.height,
                                                         ptr(rgbw), C.byref(st), C.byref(ms)), "group_render_host")
        return rgbw, st.as_dict(), float(ms.value)


def develop_host(rgbw: np.ndarray, border: int) -> np.ndarray:
    """ImageBlock::toBitmap (src/block.cpp:45-51) for a host RGBW frame.  Pure
    reshaping/division of an already rendered frame (output side, not the hot path)."""
    h, w = rgbw.shape[0] - 2 * border, rgbw.shape[1] - 2 * border
    core = rgbw[border:border + h, border:border + w]
    wgt = core[..., 3:4]
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = np.where(wgt != 0, core[..., :3] / wgt, np.float32(0))
    return rgb.astype(np.float32)
