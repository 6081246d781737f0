"""Test-side wrappers: the CPU oracle (oracle/liboracle.so) and the CPU
emulation of the device headers (tests/emu/libnori_emu.so), both exposing the
same Python API as nori_amd.render.Renderer so parity tests are one-liners.

Nothing here is imported by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from nori_amd import _capi as capi
from nori_amd._capi import ptr
from nori_amd.scene import Bsdf, Scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_P = C.c_void_p


def _make(dirpath, target):
    so = os.path.join(dirpath, target)
    srcs = [os.path.join(dirpath, f) for f in os.listdir(dirpath) if f.endswith((".cpp", ".h"))]
    dev = os.path.join(ROOT, "nori_amd", "csrc", "device")
    srcs += [os.path.join(dev, f) for f in os.listdir(dev) if f.endswith((".cpp", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "nori_hip.h"))
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", dirpath, target], check=True, capture_output=True)
    return so


_oracle = None
_emu = None
_oracle_path = None      # set by use_native_oracle() before the first oracle call


def use_native_oracle() -> bool:
    """bench.py's cpu_baseline leg: load the oracle compiled FOR THIS HOST (-O3 -march=native
    -ffp-contract=off, oracle/Makefile target `native`) instead of the portable -O2 checker.  The file is keyed
    by the CPU model, so a copy built on another machine (the snapshot travels) is never loaded.  Returns
    False -- and leaves the portable build in place -- if it cannot be built or the oracle is already loaded."""
    global _oracle_path
    if _oracle is not None:
        return _oracle_path is not None
    try:
        import hashlib
        model = [ln for ln in open("/proc/cpuinfo") if ln.startswith(("model name", "flags"))][:2]
        key = hashlib.sha1("".join(model).encode()).hexdigest()[:10]
        rel = f"_native/liboracle_native_{key}.so"
        odir = os.path.join(ROOT, "oracle")
        so = os.path.join(odir, rel)
        srcs = [os.path.join(odir, f) for f in ("oracle.cpp", "oracle_scene.h", "oracle_math.h", "oracle.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.run(["make", "-C", odir, "native", f"NATIVE_SO={rel}"], check=True, capture_output=True, timeout=300)
        _oracle_path = so
        return True
    except Exception:
        _oracle_path = None
        return False


def oracle_lib():
    global _oracle
    if _oracle is None:
        # NORI_ORACLE_LIBRARY: another build of the oracle, e.g. oracle/liboracle_glibc.so (host libm instead of the pinned
        # specification) -- the independent witness of tests/test_oracle_goldens.py::test_specified_libm_vs_host_libm_renders
        lib = C.CDLL(_oracle_path or os.environ.get("NORI_ORACLE_LIBRARY") or _make(os.path.join(ROOT, "oracle"), "liboracle.so"))
        protos = {
            "oracle_create": (C.c_int, [C.POINTER(capi.SceneDesc), C.POINTER(_P)]),
            "oracle_destroy": (None, [_P]),
            "oracle_set_accel": (C.c_int, [_P, C.c_int]),
            "oracle_border_size": (C.c_int, [_P]),
            "oracle_filter_table": (C.c_int, [_P, _P]),
            "oracle_intersect": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
            "oracle_sample_rays": (C.c_int, [_P, _P, C.c_size_t, _P]),
            "oracle_li": (C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
            "oracle_bsdf_sample": (C.c_int, [C.POINTER(capi.BsdfDesc), _P, _P, C.c_size_t, _P, _P, _P, _P]),
            "oracle_bsdf_eval": (C.c_int, [C.POINTER(capi.BsdfDesc), _P, _P, C.c_size_t, _P]),
            "oracle_bsdf_pdf": (C.c_int, [C.POINTER(capi.BsdfDesc), _P, _P, C.c_size_t, _P]),
            "oracle_warp": (C.c_int, [C.c_int, C.c_float, _P, C.c_size_t, _P]),
            "oracle_warp_pdf": (C.c_int, [C.c_int, C.c_float, _P, C.c_size_t, _P]),
            "oracle_pcg32_floats": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, _P]),
            "oracle_pcg32_uints": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, _P]),
            "oracle_splat": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
            "oracle_fresnel": (C.c_float, [C.c_float, C.c_float, C.c_float]),
            "oracle_libm_eval": (C.c_int, [C.c_int, _P, C.c_size_t, _P]),
            "oracle_render": (C.c_int, [_P, C.POINTER(capi.RenderParams), _P, C.POINTER(capi.RenderStats), C.c_int]),
            "oracle_develop": (C.c_int, [_P, _P, _P]),
        }
        _oracle = capi.bind(lib, protos)
    return _oracle


def emu_lib():
    global _emu
    if _emu is None:
        lib = C.CDLL(_make(os.path.join(ROOT, "tests", "emu"), "libnori_emu.so"))
        protos = {
            "emu_create": (C.c_int, [C.POINTER(capi.SceneDesc), C.POINTER(_P)]),
            "emu_destroy": (None, [_P]),
            "emu_accel_info": (C.c_int, [_P, C.POINTER(capi.AccelInfo)]),
            "emu_border_size": (C.c_int, [_P]),
            "emu_intersect": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
            "emu_nodeq_check": (C.c_longlong, [_P, _P, C.c_size_t, _P]),
            "emu_split_parts": (C.c_int, [_P, C.c_uint32, C.c_float, _P, _P, _P, _P, C.c_int]),
            "emu_node_records": (C.c_longlong, [_P, _P, _P, _P, C.c_size_t]),
            "emu_packed_vs_scalar": (C.c_size_t, [C.c_size_t, C.c_uint64]),
            "emu_libm_eval": (C.c_int, [C.c_int, _P, C.c_size_t, _P]),
            "emu_sample_rays": (C.c_int, [_P, _P, C.c_size_t, _P]),
            "emu_li": (C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
            "emu_li_records": (C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
            "emu_bsdf_sample": (C.c_int, [C.POINTER(capi.BsdfDesc), _P, _P, C.c_size_t, _P, _P, _P, _P]),
            "emu_bsdf_eval": (C.c_int, [C.POINTER(capi.BsdfDesc), _P, _P, C.c_size_t, _P]),
            "emu_bsdf_pdf": (C.c_int, [C.POINTER(capi.BsdfDesc), _P, _P, C.c_size_t, _P]),
            "emu_warp": (C.c_int, [C.c_int, C.c_float, _P, C.c_size_t, _P]),
            "emu_warp_pdf": (C.c_int, [C.c_int, C.c_float, _P, C.c_size_t, _P]),
            "emu_pcg32_floats": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, _P]),
            "emu_render": (C.c_int, [_P, C.POINTER(capi.RenderParams), _P, C.POINTER(capi.RenderStats)]),
            "emu_group_render": (C.c_int, [_P, C.c_int, C.POINTER(capi.RenderParams), C.c_int, C.c_int, _P, C.POINTER(capi.RenderStats)]),
            "emu_group_block_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
        }
        _emu = capi.bind(lib, protos)
    return _emu


def _f32(a, k):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, k)


class _CpuBackend:
    """Shared Python surface of Oracle and Emu (mirrors Renderer)."""
    prefix = ""
    ctx_for_ops = False

    def __init__(self, scene: Scene):
        self.scene = scene
        desc, keep = scene.c_desc()
        h = _P()
        rc = self._fn("create")(C.byref(desc), C.byref(h))
        assert rc == 0, f"{self.prefix}create failed: {rc}"
        self._h = h
        del keep

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def close(self):
        if getattr(self, "_h", None):
            self._fn("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def border(self):
        return self._fn("border_size")(self._h)

    def frame_shape(self):
        b = self.border
        c = self.scene.camera
        return (c.height + 2 * b, c.width + 2 * b, 4)

    def intersect(self, rays, shadow=False):
        rays = np.ascontiguousarray(rays, dtype=capi.RAY_DTYPE)
        its = np.zeros(rays.shape[0], dtype=capi.ITS_DTYPE)
        assert self._fn("intersect")(self._h, ptr(rays), ptr(its), rays.shape[0], int(shadow)) == 0
        return its

    def sample_rays(self, pixel_samples):
        ps = _f32(pixel_samples, 2)
        rays = np.zeros(ps.shape[0], dtype=capi.RAY_DTYPE)
        assert self._fn("sample_rays")(self._h, ptr(ps), ps.shape[0], ptr(rays)) == 0
        return rays

    def li(self, rays, seed_state, seed_seq):
        rays = np.ascontiguousarray(rays, dtype=capi.RAY_DTYPE)
        ss = np.ascontiguousarray(seed_state, dtype=np.uint64)
        sq = np.ascontiguousarray(seed_seq, dtype=np.uint64)
        out = np.zeros((rays.shape[0], 3), np.float32)
        assert self._fn("li")(self._h, ptr(rays), rays.shape[0], ptr(ss), ptr(sq), ptr(out)) == 0
        return out

    # static ops (no scene needed)
    @classmethod
    def bsdf_sample(cls, bsdf: Bsdf, wi, sample):
        wi, sample = _f32(wi, 3), _f32(sample, 2)
        n = wi.shape[0]
        wo, w = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
        eta, meas = np.zeros(n, np.float32), np.zeros(n, np.int32)
        d = bsdf.desc()
        assert getattr(cls.lib_fn(), cls.prefix + "bsdf_sample")(C.byref(d), ptr(wi), ptr(sample), n, ptr(wo), ptr(w), ptr(eta), ptr(meas)) == 0
        return wo, w, eta, meas

    @classmethod
    def bsdf_eval(cls, bsdf, wi, wo):
        wi, wo = _f32(wi, 3), _f32(wo, 3)
        out = np.zeros((wi.shape[0], 3), np.float32)
        d = bsdf.desc()
        assert getattr(cls.lib_fn(), cls.prefix + "bsdf_eval")(C.byref(d), ptr(wi), ptr(wo), wi.shape[0], ptr(out)) == 0
        return out

    @classmethod
    def bsdf_pdf(cls, bsdf, wi, wo):
        wi, wo = _f32(wi, 3), _f32(wo, 3)
        out = np.zeros(wi.shape[0], np.float32)
        d = bsdf.desc()
        assert getattr(cls.lib_fn(), cls.prefix + "bsdf_pdf")(C.byref(d), ptr(wi), ptr(wo), wi.shape[0], ptr(out)) == 0
        return out

    @classmethod
    def warp(cls, name, sample, param=0.0):
        s = _f32(sample, 2)
        out = np.zeros((s.shape[0], 3), np.float32)
        assert getattr(cls.lib_fn(), cls.prefix + "warp")(capi.WARP_NAMES[name], float(param), ptr(s), s.shape[0], ptr(out)) == 0
        return out

    @classmethod
    def warp_pdf(cls, name, points, param=0.0):
        p = _f32(points, 3)
        out = np.zeros(p.shape[0], np.float32)
        assert getattr(cls.lib_fn(), cls.prefix + "warp_pdf")(capi.WARP_NAMES[name], float(param), ptr(p), p.shape[0], ptr(out)) == 0
        return out

    @classmethod
    def libm(cls, op: str, x):
        """sin / cos / log / exp as this backend evaluates them on the render path (one specification:
        rt_math.h on the device side, oracle_libm.h in the oracle)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros_like(x)
        assert getattr(cls.lib_fn(), cls.prefix + "libm_eval")({"sin": 0, "cos": 1, "log": 2, "exp": 3}[op], ptr(x), x.size, ptr(out)) == 0
        return out

    @classmethod
    def pcg32_floats(cls, seed_state, seed_seq, count):
        ss = np.ascontiguousarray(seed_state, dtype=np.uint64)
        sq = np.ascontiguousarray(seed_seq, dtype=np.uint64)
        out = np.zeros((ss.shape[0], count), np.float32)
        assert getattr(cls.lib_fn(), cls.prefix + "pcg32_floats")(ptr(ss), ptr(sq), ss.shape[0], count, ptr(out)) == 0
        return out

    def _params(self, spp_count, spp_begin, tile_mod, tile_rem, count_traversal, seed_mode=capi.SEED_PER_SAMPLE):
        p = capi.RenderParams()
        p.spp_begin = int(spp_begin)
        p.spp_count = int(self.scene.sample_count if spp_count is None else spp_count)
        p.tile_mod, p.tile_rem = int(tile_mod), int(tile_rem)
        p.seed_mode = seed_mode
        p.count_traversal = int(bool(count_traversal))
        p.stream = None
        return p


class Oracle(_CpuBackend):
    prefix = "oracle_"

    @staticmethod
    def lib_fn():
        return oracle_lib()

    @property
    def lib(self):
        return oracle_lib()

    def __init__(self, scene, use_bvh=False):
        super().__init__(scene)
        self.set_accel(use_bvh)

    def set_accel(self, use_bvh: bool):
        assert self.lib.oracle_set_accel(self._h, int(use_bvh)) == 0

    def filter_table(self):
        t = np.zeros(33, np.float32)
        self.lib.oracle_filter_table(self._h, ptr(t))
        return t

    def splat(self, positions, values, rgbw=None):
        p, v = _f32(positions, 2), _f32(values, 3)
        if rgbw is None:
            rgbw = np.zeros(self.frame_shape(), np.float32)
        assert self.lib.oracle_splat(self._h, ptr(p), ptr(v), p.shape[0], ptr(rgbw)) == 0
        return rgbw

    def render_host(self, spp_count=None, spp_begin=0, tile_mod=1, tile_rem=0, count_traversal=False,
                    seed_mode=capi.SEED_PER_SAMPLE, threads=0):
        p = self._params(spp_count, spp_begin, tile_mod, tile_rem, count_traversal, seed_mode)
        rgbw = np.zeros(self.frame_shape(), np.float32)
        st = capi.RenderStats()
        rc = self.lib.oracle_render(self._h, C.byref(p), ptr(rgbw), C.byref(st), int(threads))
        assert rc == 0, rc
        return rgbw, st.as_dict()

    def develop(self, rgbw):
        c = self.scene.camera
        rgb = np.zeros((c.height, c.width, 3), np.float32)
        self.lib.oracle_develop(self._h, ptr(np.ascontiguousarray(rgbw, np.float32)), ptr(rgb))
        return rgb

    @staticmethod
    def pcg32_uints(state=0, seq=0, use_default=False, count=6):
        out = np.zeros(count, np.uint32)
        oracle_lib().oracle_pcg32_uints(state, seq, int(use_default), count, ptr(out))
        return out

    @staticmethod
    def fresnel(c, e, i):
        return oracle_lib().oracle_fresnel(c, e, i)


class Emu(_CpuBackend):
    prefix = "emu_"

    @staticmethod
    def lib_fn():
        return emu_lib()

    @property
    def lib(self):
        return emu_lib()

    def accel_info(self):
        info = capi.AccelInfo()
        self.lib.emu_accel_info(self._h, C.byref(info))
        return info.as_dict()

    def node_records(self):
        """(nodes as [n, 16] float32 in the 64-B layout, as [n, 8] uint32 in the 32-B layout, grid mn[3], grid scale[3]); None if the
        tree has no 32-B records"""
        cap = max(1, self.accel_info()["n_nodes"])
        n64 = np.zeros((cap, 16), np.float32); n32 = np.zeros((cap, 8), np.uint32); grid = np.zeros(6, np.float32)
        n = self.lib.emu_node_records(self._h, ptr(n64), ptr(n32), ptr(grid), cap)
        return None if n < 0 else (n64[:n], n32[:n], grid[:3].copy(), grid[3:].copy())

    def nodeq_check(self, rays):
        """rt_nodeq.h: (ray, child box) pairs the exact slab test accepts and the 32-B record rejects (must be 0), and how many
        pairs the exact test / the record accept; None if the tree has no 32-B records."""
        rays = np.ascontiguousarray(rays, dtype=capi.RAY_DTYPE)
        counts = np.zeros(2, np.uint64)
        bad = self.lib.emu_nodeq_check(self._h, ptr(rays), rays.shape[0], ptr(counts))
        return None if bad < 0 else (int(bad), int(counts[0]), int(counts[1]))

    def render_host(self, spp_count=None, spp_begin=0, tile_mod=1, tile_rem=0, count_traversal=False):
        p = self._params(spp_count, spp_begin, tile_mod, tile_rem, count_traversal)
        rgbw = np.zeros(self.frame_shape(), np.float32)
        st = capi.RenderStats()
        assert self.lib.emu_render(self._h, C.byref(p), ptr(rgbw), C.byref(st)) == 0
        return rgbw, st.as_dict()

    def group_render_host(self, n_ranks, split="tile", merge="reduce", spp_count=None):
        """The device group's driver (threads, shares, merges: group_merge.h) with CPU ranks; rc != 0 -> None."""
        p = self._params(spp_count, 0, 1, 0, False)
        rgbw = np.zeros(self.frame_shape(), np.float32)
        st = capi.RenderStats()
        rc = self.lib.emu_group_render(self._h, int(n_ranks), C.byref(p), {"tile": 0, "sample": 1}[split], {"reduce": 0, "gather": 1}[merge], ptr(rgbw), C.byref(st))
        return (rgbw, st.as_dict()) if rc == 0 else None
