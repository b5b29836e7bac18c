"""Host-side Python mirror of the reference's WASM export surface for the MSM path, backed by the
HIP engine (libmsm_amd.so) through its C ABI (include/msm_amd.h).

The reference exposes, per curve module (wasmcurves/build/bls12381.wasm, bn128.wasm):
    g1m_multiexp, g1m_multiexpAffine, g1m_multiexp_chunk, g1m_multiexpAffine_chunk, g2m_* ...
    (wasmcurves/src/build_multiexp.js:251-257,96-104; wired at build_curve_jacobian_a0.js:1426-1430)
called as  pb.g1m_multiexpAffine(pBases, pScalars, scalarSize, n, pOut)  (wasmcurves/test/bn128.js:493).
Here the same names take byte buffers instead of linear-memory offsets and return the result bytes:

    eng = msm_amd.Engine()                       # one context on one GPU
    bls = eng.curve("bls12381")
    out = bls.g1m_multiexpAffine(bases, scalars, 32, n)      # 144 bytes, normalised Jacobian

There is no CPU fallback: if the HIP library cannot be loaded or no GPU is present, construction
raises.  PyTorch is optional plumbing: CUDA tensors are accepted as device-resident inputs.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libmsm_amd.so"))

CURVES = {"bls12381": 0, "bn128": 1}
N8 = {"bls12381": 48, "bn128": 32}

FLAG_JACOBIAN = 1
FLAG_DEVICE = 2
FLAG_DEVICE_BASES = 4
FLAG_FFT_INVERSE = 8
FLAG_BASES_IN_SUBGROUP = 16


class MsmAmdError(RuntimeError):
    """code: the library's MSM_AMD_E* value when the error came from it (EINVAL -1, EHIP -2, ENOMEM -3, EPEER -4)"""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


EINVAL, EHIP, ENOMEM, EPEER = -1, -2, -3, -4


_lib = None


def load_library(path=None):
    """dlopen libmsm_amd.so (raises MsmAmdError if it is missing: the product has no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("MSM_AMD_LIB", _LIB_PATH)
    # Library load order matters in a process that also uses torch on the GPU: torch ships its own HIP runtime, and if the engine's
    # (/opt/rocm's libamdhip64) is loaded first, torch's later finds "no ROCm-capable device".  Loaded the other way round both
    # work (the engine then runs on torch's runtime).  So when torch is installed it is imported first -- importing does not touch
    # the GPU.  MSM_AMD_NO_TORCH=1 skips this (hosts that never use torch).
    if "torch" not in sys.modules and os.environ.get("MSM_AMD_NO_TORCH", "") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(path):
        raise MsmAmdError(f"HIP engine library not found at {path}; build it with `python __graft_entry__.py` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(path)
    c = ctypes
    vp, u32, u8p = c.c_void_p, c.c_uint32, c.c_void_p
    L.msm_amd_create.argtypes = [c.c_int, c.POINTER(vp)]
    L.msm_amd_destroy.argtypes = [vp]
    L.msm_amd_destroy.restype = None
    L.msm_amd_last_error.argtypes = [vp]
    L.msm_amd_last_error.restype = c.c_char_p
    L.msm_amd_set_window_bits.argtypes = [vp, u32]
    L.msm_amd_set_option.argtypes = [vp, c.c_char_p, u32]
    L.msm_amd_multiexp.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, u8p]
    L.msm_amd_multiexp_async.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, c.POINTER(c.c_uint64)]
    L.msm_amd_wait.argtypes = [vp, c.c_uint64, u8p]
    L.msm_amd_multiexp_shard_async.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, u32, u32, c.POINTER(c.c_uint64)]
    L.msm_amd_multiexp_chunk.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, u32, u32, u8p]
    L.msm_amd_multiexp_shard.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, u32, u32, u8p]
    L.msm_amd_combine.argtypes = [c.c_int, c.c_int, vp, u32, u8p]
    L.msm_amd_dist_unique_id.argtypes = [u8p]
    L.msm_amd_dist_init.argtypes = [vp, c.c_int, c.c_int, vp]
    L.msm_amd_dist_destroy.argtypes = [vp]
    L.msm_amd_dist_info.argtypes = [vp, c.POINTER(c.c_int), c.POINTER(c.c_int)]
    L.msm_amd_dist_comm_count.argtypes = [vp, c.POINTER(c.c_int), c.POINTER(c.c_int)]
    L.msm_amd_multiexp_dist.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, u8p]
    L.msm_amd_multiexp_dist_async.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, c.POINTER(c.c_uint64)]
    L.msm_amd_dist_wait.argtypes = [vp, c.c_int, c.c_int, c.c_uint64, u8p]
    L.msm_amd_dist_plan.argtypes = [vp, c.c_int, c.c_int, u32, u32, u32, u32] + [c.POINTER(u32)] * 4
    L.msm_amd_multiexp_dist_local.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, u32, u8p]
    L.msm_amd_multiexp_dist_local_async.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, u32, u32, c.POINTER(c.c_uint64)]
    L.msm_amd_prepare_bases.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, u32, c.POINTER(vp)]
    L.msm_amd_release_bases.argtypes = [vp, vp]
    L.msm_amd_bases_info.argtypes = [vp, c.POINTER(u32), c.POINTER(c.c_uint64)]
    L.msm_amd_plan_prepared.argtypes = [c.c_int, c.c_int, u32, u32, u32, c.POINTER(u32), c.POINTER(c.c_uint64)]
    L.msm_amd_multiexp_prepared.argtypes = [vp, vp, u32, vp, u32, u32, u8p]
    L.msm_amd_multiexp_prepared_async.argtypes = [vp, vp, u32, vp, u32, u32, c.POINTER(c.c_uint64)]
    L.msm_amd_multiexp_dist_prepared.argtypes = [vp, vp, u32, vp, u32, u32, u8p]
    L.msm_amd_multiexp_dist_prepared_async.argtypes = [vp, vp, u32, vp, u32, u32, c.POINTER(c.c_uint64)]
    L.msm_amd_normalize.argtypes = [c.c_int, c.c_int, vp, u8p]
    L.msm_amd_to_affine.argtypes = [c.c_int, c.c_int, vp, u8p]
    L.msm_amd_eq.argtypes = [c.c_int, c.c_int, vp, vp]
    L.msm_amd_f_to_montgomery.argtypes = [c.c_int, vp, u32, u8p]
    L.msm_amd_f_from_montgomery.argtypes = [c.c_int, vp, u32, u8p]
    for _fn in ("msm_amd_batch_to_affine", "msm_amd_batch_to_jacobian", "msm_amd_batch_lem_to_u", "msm_amd_batch_u_to_lem"):
        getattr(L, _fn).argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp]
    L.msm_amd_batch_lem_to_c.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp]
    L.msm_amd_batch_c_to_lem.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp]
    L.msm_amd_batch_in_curve_affine.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp]
    L.msm_amd_batch_in_group_affine.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp]
    L.msm_amd_batch_apply_key.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp, vp, vp]
    L.msm_amd_fr_batch_apply_key.argtypes = [vp, c.c_int, u32, vp, u32, vp, vp, vp]
    L.msm_amd_fft.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp]
    L.msm_amd_fft_join.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, vp, vp]
    L.msm_amd_fft_join_ext.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, vp, vp, u32]
    L.msm_amd_prepare_lagrange_evaluation.argtypes = [vp, c.c_int, c.c_int, u32, vp, vp, u32, vp, vp, u32]
    L.msm_amd_fft_mix.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, u32]
    L.msm_amd_fft_final.argtypes = [vp, c.c_int, c.c_int, u32, vp, u32, vp]
    L.msm_amd_fr_batch_montgomery.argtypes = [vp, c.c_int, u32, vp, u32, c.c_int, vp]
    L.msm_amd_glv_decompose_scalar.argtypes = [vp, u8p, c.POINTER(u32)]
    L.msm_amd_glv_split.argtypes = [vp, u8p, c.POINTER(u32)]
    L.msm_amd_glv_endomorphism.argtypes = [vp, u32, u8p]
    L.msm_amd_glv_preprocess.argtypes = [vp, u32, vp, vp, u32, vp, vp]
    L.msm_amd_dev_alloc.argtypes = [vp, c.c_size_t, c.POINTER(vp)]
    L.msm_amd_dev_free.argtypes = [vp, vp]
    L.msm_amd_h2d.argtypes = [vp, vp, vp, c.c_size_t]
    L.msm_amd_d2h.argtypes = [vp, vp, vp, c.c_size_t]
    L.msm_amd_gen_points.argtypes = [vp, c.c_int, c.c_int, c.c_uint64, u32, vp]
    L.msm_amd_gen_points_range.argtypes = [vp, c.c_int, c.c_int, c.c_uint64, u32, u32, vp]
    L.msm_amd_get_timings.argtypes = [vp, c.POINTER(c.c_float), c.c_int]
    L.msm_amd_get_geometry.argtypes = [vp, c.POINTER(u32)]
    _lib = L
    return L


class DeviceBuffer:
    """GPU memory owned by an Engine (hipMalloc); `.ptr` is the raw device address."""

    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, nbytes
        p = ctypes.c_void_p()
        eng._chk(eng.L.msm_amd_dev_alloc(eng.ctx, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, data):
        data = bytes(data)
        assert len(data) <= self.nbytes
        self.eng._chk(self.eng.L.msm_amd_h2d(self.eng.ctx, self.ptr, data, len(data)))
        return self

    def download(self, nbytes=None):
        nbytes = self.nbytes if nbytes is None else nbytes
        buf = ctypes.create_string_buffer(nbytes)
        self.eng._chk(self.eng.L.msm_amd_d2h(self.eng.ctx, buf, self.ptr, nbytes))
        return buf.raw

    def free(self):
        if self.ptr:
            self.eng.L.msm_amd_dev_free(self.eng.ctx, self.ptr)
            self.ptr = None

    def view(self, offset, nbytes=None):
        """The bytes [offset, offset + nbytes) of this buffer as a device pointer of their own (no ownership): a slice of resident
        bases or scalars handed to a call."""
        return DeviceView(self, offset, self.nbytes - offset if nbytes is None else nbytes)


class DeviceView(DeviceBuffer):
    def __init__(self, parent, offset, nbytes):
        assert 0 <= offset and offset + nbytes <= parent.nbytes
        self.eng, self.nbytes, self.parent = parent.eng, nbytes, parent
        self.ptr = parent.ptr + offset

    def free(self):
        self.ptr = None


def _as_ptr(x):
    """(pointer, is_device, keepalive) for bytes-like, numpy, DeviceBuffer or torch tensors."""
    if isinstance(x, DeviceBuffer):
        return x.ptr, True, x
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):   # torch.Tensor
        if not x.is_contiguous():
            raise MsmAmdError("tensor must be contiguous")
        return x.data_ptr(), bool(x.is_cuda), x
    if hasattr(x, "ctypes") and hasattr(x, "flags"):       # numpy
        if not x.flags["C_CONTIGUOUS"]:
            raise MsmAmdError("array must be C-contiguous")
        return x.ctypes.data, False, x
    b = bytes(x) if not isinstance(x, bytes) else x
    return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value, False, b


class Engine:
    """One engine context = one GPU + one stream + its scratch buffers (msm_amd_create)."""

    def __init__(self, device=0, lib_path=None):
        self.L = load_library(lib_path)
        ctx = ctypes.c_void_p()
        rc = self.L.msm_amd_create(int(device), ctypes.byref(ctx))
        if rc != 0:
            raise MsmAmdError(f"msm_amd_create(device={device}) failed with {rc}: " + ("the device has too little free memory for the kernels' scratch" if rc == ENOMEM else "no usable MI355X / HIP runtime"), rc)
        self.ctx = ctx
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.L.msm_amd_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MsmAmdError(f"libmsm_amd error {rc}: {self.L.msm_amd_last_error(self.ctx).decode()}", rc)

    def curve(self, name):
        return Curve(self, name)

    def set_window_bits(self, c):
        self._chk(self.L.msm_amd_set_window_bits(self.ctx, int(c)))

    def set_option(self, name, value):
        self._chk(self.L.msm_amd_set_option(self.ctx, name.encode(), int(value)))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # --- multi-GPU group (RCCL behind the C ABI: msm_amd_dist_*) -------------------------------------------
    def dist_init(self, rank, world, unique_id):
        """Join the group of `world` contexts (one per GPU / process); `unique_id` = the 128 bytes rank 0 got from dist_unique_id()."""
        self._chk(self.L.msm_amd_dist_init(self.ctx, int(rank), int(world), bytes(unique_id)))

    def dist_destroy(self):
        self._chk(self.L.msm_amd_dist_destroy(self.ctx))

    def dist_info(self):
        """(rank, world) of this engine's group; world = 0: not in a group"""
        r, w = ctypes.c_int(), ctypes.c_int()
        self._chk(self.L.msm_amd_dist_info(self.ctx, ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    def dist_comm_count(self):
        """(ranks the communicator itself counts -- ncclCommCount --, whether the group has been aborted)"""
        n, dead = ctypes.c_int(), ctypes.c_int()
        self._chk(self.L.msm_amd_dist_comm_count(self.ctx, ctypes.byref(n), ctypes.byref(dead)))
        return n.value, bool(dead.value)

    def dist_plan(self, curve, group, scalar_size, n, world=0, rank=0):
        """msm_amd_dist_plan: how a call of n points is split over a group of `world` ranks (0: this engine's own group and rank) --
        {gp, gw, first, count}: gp point slices x gw window groups; [first, first + count) is the slice of the points that rank
        `rank` = p * gw + w works on."""
        v = [ctypes.c_uint32() for _ in range(4)]
        rc = self.L.msm_amd_dist_plan(self.ctx, CURVES[curve], group, scalar_size, n, world, rank, *[ctypes.byref(x) for x in v])
        if rc:
            raise MsmAmdError(f"msm_amd_dist_plan failed ({rc}): not in a group, or bad world / rank / scalar_size", rc)
        return dict(zip(("gp", "gw", "first", "count"), (x.value for x in v)))

    def timings(self):
        arr = (ctypes.c_float * 8)()
        k = self.L.msm_amd_get_timings(self.ctx, arr, 8)
        names = ["gpu_total", "histogram", "scan_order", "scatter", "accumulate", "reduce", "host_combine", "to_affine"]
        return {names[i]: float(arr[i]) for i in range(k)}

    def hints(self):
        """msm_amd_get_hints: the call-to-call state of the context -- {wide, skew: calls left in that layout, slots, entries_per_bucket: what the last call reported (the density the next one expects), work_list_T, narrow_bits: the observed scalar width the next calls tile}"""
        arr = (ctypes.c_uint32 * 4)()
        self.L.msm_amd_get_hints.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.L.msm_amd_get_hints(self.ctx, arr)
        return {"wide": arr[0], "skew": arr[1], "slots": arr[2] & 0xffff, "entries_per_bucket": arr[2] >> 16, "work_list_T": arr[3] & 0xffff, "narrow_bits": arr[3] >> 16}

    def geometry(self):
        arr = (ctypes.c_uint32 * 6)()
        self.L.msm_amd_get_geometry(self.ctx, arr)
        return {"c": arr[0], "windows": arr[1], "buckets_per_window": arr[2], "windows_owned": arr[3],
                "entries": arr[4] | (arr[5] << 32)}


class Ticket:
    """Handle of a queued MSM (msm_amd_multiexp_async); wait() returns the normalised Jacobian bytes."""

    def __init__(self, curve, group, value, keepalive):
        self.curve, self.group, self.value, self._keep = curve, group, value, keepalive

    def wait(self):
        out = ctypes.create_string_buffer(self.curve.jacobian_bytes(self.group))
        self.curve.eng._chk(self.curve.eng.L.msm_amd_wait(self.curve.eng.ctx, self.value, out))
        self._keep = None
        return out.raw


class PreparedBases:
    """Handle of msm_amd_prepare_bases: the window table of a fixed set of points, resident in HBM.  multiexp / multiexp_async run an
    MSM over the first n of them (all of them by default); release() frees the table."""

    def __init__(self, curve, group, handle, n):
        self.curve, self.group, self.handle, self.n = curve, group, handle, n

    def info(self):
        arr = (ctypes.c_uint32 * 6)()
        tb = ctypes.c_uint64()
        self.curve.eng.L.msm_amd_bases_info(self.handle, arr, ctypes.byref(tb))
        return {"n": arr[0], "scalar_bits": arr[1], "c": arr[2], "windows": arr[3], "signed_windows": arr[4], "buckets": arr[5],
                "table_bytes": tb.value}

    def multiexp(self, scalars, scalar_size=32, n=None):
        ps, dev_s, ks = _as_ptr(scalars)
        out = ctypes.create_string_buffer(self.curve.jacobian_bytes(self.group))
        eng = self.curve.eng
        eng._chk(eng.L.msm_amd_multiexp_prepared(eng.ctx, self.handle, FLAG_DEVICE if dev_s else 0, ps, scalar_size, self.n if n is None else n, out))
        return out.raw

    def multiexp_async(self, scalars, scalar_size=32, n=None):
        ps, dev_s, ks = _as_ptr(scalars)
        t = ctypes.c_uint64()
        eng = self.curve.eng
        eng._chk(eng.L.msm_amd_multiexp_prepared_async(eng.ctx, self.handle, FLAG_DEVICE if dev_s else 0, ps, scalar_size,
                                                       self.n if n is None else n, ctypes.byref(t)))
        return Ticket(self.curve, self.group, t.value, (ks, self))

    def multiexp_dist_async(self, scalars_local, scalar_size=32, n=None):
        """msm_amd_multiexp_dist_prepared_async: this table holds this rank's slice of a group's points; the ticket's wait() exchanges the
        partials of all ranks and returns the sum (Engine.dist_init first)."""
        ps, dev_s, ks = _as_ptr(scalars_local)
        t = ctypes.c_uint64()
        eng = self.curve.eng
        eng._chk(eng.L.msm_amd_multiexp_dist_prepared_async(eng.ctx, self.handle, FLAG_DEVICE if dev_s else 0, ps, scalar_size,
                                                            self.n if n is None else n, ctypes.byref(t)))
        return DistTicket(self.curve, self.group, t.value, (ks, self))

    def multiexp_dist(self, scalars_local, scalar_size=32, n=None):
        return self.multiexp_dist_async(scalars_local, scalar_size, n).wait()

    def release(self):
        if self.handle:
            self.curve.eng._chk(self.curve.eng.L.msm_amd_release_bases(self.curve.eng.ctx, self.handle))
            self.handle = None


class DistTicket(Ticket):
    """Handle of a queued sharded MSM of a multi-GPU group; wait() exchanges the partials over RCCL and returns the sum."""

    def wait(self):
        out = ctypes.create_string_buffer(self.curve.jacobian_bytes(self.group))
        self.curve.eng._chk(self.curve.eng.L.msm_amd_dist_wait(self.curve.eng.ctx, self.curve.cid, self.group, self.value, out))
        self._keep = None
        return out.raw


class Curve:
    """Export-named view of one curve (the reference builds one WASM module per curve)."""

    def __init__(self, eng, name):
        if name not in CURVES:
            raise MsmAmdError(f"unknown curve {name!r} (bls12381 | bn128)")
        self.eng, self.name, self.cid, self.n8 = eng, name, CURVES[name], N8[name]

    # sizes, as the reference's constants n8q etc.
    def field_bytes(self, group): return self.n8 * group
    def affine_bytes(self, group): return 2 * self.n8 * group
    def jacobian_bytes(self, group): return 3 * self.n8 * group

    @staticmethod
    def _residency_flags(dev_b, dev_s):
        """MSM_AMD_DEVICE_PTRS when both inputs are device memory, MSM_AMD_DEVICE_BASES when only the bases are (the resident SRS of a
        prover, scalars from the host each call); device scalars with host bases is not a combination the engine has."""
        if dev_s and not dev_b:
            raise MsmAmdError("device scalars with host bases is not supported")
        if dev_b:
            return FLAG_DEVICE if dev_s else FLAG_DEVICE_BASES
        return 0

    def _multiexp(self, group, flags, bases, scalars, scalar_size, n, in_subgroup=False):
        pb, dev_b, kb = _as_ptr(bases)
        ps, dev_s, ks = _as_ptr(scalars)
        flags |= self._residency_flags(dev_b, dev_s) | (FLAG_BASES_IN_SUBGROUP if in_subgroup else 0)
        out = ctypes.create_string_buffer(self.jacobian_bytes(group))
        self.eng._chk(self.eng.L.msm_amd_multiexp(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n, out))
        return out.raw

    def _chunk(self, group, flags, bases, scalars, scalar_size, n, start_bit, chunk_size):
        pb, dev_b, kb = _as_ptr(bases)
        ps, dev_s, ks = _as_ptr(scalars)
        if dev_b != dev_s:
            raise MsmAmdError("bases and scalars must both be host or both be device buffers")
        if dev_b:
            flags |= FLAG_DEVICE
        out = ctypes.create_string_buffer(self.jacobian_bytes(group))
        self.eng._chk(self.eng.L.msm_amd_multiexp_chunk(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n,
                                                        start_bit, chunk_size, out))
        return out.raw

    # --- pipelined form: up to two calls in flight per engine (reduction of one overlaps accumulation of the next)
    def multiexp_async(self, group, bases, scalars, scalar_size, n, jacobian_bases=False, in_subgroup=False):
        pb, dev_b, kb = _as_ptr(bases)
        ps, dev_s, ks = _as_ptr(scalars)
        if dev_b != dev_s:
            raise MsmAmdError("bases and scalars must both be host or both be device buffers")
        flags = (FLAG_JACOBIAN if jacobian_bases else 0) | (FLAG_DEVICE if dev_b else 0) | (FLAG_BASES_IN_SUBGROUP if in_subgroup else 0)
        t = ctypes.c_uint64()
        self.eng._chk(self.eng.L.msm_amd_multiexp_async(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n, ctypes.byref(t)))
        return Ticket(self, group, t.value, (kb, ks))

    # --- the reference's export names -------------------------------------------------------
    # (in_subgroup=True: the caller vouches that the bases lie in the order-r subgroup -- MSM_AMD_BASES_IN_SUBGROUP, include/msm_amd.h)
    def g1m_multiexp(self, bases, scalars, scalar_size, n, in_subgroup=False): return self._multiexp(1, FLAG_JACOBIAN, bases, scalars, scalar_size, n, in_subgroup)
    def g1m_multiexpAffine(self, bases, scalars, scalar_size, n, in_subgroup=False): return self._multiexp(1, 0, bases, scalars, scalar_size, n, in_subgroup)
    def g2m_multiexp(self, bases, scalars, scalar_size, n): return self._multiexp(2, FLAG_JACOBIAN, bases, scalars, scalar_size, n)
    def g2m_multiexpAffine(self, bases, scalars, scalar_size, n): return self._multiexp(2, 0, bases, scalars, scalar_size, n)

    def g1m_multiexp_chunk(self, bases, scalars, scalar_size, n, start_bit, chunk_size):
        return self._chunk(1, FLAG_JACOBIAN, bases, scalars, scalar_size, n, start_bit, chunk_size)

    def g1m_multiexpAffine_chunk(self, bases, scalars, scalar_size, n, start_bit, chunk_size):
        return self._chunk(1, 0, bases, scalars, scalar_size, n, start_bit, chunk_size)

    def g2m_multiexp_chunk(self, bases, scalars, scalar_size, n, start_bit, chunk_size):
        return self._chunk(2, FLAG_JACOBIAN, bases, scalars, scalar_size, n, start_bit, chunk_size)

    def g2m_multiexpAffine_chunk(self, bases, scalars, scalar_size, n, start_bit, chunk_size):
        return self._chunk(2, 0, bases, scalars, scalar_size, n, start_bit, chunk_size)

    # the opt-path signature (wasmcurves/src/build_multiexp_opt.js:1987-1996): scalarSize fixed at 32
    def g1m_multiexp_multiExp(self, points, scalars, n): return self.g1m_multiexpAffine(points, scalars, 32, n)

    # --- prepared bases (msm_amd_prepare_bases): a prover's fixed points, set up once ---------------------
    def prepare_bases(self, group, bases, n, scalar_bits=0, jacobian_bases=False):
        """The window table of n fixed points (host bytes or a device buffer), built on the GPU once; returns a PreparedBases.
        scalar_bits = 0: scalars below the group order."""
        pb, dev_b, kb = _as_ptr(bases)
        flags = (FLAG_JACOBIAN if jacobian_bases else 0) | (FLAG_DEVICE if dev_b else 0)
        h = ctypes.c_void_p()
        self.eng._chk(self.eng.L.msm_amd_prepare_bases(self.eng.ctx, self.cid, group, flags, pb, n, scalar_bits, ctypes.byref(h)))
        return PreparedBases(self, group, h, n)

    # --- multi-GPU building blocks --------------------------------------------------------------
    def multiexp_shard(self, group, bases, scalars, scalar_size, n, shard, nshards, jacobian_bases=False):
        pb, dev_b, kb = _as_ptr(bases)
        ps, dev_s, ks = _as_ptr(scalars)
        flags = (FLAG_JACOBIAN if jacobian_bases else 0) | self._residency_flags(dev_b, dev_s)
        out = ctypes.create_string_buffer(self.jacobian_bytes(group))
        self.eng._chk(self.eng.L.msm_amd_multiexp_shard(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n,
                                                        shard, nshards, out))
        return out.raw

    def multiexp_shard_async(self, group, bases, scalars, scalar_size, n, shard, nshards, jacobian_bases=False):
        pb, dev_b, kb = _as_ptr(bases)
        ps, dev_s, ks = _as_ptr(scalars)
        flags = (FLAG_JACOBIAN if jacobian_bases else 0) | self._residency_flags(dev_b, dev_s)
        t = ctypes.c_uint64()
        self.eng._chk(self.eng.L.msm_amd_multiexp_shard_async(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n,
                                                              shard, nshards, ctypes.byref(t)))
        return Ticket(self, group, t.value, (kb, ks))

    def multiexp_dist(self, group, bases, scalars, scalar_size, n, jacobian_bases=False):
        """msm_amd_multiexp_dist: this rank's windows on this rank's GPU, one ncclAllGather of the partials, the sum (same bytes on
        every rank).  The engine must have joined a group (Engine.dist_init)."""
        pb, dev_b, kb = _as_ptr(bases)
        ps, dev_s, ks = _as_ptr(scalars)
        flags = (FLAG_JACOBIAN if jacobian_bases else 0) | self._residency_flags(dev_b, dev_s)
        out = ctypes.create_string_buffer(self.jacobian_bytes(group))
        self.eng._chk(self.eng.L.msm_amd_multiexp_dist(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n, out))
        return out.raw

    def multiexp_dist_async(self, group, bases, scalars, scalar_size, n, jacobian_bases=False):
        pb, dev_b, kb = _as_ptr(bases)
        ps, dev_s, ks = _as_ptr(scalars)
        flags = (FLAG_JACOBIAN if jacobian_bases else 0) | self._residency_flags(dev_b, dev_s)
        t = ctypes.c_uint64()
        self.eng._chk(self.eng.L.msm_amd_multiexp_dist_async(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n, ctypes.byref(t)))
        return DistTicket(self, group, t.value, (kb, ks))

    def multiexp_dist_local(self, group, bases_local, scalars_local, scalar_size, n_local, n_total, jacobian_bases=False):
        """msm_amd_multiexp_dist_local: like multiexp_dist for a rank that holds only its slice -- n_local points -- of the n_total
        points (Engine.dist_plan says which)."""
        return self.multiexp_dist_local_async(group, bases_local, scalars_local, scalar_size, n_local, n_total, jacobian_bases).wait()

    def multiexp_dist_local_async(self, group, bases_local, scalars_local, scalar_size, n_local, n_total, jacobian_bases=False):
        pb, dev_b, kb = _as_ptr(bases_local)
        ps, dev_s, ks = _as_ptr(scalars_local)
        flags = (FLAG_JACOBIAN if jacobian_bases else 0) | self._residency_flags(dev_b, dev_s)
        t = ctypes.c_uint64()
        self.eng._chk(self.eng.L.msm_amd_multiexp_dist_local_async(self.eng.ctx, self.cid, group, flags, pb, ps, scalar_size, n_local, n_total, ctypes.byref(t)))
        return DistTicket(self, group, t.value, (kb, ks))

    # --- batch conversions either side of the MSM (the reference's g?m_batch* exports), host buffers ---------
    def _conv(self, fn, group, data, in_sz, out_sz):
        data = bytes(data)
        n = len(data) // in_sz
        out = ctypes.create_string_buffer(max(1, n * out_sz))
        self.eng._chk(getattr(self.eng.L, fn)(self.eng.ctx, self.cid, group, 0, data, n, out))
        return out.raw[:n * out_sz]

    def g1m_batchToAffine(self, jac): return self._conv("msm_amd_batch_to_affine", 1, jac, self.jacobian_bytes(1), self.affine_bytes(1))
    def g2m_batchToAffine(self, jac): return self._conv("msm_amd_batch_to_affine", 2, jac, self.jacobian_bytes(2), self.affine_bytes(2))
    def g1m_batchToJacobian(self, aff): return self._conv("msm_amd_batch_to_jacobian", 1, aff, self.affine_bytes(1), self.jacobian_bytes(1))
    def g2m_batchToJacobian(self, aff): return self._conv("msm_amd_batch_to_jacobian", 2, aff, self.affine_bytes(2), self.jacobian_bytes(2))
    def g1m_batchLEMtoU(self, aff): return self._conv("msm_amd_batch_lem_to_u", 1, aff, self.affine_bytes(1), self.affine_bytes(1))
    def g2m_batchLEMtoU(self, aff): return self._conv("msm_amd_batch_lem_to_u", 2, aff, self.affine_bytes(2), self.affine_bytes(2))
    def g1m_batchUtoLEM(self, u): return self._conv("msm_amd_batch_u_to_lem", 1, u, self.affine_bytes(1), self.affine_bytes(1))
    def g2m_batchUtoLEM(self, u): return self._conv("msm_amd_batch_u_to_lem", 2, u, self.affine_bytes(2), self.affine_bytes(2))

    def g1m_batchLEMtoC(self, aff): return self._conv("msm_amd_batch_lem_to_c", 1, aff, self.affine_bytes(1), self.field_bytes(1))
    def g2m_batchLEMtoC(self, aff): return self._conv("msm_amd_batch_lem_to_c", 2, aff, self.affine_bytes(2), self.field_bytes(2))
    def g1m_batchCtoLEM(self, comp): return self._conv("msm_amd_batch_c_to_lem", 1, comp, self.field_bytes(1), self.affine_bytes(1))
    def g2m_batchCtoLEM(self, comp): return self._conv("msm_amd_batch_c_to_lem", 2, comp, self.field_bytes(2), self.affine_bytes(2))

    def batch_in_curve_affine(self, group, points):
        """g?m_inCurveAffine over a batch: list of 0/1 flags"""
        points = bytes(points)
        n = len(points) // self.affine_bytes(group)
        out = ctypes.create_string_buffer(max(1, n))
        self.eng._chk(self.eng.L.msm_amd_batch_in_curve_affine(self.eng.ctx, self.cid, group, 0, points, n, out))
        return list(out.raw[:n])

    def batch_apply_key(self, group, points, first, inc, jacobian_in=False):
        """g?m_batchApplyKey (jacobian_in) / g?m_batchApplyKeyMixed: n normalised Jacobian points"""
        points = bytes(points)
        isz = self.jacobian_bytes(group) if jacobian_in else self.affine_bytes(group)
        n = len(points) // isz
        out = ctypes.create_string_buffer(max(1, n) * self.jacobian_bytes(group))
        self.eng._chk(self.eng.L.msm_amd_batch_apply_key(self.eng.ctx, self.cid, group, FLAG_JACOBIAN if jacobian_in else 0, points, n,
                                                          bytes(first), bytes(inc), out))
        return out.raw[:n * self.jacobian_bytes(group)]

    def g1m_batchApplyKey(self, jac, first, inc): return self.batch_apply_key(1, jac, first, inc, True)
    def g1m_batchApplyKeyMixed(self, aff, first, inc): return self.batch_apply_key(1, aff, first, inc, False)
    def g2m_batchApplyKey(self, jac, first, inc): return self.batch_apply_key(2, jac, first, inc, True)
    def g2m_batchApplyKeyMixed(self, aff, first, inc): return self.batch_apply_key(2, aff, first, inc, False)

    def frm_batchApplyKey(self, elems, first, inc):
        elems = bytes(elems)
        n = len(elems) // 32
        out = ctypes.create_string_buffer(max(1, n) * 32)
        self.eng._chk(self.eng.L.msm_amd_fr_batch_apply_key(self.eng.ctx, self.cid, 0, elems, n, bytes(first), bytes(inc), out))
        return out.raw[:n * 32]

    def fft(self, group, data, inverse=False):
        """frm_fft / frm_ifft (group 0: Fr elements) or g?m_fft / g?m_ifft (Jacobian points in, normalised Jacobian out)"""
        data = bytes(data)
        esz = 32 if group == 0 else self.jacobian_bytes(group)
        n = len(data) // esz
        out = ctypes.create_string_buffer(max(1, n) * esz)
        self.eng._chk(self.eng.L.msm_amd_fft(self.eng.ctx, self.cid, group, FLAG_FFT_INVERSE if inverse else 0, data, n, out))
        return out.raw[:n * esz]

    # --- building blocks of the blocked FFT (build_fft.js:657-1353); group 0 = Fr elements, 1 / 2 = Jacobian points -> normalised
    def _elem_bytes(self, group):
        return 32 if group == 0 else self.jacobian_bytes(group)

    def _pair_op(self, fn, group, buff1, buff2, first, inc, *extra, flags=0):
        b1, b2 = ctypes.create_string_buffer(bytes(buff1), len(buff1)), ctypes.create_string_buffer(bytes(buff2), len(buff2))
        n = len(buff1) // self._elem_bytes(group)
        self.eng._chk(getattr(self.eng.L, fn)(self.eng.ctx, self.cid, group, flags, b1, b2, n, bytes(first), bytes(inc), *extra))
        return b1.raw[:len(buff1)], b2.raw[:len(buff2)]

    def fft_join(self, group, buff1, buff2, first, inc):
        """<p>_fftJoin: returns the two transformed buffers"""
        return self._pair_op("msm_amd_fft_join", group, buff1, buff2, first, inc)

    def fft_join_ext(self, group, buff1, buff2, first, inc, total_bits, inverse=False):
        """<p>_fftJoinExt / <p>_fftJoinExtInv"""
        return self._pair_op("msm_amd_fft_join_ext", group, buff1, buff2, first, inc, int(total_bits), flags=FLAG_FFT_INVERSE if inverse else 0)

    def prepare_lagrange_evaluation(self, group, buff1, buff2, first, inc, total_bits):
        return self._pair_op("msm_amd_prepare_lagrange_evaluation", group, buff1, buff2, first, inc, int(total_bits))

    def fft_mix(self, group, buff, exp):
        """<p>_fftMix: one radix-2 pass of span 2^exp"""
        b = ctypes.create_string_buffer(bytes(buff), len(buff))
        self.eng._chk(self.eng.L.msm_amd_fft_mix(self.eng.ctx, self.cid, group, 0, b, len(buff) // self._elem_bytes(group), int(exp)))
        return b.raw[:len(buff)]

    def fft_final(self, group, buff, factor):
        """<p>_fftFinal: out[i] = factor * in[n - 1 - i]"""
        b = ctypes.create_string_buffer(bytes(buff), len(buff))
        self.eng._chk(self.eng.L.msm_amd_fft_final(self.eng.ctx, self.cid, group, 0, b, len(buff) // self._elem_bytes(group), bytes(factor)))
        return b.raw[:len(buff)]

    def _fr_mont(self, elems, to):
        elems = bytes(elems)
        n = len(elems) // 32
        out = ctypes.create_string_buffer(max(1, n) * 32)
        self.eng._chk(self.eng.L.msm_amd_fr_batch_montgomery(self.eng.ctx, self.cid, 0, elems, n, to, out))
        return out.raw[:n * 32]

    def frm_batchToMontgomery(self, elems): return self._fr_mont(elems, 1)
    def frm_batchFromMontgomery(self, elems): return self._fr_mont(elems, 0)

    def frm_fft(self, elems): return self.fft(0, elems)
    def frm_ifft(self, elems): return self.fft(0, elems, True)
    def g1m_fft(self, jac): return self.fft(1, jac)
    def g1m_ifft(self, jac): return self.fft(1, jac, True)
    def g2m_fft(self, jac): return self.fft(2, jac)
    def g2m_ifft(self, jac): return self.fft(2, jac, True)

    def batch_in_group_affine(self, group, points):
        """g?m_inGroupAffine over a batch (bls12381 only): list of 0/1 flags"""
        points = bytes(points)
        n = len(points) // self.affine_bytes(group)
        out = ctypes.create_string_buffer(max(1, n))
        self.eng._chk(self.eng.L.msm_amd_batch_in_group_affine(self.eng.ctx, self.cid, group, 0, points, n, out))
        return list(out.raw[:n])

    # --- GLV pre-pass (bls12381 G1 only): g1m_glv_preprocessEndomorphism ----------------------------------
    def g1m_glv_preprocessEndomorphism(self, points, scalars, n):
        if self.name != "bls12381":
            raise MsmAmdError("the GLV pre-pass exists for bls12381 G1 only (as in the reference)")
        pp, dev_p, kp = _as_ptr(points)
        ps, dev_s, ks = _as_ptr(scalars)
        if dev_p or dev_s:
            raise MsmAmdError("host buffers expected (use msm_amd_glv_preprocess with MSM_AMD_DEVICE_PTRS from C for resident data)")
        po = ctypes.create_string_buffer(2 * n * 96)
        so = ctypes.create_string_buffer(2 * n * 32)
        self.eng._chk(self.eng.L.msm_amd_glv_preprocess(self.eng.ctx, 0, pp, ps, n, po, so))
        return po.raw, so.raw

    def gen_points(self, group, seed, n, first=0):
        """n synthetic affine points k_i * G, i = first .. first + n - 1 of the sequence of `seed`, in a new device buffer"""
        buf = self.eng.alloc(max(16, n * self.affine_bytes(group)))
        self.eng._chk(self.eng.L.msm_amd_gen_points_range(self.eng.ctx, self.cid, group, seed, first, n, buf.ptr))
        return buf


def dist_unique_id():
    """128-byte RCCL id (msm_amd_dist_unique_id); rank 0 creates it and hands it to the other ranks."""
    L = load_library()
    out = ctypes.create_string_buffer(128)
    if L.msm_amd_dist_unique_id(out):
        raise MsmAmdError("msm_amd_dist_unique_id failed (RCCL not loadable?)")
    return out.raw


# --- planning without a GPU or a context ---
def plan_prepared(curve, group, n, scalar_bits=0, window_bits=0):
    """msm_amd_plan_prepared: the layout and the size of the table Curve.prepare_bases would build (PreparedBases.info's fields)."""
    L = load_library()
    arr = (ctypes.c_uint32 * 6)()
    tb = ctypes.c_uint64()
    rc = L.msm_amd_plan_prepared(CURVES[curve], group, n, scalar_bits, window_bits, arr, ctypes.byref(tb))
    if rc:
        raise MsmAmdError(f"msm_amd_plan_prepared failed ({rc}): bad curve / group / n / scalar_bits, or windows x points >= 2^31", rc)
    return {"n": arr[0], "scalar_bits": arr[1], "c": arr[2], "windows": arr[3], "signed_windows": arr[4], "buckets": arr[5], "table_bytes": tb.value}


def plan_split(curve, group, scalar_size, n, world, rank):
    """msm_amd_dist_plan without a context: {gp, gw, first, count} of rank `rank` of `world` for a call of n points (Engine.dist_plan
    adds the engine's "dist_gw" option and its own group)."""
    L = load_library()
    if world < 1:
        raise MsmAmdError("plan_split: world >= 1", EINVAL)
    v = [ctypes.c_uint32() for _ in range(4)]
    rc = L.msm_amd_dist_plan(None, CURVES[curve], group, scalar_size, n, world, rank, *[ctypes.byref(x) for x in v])
    if rc:
        raise MsmAmdError(f"msm_amd_dist_plan failed ({rc}): bad world / rank / scalar_size", rc)
    return dict(zip(("gp", "gw", "first", "count"), (x.value for x in v)))


# --- host-only helpers (no GPU, no context): the comparison functions the reference's tests use ---
def combine(curve, group, partials):
    L = load_library()
    jsz = 3 * N8[curve] * group
    cnt = len(partials) // jsz
    out = ctypes.create_string_buffer(jsz)
    rc = L.msm_amd_combine(CURVES[curve], group, bytes(partials), cnt, out)
    if rc:
        raise MsmAmdError(f"msm_amd_combine failed: {rc}")
    return out.raw


def normalize(curve, group, jac):
    L = load_library()
    out = ctypes.create_string_buffer(3 * N8[curve] * group)
    if L.msm_amd_normalize(CURVES[curve], group, bytes(jac), out):
        raise MsmAmdError("msm_amd_normalize failed")
    return out.raw


def to_affine(curve, group, jac):
    L = load_library()
    out = ctypes.create_string_buffer(2 * N8[curve] * group)
    if L.msm_amd_to_affine(CURVES[curve], group, bytes(jac), out):
        raise MsmAmdError("msm_amd_to_affine failed")
    return out.raw


def eq(curve, group, a, b):
    L = load_library()
    rc = L.msm_amd_eq(CURVES[curve], group, bytes(a), bytes(b))
    if rc < 0:
        raise MsmAmdError("msm_amd_eq failed")
    return rc == 1


def glv_decompose_scalar(scalar32):
    """g1m_glv_decomposeScalar: returns (|k1| as int, |k2| as int, sign word)"""
    L = load_library()
    out = ctypes.create_string_buffer(64)
    sign = ctypes.c_uint32()
    if L.msm_amd_glv_decompose_scalar(bytes(scalar32), out, ctypes.byref(sign)):
        raise MsmAmdError("msm_amd_glv_decompose_scalar failed")
    return int.from_bytes(out.raw[:32], "little"), int.from_bytes(out.raw[32:], "little"), sign.value


def glv_split(scalar32):
    """The engine's own centred split (msm_amd_glv_split): returns (|k1|, |k2|, sign word)."""
    L = load_library()
    out = ctypes.create_string_buffer(32)
    sign = ctypes.c_uint32()
    if L.msm_amd_glv_split(bytes(scalar32), out, ctypes.byref(sign)):
        raise MsmAmdError("msm_amd_glv_split failed")
    return int.from_bytes(out.raw[:16], "little"), int.from_bytes(out.raw[16:], "little"), sign.value


def glv_endomorphism(point_affine, sign):
    L = load_library()
    out = ctypes.create_string_buffer(96)
    if L.msm_amd_glv_endomorphism(bytes(point_affine), int(sign), out):
        raise MsmAmdError("msm_amd_glv_endomorphism failed")
    return out.raw


def f_to_montgomery(curve, data):
    L = load_library()
    out = ctypes.create_string_buffer(len(data))
    if L.msm_amd_f_to_montgomery(CURVES[curve], bytes(data), len(data) // N8[curve], out):
        raise MsmAmdError("msm_amd_f_to_montgomery failed")
    return out.raw


def f_from_montgomery(curve, data):
    L = load_library()
    out = ctypes.create_string_buffer(len(data))
    if L.msm_amd_f_from_montgomery(CURVES[curve], bytes(data), len(data) // N8[curve], out):
        raise MsmAmdError("msm_amd_f_from_montgomery failed")
    return out.raw
