"""
ctypes binding of ``libspringcraft_hip.so`` (C ABI: ``include/springcraft_hip.h``).

This is the only place where Python touches native code.  There is **no CPU fallback**: if the
shared library is missing, or no gfx950 device is visible, every compute entry point raises
:class:`HipUnavailableError` — the product path never routes through NumPy or the oracle.
"""

import ctypes as C
import os
import threading
import warnings
from os.path import abspath, dirname, exists, join

import numpy as np

__all__ = ["HipUnavailableError", "lib", "context", "library_path", "EXPORTED_SYMBOLS"]

_PKG = dirname(abspath(__file__))
_LIB_NAME = "libspringcraft_hip.so"

SC_OK = 0
SC_ERR_INVALID_ARG = 1
SC_ERR_INDEX = 2
SC_ERR_SELF_PAIR = 3
SC_ERR_NO_DEVICE = 4
SC_ERR_HIP = 5
SC_ERR_NOMEM = 6
SC_ERR_NOCONV = 7

SC_FF_INVARIANT = 0
SC_FF_HINSEN = 1
SC_FF_PARAMETER_FREE = 2
SC_FF_TABULATED = 3

# every symbol include/springcraft_hip.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "sc_ctx_create", "sc_ctx_create_on_stream", "sc_ctx_destroy", "sc_last_error",
    "sc_ctx_synchronize", "sc_host_alloc", "sc_host_free", "sc_device_info", "sc_contacts", "sc_pairs", "sc_kirchhoff_f64",
    "sc_hessian_f64", "sc_kirchhoff_from_pairs_f64", "sc_hessian_from_pairs_f64", "sc_eigh_f64",
    "sc_anm_eigen_f64", "sc_gnm_eigen_f64", "sc_dev_kirchhoff_f64", "sc_dev_hessian_f64",
    "sc_dev_eigh_f64", "sc_eigh_workspace_bytes", "sc_ctx_set_profiling", "sc_last_eigh_timings",
    "sc_eigh_range_f64", "sc_anm_eigen_range_f64", "sc_dev_eigh_range_f64", "sc_pinvh_f64",
    "sc_modes_from_coord", "sc_modes_from_matrix", "sc_modes_destroy", "sc_modes_order", "sc_modes_get",
    "sc_modes_msf", "sc_modes_dcc", "sc_modes_prs", "sc_ctx_set_two_stage", "sc_last_eigh_phase_ms",
    "sc_ctx_get_counter", "sc_batch_plan_create", "sc_batch_plan_assemble_f64", "sc_batch_plan_order",
    "sc_batch_plan_destroy", "sc_batch_plan_contacts", "sc_batch_plan_pairs", "sc_batch_plan_fill_from_pairs_f64",
]


class HipUnavailableError(RuntimeError):
    """The HIP extension (or an MI355X device) is not available; there is no fallback."""


class TabDesc(C.Structure):
    _fields_ = [
        ("n_bins", C.c_int32),
        ("reserved", C.c_int32),
        ("edges_sq", C.c_void_p),
        ("bonded", C.c_void_p),
        ("intra_chain", C.c_void_p),
        ("inter_chain", C.c_void_p),
        ("atom_type", C.c_void_p),
        ("chain", C.c_void_p),
        ("bonded_next", C.c_void_p),
    ]


class FFDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("has_cutoff", C.c_int32),
        ("cutoff", C.c_double),
        ("cutoff_sq", C.c_double),
        ("tab", C.POINTER(TabDesc)),
    ]


class PatchDesc(C.Structure):
    _fields_ = [
        ("n_shutdown", C.c_int64),
        ("shutdown", C.c_void_p),
        ("n_pair_off", C.c_int64),
        ("pair_off", C.c_void_p),
        ("n_pair_on", C.c_int64),
        ("pair_on", C.c_void_p),
        ("on_force_constants", C.c_void_p),
        ("base_cutoff_masks_gamma", C.c_int32),
        ("reserved", C.c_int32),
    ]


class StructureDesc(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_int64),
        ("ff", C.POINTER(FFDesc)),
        ("patch", C.POINTER(PatchDesc)),
    ]


_lib = None


def library_path():
    return os.environ.get("SPRINGCRAFT_HIP_LIB", join(_PKG, _LIB_NAME))


def _share_hip_runtime_with_torch():
    """
    PyTorch-ROCm wheels bundle their own libamdhip64.so.  A process must not end up with two HIP runtimes (whichever
    initialises second sees no GPU), so when torch is installed its copy is loaded first — without importing torch —
    and libspringcraft_hip.so binds to it, whatever the import order of the two packages.
    """
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = join(dirname(spec.origin), "lib", "libamdhip64.so")
    if exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load (once) and return the ctypes handle, with argument types declared."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not exists(path):
        raise HipUnavailableError(
            f"{path} not found: build it with `python springcraft_amd/csrc/build.py` "
            "(there is no CPU fallback)"
        )
    _share_hip_runtime_with_torch()
    try:
        L = C.CDLL(path)
    except OSError as e:  # e.g. ROCm runtime missing
        raise HipUnavailableError(f"cannot load {path}: {e}") from e

    vp, i64, dbl, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int
    P = C.POINTER
    sig = {
        "sc_ctx_create": (i32, [i32, P(vp)]),
        "sc_ctx_create_on_stream": (i32, [i32, vp, P(vp)]),
        "sc_ctx_destroy": (None, [vp]),
        "sc_last_error": (C.c_char_p, [vp]),
        "sc_ctx_synchronize": (i32, [vp]),
        "sc_host_alloc": (i32, [C.c_size_t, P(vp)]),
        "sc_host_free": (i32, [vp]),
        "sc_device_info": (i32, [vp, C.c_char_p, C.c_size_t]),
        "sc_contacts": (i32, [vp, vp, i64, P(FFDesc), P(PatchDesc), vp, P(i64)]),
        "sc_pairs": (i32, [vp, vp, i64, P(FFDesc), P(PatchDesc), i64, vp, vp, P(i64)]),
        "sc_kirchhoff_f64": (i32, [vp, vp, i64, P(FFDesc), P(PatchDesc), vp, vp]),
        "sc_hessian_f64": (i32, [vp, vp, i64, P(FFDesc), P(PatchDesc), vp, vp]),
        "sc_kirchhoff_from_pairs_f64": (i32, [vp, i64, vp, i64, vp, vp]),
        "sc_hessian_from_pairs_f64": (i32, [vp, vp, i64, vp, i64, vp, vp]),
        "sc_eigh_f64": (i32, [vp, vp, i64, vp, vp]),
        "sc_anm_eigen_f64": (i32, [vp, vp, i64, P(FFDesc), P(PatchDesc), vp, vp, vp]),
        "sc_gnm_eigen_f64": (i32, [vp, vp, i64, P(FFDesc), P(PatchDesc), vp, vp, vp]),
        "sc_dev_kirchhoff_f64": (i32, [vp, vp, i64, i64, P(FFDesc), vp, vp]),
        "sc_dev_hessian_f64": (i32, [vp, vp, i64, i64, P(FFDesc), vp, vp]),
        "sc_dev_eigh_f64": (i32, [vp, vp, i64, i64, vp, vp]),
        "sc_eigh_workspace_bytes": (i64, [i64, i64, i32]),
        "sc_eigh_range_f64": (i32, [vp, vp, i64, i64, i64, vp, vp]),
        "sc_pinvh_f64": (i32, [vp, vp, i64, dbl, vp]),
        "sc_anm_eigen_range_f64": (i32, [vp, vp, i64, P(FFDesc), P(PatchDesc), vp, i64, i64, vp, vp]),
        "sc_dev_eigh_range_f64": (i32, [vp, vp, i64, i64, i64, i64, vp, vp]),
        "sc_ctx_set_profiling": (i32, [vp, i32]),
        "sc_ctx_set_two_stage": (i32, [vp, i32]),
        "sc_last_eigh_timings": (i32, [vp, P(dbl)]),
        "sc_last_eigh_phase_ms": (i32, [vp, C.c_char_p, P(dbl)]),
        "sc_ctx_get_counter": (i32, [vp, C.c_char_p, P(i64)]),
        "sc_batch_plan_create": (i32, [vp, i32, P(StructureDesc), i64, i64, P(vp)]),
        "sc_batch_plan_assemble_f64": (i32, [vp, vp, vp, vp]),
        "sc_batch_plan_contacts": (i32, [vp, vp, vp]),
        "sc_batch_plan_pairs": (i32, [vp, vp, i64, vp, vp, vp]),
        "sc_batch_plan_fill_from_pairs_f64": (i32, [vp, vp, vp, vp, vp, vp, vp]),
        "sc_batch_plan_order": (i64, [vp]),
        "sc_batch_plan_destroy": (None, [vp]),
        "sc_modes_from_coord": (i32, [vp, vp, i64, i32, P(FFDesc), P(PatchDesc), vp, P(vp)]),
        "sc_modes_from_matrix": (i32, [vp, vp, i64, i32, P(vp)]),
        "sc_modes_destroy": (None, [vp]),
        "sc_modes_order": (i64, [vp]),
        "sc_modes_get": (i32, [vp, vp, vp]),
        "sc_modes_msf": (i32, [vp, vp, i64, vp]),
        "sc_modes_dcc": (i32, [vp, vp, i64, i32, vp]),
        "sc_modes_prs": (i32, [vp, dbl, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class Context:
    """Owns one ``sc_ctx`` (device + stream + cached workspace)."""

    def __init__(self, device=0, stream=None):
        L = lib()
        h = C.c_void_p()
        if stream is None:
            rc = L.sc_ctx_create(int(device), C.byref(h))
        else:
            rc = L.sc_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        if rc == SC_ERR_NO_DEVICE:
            raise HipUnavailableError(
                f"no gfx950 (MI355X) device {device} visible to HIP; springcraft_amd has no CPU fallback"
            )
        if rc != SC_OK:
            raise HipUnavailableError(f"sc_ctx_create failed with status {rc}")
        self._h = h
        self._L = L
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def check(self, rc):
        """Map a C status to the exception type the reference raises at the same place."""
        if rc == SC_OK:
            return
        msg = self._L.sc_last_error(self._h)
        msg = msg.decode() if msg else f"status {rc}"
        if rc in (SC_ERR_INVALID_ARG, SC_ERR_SELF_PAIR):
            raise ValueError(msg)
        if rc == SC_ERR_INDEX:
            raise IndexError(msg)
        if rc == SC_ERR_NOMEM:
            raise MemoryError(msg)
        if rc == SC_ERR_NOCONV:   # what np.linalg.eigh raises at nma.py:61 (non-finite input, no convergence)
            raise np.linalg.LinAlgError(msg)
        raise RuntimeError(f"springcraft_hip: {msg} (status {rc})")

    def synchronize(self):
        self.check(self._L.sc_ctx_synchronize(self._h))

    def set_two_stage(self, mode):
        """Eigensolver path: None / -1 automatic, False / 0 one-stage, True / 1 two-stage tridiagonalisation."""
        self.check(self._L.sc_ctx_set_two_stage(self._h, -1 if mode is None else int(mode)))

    def counter(self, name):
        """Event counter of this context (``sc_ctx_get_counter``), e.g. ``"chase_timeouts"``."""
        v = C.c_int64(0)
        if self._L.sc_ctx_get_counter(self._h, name.encode(), C.byref(v)) != SC_OK:
            raise ValueError(f"no counter named {name!r}")
        return int(v.value)

    def info(self):
        buf = C.create_string_buffer(256)
        self.check(self._L.sc_device_info(self._h, buf, 256))
        return buf.value.decode()

    def close(self):
        """
        Destroys the context.  A deferred solver status nobody asked for (``synchronize`` / a solver's ``finish``) would
        be lost with it -- the reference would have raised LinAlgError at nma.py:61 -- so it is turned into a warning.
        """
        if self._h is not None and self._h.value:
            try:
                rc = self._L.sc_ctx_synchronize(self._h)
                if rc != SC_OK:
                    msg = self._L.sc_last_error(self._h)
                    what = ("an unreported solver failure" if rc == SC_ERR_NOCONV
                            else f"an unreported error (status {rc})")
                    warnings.warn(f"springcraft_amd context closed with {what}: "
                                  + (msg.decode() if msg else "Eigenvalues did not converge"), RuntimeWarning, stacklevel=2)
            finally:
                # (the handle is released whatever the warning machinery does, e.g. at interpreter shutdown)
                self._L.sc_ctx_destroy(self._h)
                self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Modes:
    """
    Owns one ``sc_modes``: all eigenpairs of a model, resident in device memory, plus the consumers that
    work on them there (msf, dcc, prs).  ``dim`` is 1 for a GNM and 3 for an ANM.
    """

    def __init__(self, ctx, handle, dim):
        self._ctx = ctx
        self._h = handle
        self._L = lib()
        self.dim = dim
        self.order = int(self._L.sc_modes_order(handle))

    @classmethod
    def from_coord(cls, ctx, coord, dim, ff_desc, patch_desc, inv_sqrt_mass):
        h = C.c_void_p()
        pd = C.byref(patch_desc) if patch_desc is not None else None
        ctx.check(lib().sc_modes_from_coord(ctx.handle, ptr(coord), len(coord), dim, C.byref(ff_desc), pd,
                                            ptr(inv_sqrt_mass), C.byref(h)))
        return cls(ctx, h, dim)

    @classmethod
    def from_matrix(cls, ctx, matrix, dim):
        a = np.ascontiguousarray(matrix, dtype=np.float64)
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError(f"Expected a square matrix, got shape {a.shape}")
        h = C.c_void_p()
        ctx.check(lib().sc_modes_from_matrix(ctx.handle, ptr(a), a.shape[0], dim, C.byref(h)))
        return cls(ctx, h, dim)

    def values(self):
        w = np.empty(self.order)
        self._ctx.check(self._L.sc_modes_get(self._h, ptr(w), None))
        return w

    def eigen(self):
        w = np.empty(self.order)
        v = host_array((self.order, self.order))
        self._ctx.check(self._L.sc_modes_get(self._h, ptr(w), ptr(v)))
        return w, v

    @staticmethod
    def _index_list(mode_idx):
        return np.ascontiguousarray(mode_idx, dtype=np.int64).ravel()

    def msf(self, mode_idx):
        idx = self._index_list(mode_idx)
        out = np.empty(self.order // self.dim)
        self._ctx.check(self._L.sc_modes_msf(self._h, ptr(idx), len(idx), ptr(out)))
        return out

    def dcc(self, mode_idx, norm):
        idx = self._index_list(mode_idx)
        n = self.order // self.dim
        out = host_array((n, n))
        self._ctx.check(self._L.sc_modes_dcc(self._h, ptr(idx), len(idx), int(bool(norm)), ptr(out)))
        return out

    def prs(self, rcond, norm):
        n = self.order // 3
        out = host_array((n, n))
        self._ctx.check(self._L.sc_modes_prs(self._h, float(rcond), int(bool(norm)), ptr(out)))
        return out

    def close(self):
        if self._h is not None and self._h.value:
            self._L.sc_modes_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def context(device=None):
    """Process-wide cached context for ``device`` (default: ``SPRINGCRAFT_HIP_DEVICE`` or 0)."""
    if device is None:
        device = int(os.environ.get("SPRINGCRAFT_HIP_DEVICE", "0"))
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = Context(device)
        _contexts[device] = ctx
    return ctx


# ---- small helpers used by the host modules -------------------------------------------------

def ptr(a):
    """Host pointer of a C-contiguous NumPy array (or None)."""
    if a is None:
        return None
    assert a.flags.c_contiguous
    return C.c_void_p(a.ctypes.data)


# Result arrays on page-locked host memory.  A large result -- the (n, n) eigenvectors above all: 288 MB at N = 2000 --
# crosses PCIe in 5 ms when its destination is page-locked and in 17-25 ms when it is fresh pageable memory
# (tools/d2h_probe.py).  Allocating page-locked memory costs more than that (33 ms for 288 MB), so the blocks are pooled:
# a NumPy array handed out here lives on a block from sc_host_alloc; when the array and all its views are gone, the
# block goes back to the pool and serves the next result of that size.  Bounded on both ends: at most
# ``_PIN_LIVE_MAX`` bytes handed out at a time (beyond that, and for small arrays, ordinary np.empty) and at most
# ``_PIN_POOL_MAX`` bytes kept idle.  SPRINGCRAFT_PINNED_RESULTS=0 turns it off.
_PIN_MIN = 8 << 20
_PIN_LIVE_MAX = 3 << 30
_PIN_POOL_MAX = 1 << 30
_pin_free = {}          # bytes -> [address, ...]
_pin_live = 0
_pin_idle = 0
_pin_lock = threading.RLock()   # re-entrant: a finalizer (_pin_release) may run inside a GC pass started under the lock


def _pin_release(address, nbytes):
    global _pin_live, _pin_idle
    try:
        with _pin_lock:
            _pin_live -= nbytes
            if _pin_idle + nbytes <= _PIN_POOL_MAX:
                _pin_free.setdefault(nbytes, []).append(address)
                _pin_idle += nbytes
                return
        _lib.sc_host_free(C.c_void_p(address))
    except Exception:   # interpreter shutdown: the process' memory goes with it
        pass


def host_array(shape):
    """A C-contiguous float64 array for a result that comes back from the device (see above)."""
    global _pin_live, _pin_idle
    import weakref

    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    count = int(np.prod(shape)) if shape else 1
    nbytes = 8 * count
    if nbytes < _PIN_MIN or os.environ.get("SPRINGCRAFT_PINNED_RESULTS", "1") == "0":
        return np.empty(shape)
    address = None
    with _pin_lock:
        if _pin_live + nbytes > _PIN_LIVE_MAX:
            return np.empty(shape)
        blocks = _pin_free.get(nbytes)
        if blocks:
            address = blocks.pop()
            _pin_idle -= nbytes
        _pin_live += nbytes
    if address is None:
        p = C.c_void_p()
        try:
            rc = lib().sc_host_alloc(nbytes, C.byref(p))
        except Exception:
            rc = 1
        if rc != SC_OK or not p.value:
            with _pin_lock:
                _pin_live -= nbytes
            return np.empty(shape)
        address = p.value
    block = (C.c_double * count).from_address(address)
    weakref.finalize(block, _pin_release, address, nbytes)   # runs when the last array / view on the block is gone
    return np.frombuffer(block, dtype=np.float64).reshape(shape)


def make_ff_desc(kind, cutoff_distance):
    d = FFDesc()
    d.kind = kind
    if cutoff_distance is None:
        d.has_cutoff = 0
        d.cutoff = float("nan")
        d.cutoff_sq = float("nan")
    else:
        d.has_cutoff = 1
        d.cutoff = float(cutoff_distance)
        # interaction.py:166 squares the cutoff in Python / NumPy float64 arithmetic
        d.cutoff_sq = float(np.float64(cutoff_distance) ** 2)
    return d


def make_tab_desc(ff_desc, edges, bonded, intra, inter, atom_type, chain, bonded_next):
    """
    Attach the tables of a TabulatedForceField to ``ff_desc`` (kind SC_FF_TABULATED).  The arrays are kept
    alive on the descriptor object itself (``ff_desc._keep``).
    """
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    keep = [f32(bonded), f32(intra), f32(inter),
            np.ascontiguousarray(atom_type, dtype=np.int32),
            np.ascontiguousarray(chain, dtype=np.int32),
            np.ascontiguousarray(bonded_next, dtype=np.uint8)]
    t = TabDesc()
    t.n_bins = keep[0].shape[-1]
    if edges is not None:
        e2 = np.ascontiguousarray(np.asarray(edges, dtype=np.float64) ** 2)   # forcefield.py:521 squares the edges
        keep.append(e2)
        t.edges_sq = e2.ctypes.data
    t.bonded, t.intra_chain, t.inter_chain = (k.ctypes.data for k in keep[:3])
    t.atom_type, t.chain, t.bonded_next = (k.ctypes.data for k in keep[3:6])
    keep.append(t)
    ff_desc.kind = SC_FF_TABULATED
    ff_desc.tab = C.pointer(t)
    ff_desc._keep = keep
    return ff_desc


def make_patch_desc(shutdown, pair_off, pair_on, on_force_constants, mask_gamma, keep):
    """
    Build an ``sc_patch_desc``; ``keep`` is a list that receives the arrays whose memory the
    descriptor points into (so they outlive the call).
    """
    if shutdown is None and pair_off is None and pair_on is None:
        return None
    d = PatchDesc()

    def arr(x, cols):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.int64))
        a = a.reshape(-1, cols) if cols > 1 else a.reshape(-1)
        keep.append(a)
        return a

    if shutdown is not None:
        a = arr(shutdown, 1)
        d.n_shutdown, d.shutdown = len(a), a.ctypes.data
    if pair_off is not None:
        a = arr(pair_off, 2)
        d.n_pair_off, d.pair_off = len(a), a.ctypes.data
    if pair_on is not None:
        a = arr(pair_on, 2)
        d.n_pair_on, d.pair_on = len(a), a.ctypes.data
        if on_force_constants is not None:
            g = np.ascontiguousarray(np.asarray(on_force_constants, dtype=np.float64)).reshape(-1)
            if len(g) != len(a):
                raise IndexError(
                    f"{len(g)} force constants were given for {len(a)} switched on contact_pairs"
                )
            keep.append(g)
            d.on_force_constants = g.ctypes.data
    d.base_cutoff_masks_gamma = 1 if mask_gamma else 0
    return d
