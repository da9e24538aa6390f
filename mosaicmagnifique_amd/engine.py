"""Numpy-facing wrapper of the C ABI (one engine = one GPU).

Nothing is computed here: every method marshals numpy arrays into the entry
points of include/mosaic_hip.h and turns a non-zero ``mm_status`` into
``EngineError``.
"""
import ctypes

import numpy as np

from . import _lib
from .grid import CELL_DTYPE


class EngineError(RuntimeError):
    def __init__(self, status, message):
        self.status = status
        name = _lib.STATUS_NAMES[status] if 0 <= status < len(_lib.STATUS_NAMES) else str(status)
        super().__init__("%s: %s" % (name, message))


class Cancelled(EngineError):
    pass


def device_count():
    n = ctypes.c_int(0)
    st = _lib.load().mm_device_count(ctypes.byref(n))
    if st != _lib.MM_OK:
        raise EngineError(st, (_lib.load().mm_last_error(None) or b"").decode())
    return n.value


def has_pretier():
    """mm_debug_has_pretier: 1 if the library was built with the packed-fp16 pre-tier (-DMM_WITH_PRETIER=1; default: not)."""
    return bool(_lib.load().mm_debug_has_pretier())


def trim_cache():
    """mm_trim_cache: release the device resources destroyed engines left parked; returns the number of sets released."""
    return _lib.load().mm_trim_cache()


class Engine:
    def __init__(self, device=0):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        st = self._lib.mm_create(int(device), ctypes.byref(self._h))
        if st != _lib.MM_OK:
            raise EngineError(st, (self._lib.mm_last_error(None) or b"").decode())
        self.device = int(device)
        self.n_library = 0
        self.size = 0
        self.options = {}           # what set_option was given (the C ABI has no getter)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mm_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st):
        if st == _lib.MM_OK:
            return
        msg = (self._lib.mm_last_error(self._h) or b"").decode()
        raise (Cancelled if st == _lib.MM_CANCELLED else EngineError)(st, msg)

    # -- staging -------------------------------------------------------------
    def set_library(self, images):
        """f32 ``[n][size][size][3]`` (after preprocessLibraryImages)."""
        a = np.ascontiguousarray(images, dtype=np.float32)
        if a.ndim != 4 or a.shape[1] != a.shape[2] or a.shape[3] != 3:
            raise ValueError("library must be [n][size][size][3]")
        self._check(self._lib.mm_set_library(self._h, _lib.ptr(a), a.shape[0], a.shape[1]))
        self.n_library, self.size = a.shape[0], a.shape[1]

    def set_masks(self, masks):
        """u8 ``[4][size][size]`` in the order flipH + 2*flipV."""
        a = np.ascontiguousarray(masks, dtype=np.uint8)
        if a.ndim != 3 or a.shape[0] != 4 or a.shape[1] != a.shape[2]:
            raise ValueError("masks must be [4][size][size]")
        self._check(self._lib.mm_set_masks(self._h, _lib.ptr(a), a.shape[1]))

    def set_target(self, image):
        """f32 ``[rows][cols][3]`` (after preprocessMainImage)."""
        a = np.ascontiguousarray(image, dtype=np.float32)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("target must be [rows][cols][3]")
        self._check(self._lib.mm_set_target(self._h, _lib.ptr(a), a.shape[0], a.shape[1]))

    # -- pre-processing, detail and size steps on the device -------------------
    def set_library_u8(self, images, diff_type, shrink=1):
        """u8 BGR ``[n][size][size][3]`` as setLibrary receives it; resized to ``size / shrink``
        and colour-converted on the device (preprocessLibraryImages)."""
        a = np.ascontiguousarray(images, dtype=np.uint8)
        if a.ndim != 4 or a.shape[1] != a.shape[2] or a.shape[3] != 3:
            raise ValueError("library must be [n][size][size][3]")
        self._check(self._lib.mm_set_library_u8(self._h, _lib.ptr(a), a.shape[0], a.shape[1], int(shrink), int(diff_type)))
        self.n_library, self.size = a.shape[0], a.shape[1] // int(shrink)

    def set_target_u8(self, image, diff_type):
        """u8 BGR ``[rows][cols][3]``; colour-converted on the device (preprocessMainImage)."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("target must be [rows][cols][3]")
        self._check(self._lib.mm_set_target_u8(self._h, _lib.ptr(a), a.shape[0], a.shape[1], int(diff_type)))

    def hue_rotate_u8(self, image, degrees):
        """One colour-scheme variant (ColourScheme.cpp:44-177): the u8 BGR image with its hue
        rotated by ``degrees`` through f32 HSV_FULL, on the device."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("image must be [rows][cols][3]")
        out = np.empty_like(a)
        self._check(self._lib.mm_hue_rotate_u8(self._h, _lib.ptr(a), a.shape[0], a.shape[1], float(degrees), _lib.ptr(out)))
        return out

    def shrink_library(self, factor=2):
        """batchResizeMat between size steps: the resident library at 1/factor of its size."""
        self._check(self._lib.mm_shrink_library(self._h, int(factor)))
        self.size //= int(factor)

    def set_library_u8_sized(self, images, diff_type, detail_size):
        """u8 BGR library resized on the device to ANY detail cell size <= its own (general
        INTER_AREA where the ratio is not whole) and colour-converted (preprocessLibraryImages)."""
        a = np.ascontiguousarray(images, dtype=np.uint8)
        if a.ndim != 4 or a.shape[1] != a.shape[2] or a.shape[3] != 3:
            raise ValueError("library must be [n][size][size][3]")
        self._check(self._lib.mm_set_library_u8_sized(self._h, _lib.ptr(a), a.shape[0], a.shape[1], int(detail_size), int(diff_type)))
        self.n_library, self.size = a.shape[0], int(detail_size)

    def set_cell_resize(self, normal_size, detail_size):
        """Any detail level: cells of ``normal_size`` pixels are resized to ``detail_size`` (INTER_AREA)."""
        self._check(self._lib.mm_set_cell_resize(self._h, int(normal_size), int(detail_size)))

    def set_cell_scale(self, factor):
        """Detail level 1/factor: cells are box-averaged from factor x factor image pixels."""
        self._check(self._lib.mm_set_cell_scale(self._h, int(factor)))

    def get_library(self):
        n, size = ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.mm_library_shape(self._h, ctypes.byref(n), ctypes.byref(size)))
        out = np.empty((n.value, size.value, size.value, 3), dtype=np.float32)
        self._check(self._lib.mm_get_library(self._h, _lib.ptr(out)))
        return out

    def get_target(self, rows, cols):
        out = np.empty((rows, cols, 3), dtype=np.float32)
        self._check(self._lib.mm_get_target(self._h, _lib.ptr(out)))
        return out

    # -- hot path ------------------------------------------------------------
    def _callbacks(self, progress, cancel):
        cb = _lib.PROGRESS_CB(lambda done, _u: progress(done)) if progress else None
        flag = cancel if cancel is not None else None      # ctypes.c_int owned by the caller
        return cb, (ctypes.byref(flag) if flag is not None else None)

    def generate(self, cells, grid_w, grid_h, diff_type, repeat_range=0, repeat_addition=0,
                 want_scores=True, progress=None, cancel=None, out=None):
        """mm_generate -> (best_fit int64 [grid_h][grid_w] with sentinel n_library,
        best_score f64 [grid_h][grid_w] or None).  out = (best_fit, best_score) arrays to fill
        instead of fresh ones."""
        rec = cells if (isinstance(cells, np.ndarray) and cells.dtype == CELL_DTYPE and cells.flags["C_CONTIGUOUS"]) \
            else _lib.as_cells(np.asarray(cells, dtype=CELL_DTYPE))
        if out is not None:
            best, score = out
            assert best.shape == (grid_h, grid_w) and best.dtype == np.int64 and best.flags["C_CONTIGUOUS"]
            assert score is None or (score.shape == (grid_h, grid_w) and score.dtype == np.float64 and score.flags["C_CONTIGUOUS"])
        else:
            best = np.empty((grid_h, grid_w), dtype=np.int64)
            score = np.empty((grid_h, grid_w), dtype=np.float64) if want_scores else None
        cb, cflag = self._callbacks(progress, cancel)
        self._check(self._lib.mm_generate(
            self._h, _lib.ptr(rec), len(rec), grid_w, grid_h, int(diff_type),
            int(repeat_range), int(repeat_addition), _lib.ptr(best), _lib.ptr(score),
            ctypes.cast(cb, ctypes.c_void_p) if cb else None, None, cflag))
        return best, score

    def candidates_per_cell(self, repeat_range, repeat_addition):
        return self._lib.mm_candidates_per_cell(self.n_library, int(repeat_range), int(repeat_addition))

    def speculative_candidates(self, repeat_range, repeat_addition, spec_k=4):
        """Length of the short lists worth scoring first (= candidates_per_cell when speculation does not pay)."""
        return self._lib.mm_speculative_candidates(self.n_library, int(repeat_range), int(repeat_addition), int(spec_k))

    def score_cells(self, cells, diff_type, k, out_index=None, out_score=None, on_device=False,
                    progress=None, cancel=None, k_fill=None):
        """mm_score_cells[_partial].  Host mode returns (index int32 [n][k], score f64 [n][k]);
        device mode writes to the given raw device addresses (ints).  k_fill < k fills only the
        first k_fill entries of each k-wide row (speculative short lists)."""
        rec = _lib.as_cells(np.asarray(cells, dtype=CELL_DTYPE))
        cb, cflag = self._callbacks(progress, cancel)
        if on_device:
            pi, ps = ctypes.c_void_p(int(out_index)), ctypes.c_void_p(int(out_score))
            idx = sc = None
        else:
            idx = np.empty((len(rec), k), dtype=np.int32)
            sc = np.empty((len(rec), k), dtype=np.float64)
            pi, ps = _lib.ptr(idx), _lib.ptr(sc)
        self._check(self._lib.mm_score_cells_partial(
            self._h, _lib.ptr(rec), len(rec), int(diff_type), int(k), int(k if k_fill is None else k_fill), pi, ps,
            1 if on_device else 0, ctypes.cast(cb, ctypes.c_void_p) if cb else None, None, cflag))
        return idx, sc

    def resolve(self, cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition, k,
                cand_index, cand_score, on_device=False, want_scores=True, deep=None, k_fill=None):
        """mm_resolve[_verify] over ALL cells of the grid with their gathered candidate lists.
        With ``deep`` (u8 per cell: 1 = full-length list) the picks of the other cells are verified
        and a third value is returned: u8 per cell, 1 = could not be verified (deepen and retry)."""
        rec = _lib.as_cells(np.asarray(cells, dtype=CELL_DTYPE))
        best = np.empty((grid_h, grid_w), dtype=np.int64)
        score = np.empty((grid_h, grid_w), dtype=np.float64) if want_scores else None
        if on_device:
            pi, ps = ctypes.c_void_p(int(cand_index)), ctypes.c_void_p(int(cand_score))
        else:
            ci = np.ascontiguousarray(cand_index, dtype=np.int32)
            cs = np.ascontiguousarray(cand_score, dtype=np.float64)
            pi, ps = _lib.ptr(ci), _lib.ptr(cs)
        if deep is None:
            self._check(self._lib.mm_resolve(
                self._h, _lib.ptr(rec), len(rec), grid_w, grid_h, int(diff_type), int(repeat_range),
                int(repeat_addition), int(k), pi, ps, 1 if on_device else 0, _lib.ptr(best), _lib.ptr(score)))
            return best, score
        deep = np.ascontiguousarray(deep, dtype=np.uint8)
        insufficient = np.zeros(len(rec), dtype=np.uint8)
        self._check(self._lib.mm_resolve_verify(
            self._h, _lib.ptr(rec), len(rec), grid_w, grid_h, int(diff_type), int(repeat_range),
            int(repeat_addition), int(k), int(k if k_fill is None else k_fill), pi, ps, 1 if on_device else 0,
            _lib.ptr(deep), _lib.ptr(insufficient), _lib.ptr(best), _lib.ptr(score)))
        return best, score, insufficient

    def set_library_bgr(self, images):
        """u8 BGR ``[n][size][size][3]`` as setLibrary received it (for build_photomosaic)."""
        a = np.ascontiguousarray(images, dtype=np.uint8)
        if a.ndim != 4 or a.shape[1] != a.shape[2] or a.shape[3] != 3:
            raise ValueError("library must be [n][size][size][3]")
        self._check(self._lib.mm_set_library_bgr(self._h, _lib.ptr(a), a.shape[0], a.shape[1]))

    def build_photomosaic(self, cells, grid_w, grid_h, best_fit, rows, cols, background=(0, 0, 0, 0),
                          step_index=0, want_output=True):
        """mm_build_photomosaic[_step] -> u8 BGRA [rows][cols][4] (None if not wanted).
        step_index > 0 continues the mosaic of the earlier steps."""
        rec = _lib.as_cells(np.asarray(cells, dtype=CELL_DTYPE))
        fit = np.ascontiguousarray(best_fit, dtype=np.int64)
        assert fit.shape == (grid_h, grid_w)
        bg = np.array(background, dtype=np.uint8)
        out = np.empty((rows, cols, 4), dtype=np.uint8) if want_output else None
        self._check(self._lib.mm_build_photomosaic_step(
            self._h, _lib.ptr(rec), len(rec), grid_w, grid_h, _lib.ptr(fit), _lib.ptr(bg), rows, cols,
            int(step_index), _lib.ptr(out) if want_output else None))
        return out

    def shrink_library_bgr(self, factor=2):
        """batchResizeMat on the u8 library between the size steps of buildPhotomosaic."""
        self._check(self._lib.mm_shrink_library_bgr(self._h, int(factor)))

    def set_main_image_bgr(self, image):
        """The u8 BGR main image as setMainImage receives it, kept on the device for ``cell_entropy``."""
        a = np.ascontiguousarray(image, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("image must be [rows][cols][3]")
        self._check(self._lib.mm_set_main_image_bgr(self._h, _lib.ptr(a), a.shape[0], a.shape[1]))

    def cell_entropy(self, cells, normal_size, want_entropy=False):
        """mm_cell_entropy -> u8 [n] (1 = the cell's entropy reaches 0.7 * MAX_ENTROPY), and the entropies if wanted."""
        rec = _lib.as_cells(np.asarray(cells, dtype=CELL_DTYPE))
        exceeded = np.zeros(len(rec), dtype=np.uint8)
        entropy = np.zeros(len(rec), dtype=np.float64) if want_entropy else None
        self._check(self._lib.mm_cell_entropy(self._h, _lib.ptr(rec), len(rec), int(normal_size), _lib.ptr(exceeded), _lib.ptr(entropy)))
        return (exceeded, entropy) if want_entropy else exceeded

    def set_option(self, name, value):
        """mm_set_option: "filter" 1 (default, two-tier fp32 filter + fp64 re-score) or 0 (fp64 only)."""
        self._check(self._lib.mm_set_option(self._h, name.encode(), int(value)))
        self.options[name] = int(value)

    def stream_read_gbs(self, nbytes=4 << 30, repeats=5):
        """Calibration: GB/s of a coalesced float4 read stream over ``nbytes`` of device memory."""
        out = ctypes.c_double(0.0)
        self._check(self._lib.mm_debug_stream_read(self._h, int(nbytes), int(repeats), ctypes.byref(out)))
        return out.value

    def filter_band_sweep(self, mode, n_pairs, seed=0):
        """mm_debug_filter_band_sweep -> dict(max_err, max_abs_part, band_share, max_value, handed_over, measured)."""
        out = (ctypes.c_double * 6)()
        self._check(self._lib.mm_debug_filter_band_sweep(self._h, int(mode), int(seed), int(n_pairs), out))
        return dict(zip(("max_err", "max_abs_part", "band_share", "max_value", "handed_over", "measured"), list(out)))

    def timing(self):
        return self.timing_raw().as_dict()

    def stage_timing(self):
        """mm_get_stage_timing: host wall clocks of everything that is not the hot path, since the engine was created."""
        t = _lib.MMStageTiming()
        self._check(self._lib.mm_get_stage_timing(self._h, ctypes.byref(t)))
        return t.as_dict()

    def timing_raw(self):
        """mm_get_timing as the ctypes struct (cheap enough to call after every step of a timed loop)."""
        t = _lib.MMTiming()
        self._check(self._lib.mm_get_timing(self._h, ctypes.byref(t)))
        return t

    def colour_diff_batch(self, diff_type, first, second):
        """n pairs of f32 triplets -> f64 differences (first = main pixel, second = library pixel)."""
        a = np.ascontiguousarray(first, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(second, dtype=np.float32).reshape(-1, 3)
        if a.shape != b.shape:
            raise ValueError("pair arrays differ in shape")
        out = np.empty(a.shape[0], dtype=np.float64)
        self._check(self._lib.mm_colour_diff_batch(self._h, int(diff_type), _lib.ptr(a), _lib.ptr(b),
                                                   a.shape[0], _lib.ptr(out)))
        return out

    def filter_diff_batch(self, diff_type, first, second):
        """The fp32 filter form of the difference (diagnostics; never decides a result)."""
        a = np.ascontiguousarray(first, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(second, dtype=np.float32).reshape(-1, 3)
        out = np.empty(a.shape[0], dtype=np.float32)
        self._check(self._lib.mm_filter_diff_batch(self._h, int(diff_type), _lib.ptr(a), _lib.ptr(b),
                                                   a.shape[0], _lib.ptr(out)))
        return out


class EngineGroup:
    """mm_group_*: one process, one engine per listed device, the cell-sharded generate with ONE RCCL all-gather of
    the candidate lists behind the C ABI (host merge when RCCL refuses the device list, e.g. one device twice)."""

    def __init__(self, devices):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        st = self._lib.mm_group_create(devs, len(devices), ctypes.byref(self._h))
        if st != _lib.MM_OK:
            raise EngineError(st, (self._lib.mm_group_last_error(None) or b"").decode())
        self.devices = [int(d) for d in devices]
        self.n_library = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mm_group_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st):
        if st == _lib.MM_OK:
            return
        msg = (self._lib.mm_group_last_error(self._h) or b"").decode()
        raise (Cancelled if st == _lib.MM_CANCELLED else EngineError)(st, msg)

    @property
    def uses_rccl(self):
        return bool(self._lib.mm_group_uses_rccl(self._h))

    def __len__(self):
        return self._lib.mm_group_size(self._h)

    def timing(self, index=0):
        t = _lib.MMTiming()
        st = self._lib.mm_get_timing(self._lib.mm_group_engine(self._h, int(index)), ctypes.byref(t))
        self._check(st)
        return t.as_dict()

    def set_library(self, images):
        a = np.ascontiguousarray(images, dtype=np.float32)
        self._check(self._lib.mm_group_set_library(self._h, _lib.ptr(a), a.shape[0], a.shape[1]))
        self.n_library = a.shape[0]

    def set_library_u8(self, images, diff_type, detail_size=None):
        a = np.ascontiguousarray(images, dtype=np.uint8)
        self._check(self._lib.mm_group_set_library_u8(self._h, _lib.ptr(a), a.shape[0], a.shape[1],
                                                      int(detail_size or a.shape[1]), int(diff_type)))
        self.n_library = a.shape[0]

    def set_masks(self, masks):
        a = np.ascontiguousarray(masks, dtype=np.uint8)
        self._check(self._lib.mm_group_set_masks(self._h, _lib.ptr(a), a.shape[1]))

    def set_target(self, image):
        a = np.ascontiguousarray(image, dtype=np.float32)
        self._check(self._lib.mm_group_set_target(self._h, _lib.ptr(a), a.shape[0], a.shape[1]))

    def set_target_u8(self, image, diff_type):
        a = np.ascontiguousarray(image, dtype=np.uint8)
        self._check(self._lib.mm_group_set_target_u8(self._h, _lib.ptr(a), a.shape[0], a.shape[1], int(diff_type)))

    def set_option(self, name, value):
        self._check(self._lib.mm_group_set_option(self._h, name.encode(), int(value)))

    def generate(self, cells, grid_w, grid_h, diff_type, repeat_range=0, repeat_addition=0, want_scores=True):
        rec = _lib.as_cells(np.asarray(cells, dtype=CELL_DTYPE))
        best = np.empty((grid_h, grid_w), dtype=np.int64)
        score = np.empty((grid_h, grid_w), dtype=np.float64) if want_scores else None
        self._check(self._lib.mm_group_generate(self._h, _lib.ptr(rec), len(rec), grid_w, grid_h, int(diff_type),
                                                int(repeat_range), int(repeat_addition), _lib.ptr(best), _lib.ptr(score),
                                                None, None, None))
        return best, score
