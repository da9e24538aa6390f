"""``HIPPhotomosaicGenerator``: the reference's generator interface over the HIP engine.

Same names, argument meaning and error behaviour as ``PhotomosaicGeneratorBase``
(src/Photomosaic/PhotomosaicGeneratorBase.h:36-81) and its device back-end
``CUDAPhotomosaicGenerator(device)`` (CUDA/CUDAPhotomosaicGenerator.h:31), so a
test written against the reference reads the same here.  ``cv::Mat`` becomes a
numpy array, ``MosaicBestFit`` a list (one entry per size step) of int64 arrays
over the padded grid with ``grid.NULLOPT`` for ``std::nullopt``.

All arithmetic of the hot path happens behind the C ABI; this class only does
what the reference's host does around it: colour conversion of the inputs,
marshalling, and writing the result back where the grid state ``has_value()``
(CUDAPhotomosaicGenerator.cpp:310-324).
"""
import ctypes
import logging

import numpy as np

from . import colourdifference as ColourDifference
from . import colourscheme as ColourScheme
from . import grid as GridUtility
from . import synth
from .cellshape import CellGroup
from .engine import Cancelled, Engine, EngineError

log = logging.getLogger("mosaicmagnifique_amd")


class HIPPhotomosaicGenerator:
    def __init__(self, device=0):
        self.m_engine = Engine(device)
        self.m_img = None
        self.m_lib = None
        self.m_colourDiffType = ColourDifference.Type.RGB_EUCLIDEAN
        self.m_colourSchemeType = ColourScheme.Type.NONE
        self.m_cells = CellGroup()
        self.m_bestFits = []
        self.m_repeatRange = 0
        self.m_repeatAddition = 0
        self.m_progress = 0
        self.m_wasCanceled = ctypes.c_int(0)
        self.m_bestScores = []
        self._progress_listeners = []
        # What is resident on the device is tracked by a counter every setter bumps (the C++
        # mirror's m_staleInputs), never by object identity: an array edited in place or a freed
        # id() reused by a new array must not leave stale data on the GPU.
        self._inputs_version = 0
        self._staged = None          # _inputs_version the resident library / target were staged at
        self._staged_bgr = None
        self._masks_key = None       # (_inputs_version, size step) the resident masks belong to
        self._lib_step = 0          # size step the resident library is at (it is halved per step)
        self._rows = self._cols = 0
        self._cells_version = 0      # bumped by setCellGroup / invalidate
        self._cell_cache = {}        # step -> (key, mm_cell records): the integer geometry of a step's valid cells
        self.lastError = ""

    # -- setters (PhotomosaicGeneratorBase.cpp:44-93) -------------------------
    def setMainImage(self, img):
        """u8 BGR ``[rows][cols][3]`` (converted like preprocessMainImage) or an
        already preprocessed f32 array."""
        self.m_img = np.asarray(img)
        self._inputs_version += 1

    def setLibrary(self, lib):
        """Sequence / array of square u8 BGR images at the cell size, or f32 preprocessed."""
        self.m_lib = np.asarray(lib)
        self._inputs_version += 1

    def setColourDifference(self, t):
        t = ColourDifference.Type(t)
        if t == ColourDifference.Type.MAX:
            raise ValueError("No function for given type")     # ColourDifference.cpp:23
        self.m_colourDiffType = t
        self._inputs_version += 1

    def setColourScheme(self, t=ColourScheme.Type.NONE):
        """PhotomosaicGeneratorBase::setColourScheme (PhotomosaicGeneratorBase.cpp:64-69).

        preprocessMainImage builds one main image per variant of the scheme (:231-232), and
        findCellBestFit scans every variant's cell per library image (CPUPhotomosaicGenerator.cpp:
        139).  But getCellAt creates the V cells as V copies of ONE ``cv::Mat`` header
        (``std::vector<cv::Mat> cells(n, cv::Mat(...))``, PhotomosaicGeneratorBase.cpp:310), so
        they share one pixel buffer and each ``copyTo`` overwrites the previous variant (:311-312;
        a later resize copies that same content): every cell the scan sees holds the LAST
        variant.  The reference's result with a colour scheme is therefore exactly its result
        with the last variant as the main image -- on the CPU and on the CUDA path alike -- and
        that is what this generator computes (identical indices; tests/test_gpu_colour_scheme.py).
        The variant list itself is available from ``mainImageVariants``."""
        t = ColourScheme.Type(t)
        if t == ColourScheme.Type.MAX:
            raise ValueError("No function for given type")      # ColourScheme.cpp:31
        self.m_colourSchemeType = t
        self._inputs_version += 1

    def setCellGroup(self, cellGroup):
        """A COPY, as the reference's setter takes one (PhotomosaicGeneratorBase.cpp:72-75 assigns the CellGroup by
        value): a shape swapped into the caller's group afterwards does not reach masks already staged."""
        import copy
        self.m_cells = copy.deepcopy(cellGroup)
        self._inputs_version += 1
        self._cells_version += 1

    def invalidate(self):
        """Call after editing an input array or the cell group IN PLACE: the next generate re-stages."""
        self._inputs_version += 1
        self._cells_version += 1

    def getCellGroup(self):
        return self.m_cells

    def gridStateOnDevice(self):
        """GridGenerator::getGridState(cellGroup, mainImage) with the entropy test of every size step on this
        generator's device (mm_cell_entropy); needs the u8 BGR main image and the cell group."""
        if self.m_img is None or self.m_img.dtype != np.uint8:
            raise ValueError("the entropy subdivision needs the u8 BGR main image")
        state = GridUtility.getGridState(self.m_cells, self.m_img.shape[0], self.m_img.shape[1], mainImage=self.m_img,
                                         engine=self.m_engine)
        self._masks_key = None              # the engine now holds some step's detail masks
        return state

    def setGridState(self, gridState):
        self.m_bestFits = [np.array(s, dtype=np.int64) for s in gridState]

    def setRepeat(self, repeatRange=0, repeatAddition=0):
        self.m_repeatRange = int(repeatRange)
        self.m_repeatAddition = int(repeatAddition)

    def mainImageVariants(self):
        """m_colourSchemeFunc(m_img) (PhotomosaicGeneratorBase.cpp:231-232): the u8 main image
        followed by its hue-rotated variants, rotated on the device."""
        if self.m_img is None or self.m_img.dtype != np.uint8:
            raise ValueError("colour-scheme variants need the u8 BGR main image")
        return ColourScheme.getFunction(self.m_colourSchemeType)(self.m_img, rotate=self.m_engine.hue_rotate_u8)

    # -- progress / cancel ---------------------------------------------------
    def connectProgress(self, fn):
        """The ``progress(int)`` signal."""
        self._progress_listeners.append(fn)

    def getMaxProgress(self):
        """PhotomosaicGeneratorBase::getMaxProgress (:210-214)."""
        s = len(self.m_bestFits)
        return int(4 ** (s - 1)) * s * self.m_bestFits[0].shape[1] * self.m_bestFits[0].shape[0]

    def cancel(self):
        self.m_wasCanceled.value = 1

    # -- preprocessing (PhotomosaicGeneratorBase.cpp:223-290) -----------------
    def _stage(self):
        """Inputs onto the device.  u8 BGR inputs are resized (library: to the detail cell
        size, on the u8 image as the reference does, :262-271) and colour-converted there;
        f32 inputs are taken as already preprocessed (library at the detail cell size)."""
        t = self.m_colourDiffType
        detail_size = self.m_cells.getCellSize(0, True)
        if self.m_lib.dtype == np.uint8:
            if self.m_lib.shape[1] < detail_size:
                raise NotImplementedError("library images of %d px below the detail cell of %d px: INTER_CUBIC "
                                          "enlargement is outside the hot path" % (self.m_lib.shape[1], detail_size))
            self.m_engine.set_library_u8_sized(self.m_lib, int(t), detail_size)
        else:
            if self.m_lib.shape[1] != detail_size:
                raise ValueError("a preprocessed (f32) library must be at the detail cell size %d" % detail_size)
            self.m_engine.set_library(np.ascontiguousarray(self.m_lib, dtype=np.float32))
        if self.m_colourSchemeType != ColourScheme.Type.NONE:
            # every variant's cell aliases the last variant (setColourScheme): compare against that one
            self.m_engine.set_target_u8(self.mainImageVariants()[-1], int(t))
        elif self.m_img.dtype == np.uint8:
            self.m_engine.set_target_u8(self.m_img, int(t))
        else:
            self.m_engine.set_target(np.ascontiguousarray(self.m_img, dtype=np.float32))
        self._rows, self._cols = self.m_img.shape[0], self.m_img.shape[1]

    def preprocessMainImage(self):
        """The main image as the engine holds it after preprocessMainImage (f32 BGR / Lab)."""
        self._ensure_staged()
        return self.m_engine.get_target(self._rows, self._cols)

    def preprocessLibraryImages(self):
        """The library as the engine holds it after preprocessLibraryImages: f32, at the detail
        cell size of size step 0."""
        self._ensure_staged(force=self._lib_step != 0)
        return self.m_engine.get_library()

    def _ensure_staged(self, force=False):
        key = (self._inputs_version, self.m_cells.getDetail(), self.m_cells.getSizeSteps())
        if force or key != self._staged:
            self._stage()
            self._staged = key
            self._lib_step = 0

    # -- the hot path --------------------------------------------------------
    def generateBestFits(self):
        """Returns True if successful, False on cancel or device error (the
        reference's bool contract, CUDAPhotomosaicGenerator.cpp:367); the error
        text is in ``lastError`` (the reference shows it in a message box)."""
        try:
            return self._generate()
        except Cancelled:
            log.info("Photomosaic generation cancelled.")
            return False
        except EngineError as err:
            self.lastError = str(err)
            log.critical(self.lastError)
            return False

    def _generate(self):
        """The step loop of generateBestFits (CPUPhotomosaicGenerator.cpp:46-100): per size
        step the step's cell shapes, the grid of that step, and a library halved once more."""
        steps = len(self.m_bestFits)
        if steps > self.m_cells.getSizeSteps() + 1:
            raise ValueError("grid state has more size steps than the cell group")
        self._ensure_staged(force=self._lib_step != 0)
        self.m_progress = 0
        listeners = self._progress_listeners
        out_fits, out_scores = [], []
        for step in range(steps):
            if self.m_wasCanceled.value:
                raise Cancelled(3, "cancelled")
            if step > 0:                                # ImageUtility::batchResizeMat(lib), :95-99
                if self.m_engine.size % 2 != 0:
                    raise NotImplementedError("size step %d halves an odd library size %d" % (step, self.m_engine.size))
                self.m_engine.shrink_library(2)
                self._lib_step = step
            detailShape = self.m_cells.getCell(step, True)
            if detailShape.getSize() != self.m_engine.size:
                raise NotImplementedError("detail cell of %d px against a %d px library at step %d"
                                          % (detailShape.getSize(), self.m_engine.size, step))
            masks_key = (self._staged, step)
            if self._masks_key != masks_key:
                self.m_engine.set_masks(detailShape.masks4())
                self._masks_key = masks_key
            self.m_engine.set_cell_resize(self.m_cells.getCellSize(step), detailShape.getSize())
            state = self.m_bestFits[step]
            cells = self._cells_of(step, state)
            # progress counts every grid position, weighted 4^(steps-1-step) (:54,:85-88)
            weight = int(4 ** (steps - 1 - step))
            base = self.m_progress

            def on_progress(done, base=base, weight=weight):
                self.m_progress = base + done * weight
                for fn in listeners:
                    fn(self.m_progress)

            best, score = self.m_engine.generate(
                cells, state.shape[1], state.shape[0], int(self.m_colourDiffType),
                self.m_repeatRange, self.m_repeatAddition, want_scores=True,
                progress=on_progress if listeners else None, cancel=self.m_wasCanceled)
            self.m_progress = base + state.size * weight

            # Write back only where the grid state has a value (CUDAPhotomosaicGenerator.cpp:310-324).
            valid = state >= 0
            out = state.copy()
            out[valid] = best[valid]
            # "Failed to find a best fit... The cell will be removed" (CPUPhotomosaicGenerator.cpp:173-177)
            out[valid & (best >= self.m_engine.n_library)] = GridUtility.NULLOPT
            out_fits.append(out)
            out_scores.append(score)
        self.m_bestFits = out_fits
        self.m_bestScores = out_scores
        self.m_engine.set_cell_scale(1)
        for fn in listeners:
            fn(self.getMaxProgress())
        return True

    def _cells_of(self, step, state):
        """``GridUtility.cell_records`` of a step, kept between calls: the records depend on the cell group, the image
        size and WHICH grid positions hold a value -- not on the values -- so a second generateBestFits over the same
        grid (another colour difference, repeat setting or library) reuses them (312 500 records of the reference
        harness's cell size 8 are ~10 ms of numpy per call otherwise)."""
        import hashlib
        valid = state >= 0
        key = (self._cells_version, self._rows, self._cols, state.shape, self.m_cells.getDetail(),
               hashlib.blake2b(np.packbits(valid).tobytes(), digest_size=16).digest())
        hit = self._cell_cache.get(step)
        if hit is None or hit[0] != key:
            hit = (key, GridUtility.cell_records(self.m_cells, state, self._rows, self._cols, step))
            self._cell_cache[step] = hit
        return hit[1]

    def buildPhotomosaic(self, backgroundColour=(0, 0, 0, 0)):
        """PhotomosaicGeneratorBase::buildPhotomosaic (PhotomosaicGeneratorBase.cpp:110-207) on the
        device: u8 BGRA ``[rows][cols][4]``.  Needs the library as u8 BGR (what setLibrary got), at
        the NORMAL cell size of step 0.  Size steps: each step's cells are painted with the library
        halved once more, only where no earlier step's cell masks cover the pixel (:189-196)."""
        if self.m_lib is None or self.m_lib.dtype != np.uint8:
            raise ValueError("buildPhotomosaic needs the u8 BGR library given to setLibrary")
        if self.m_lib.shape[1] != self.m_cells.getCellSize(0):
            raise ValueError("buildPhotomosaic needs the library at the cell size %d" % self.m_cells.getCellSize(0))
        rows, cols = self.m_img.shape[0], self.m_img.shape[1]
        self.m_engine.set_library_bgr(self.m_lib)
        self._masks_key = None
        out = None
        for index, step in enumerate(self.m_bestFits):
            if index > 0:
                self.m_engine.shrink_library_bgr(2)
            shape = self.m_cells.getCell(index)
            self.m_engine.set_masks(shape.masks4())
            # every cell the grid names (with or without a value: the engine skips the latter)
            state = np.where(step >= 0, 0, GridUtility.NULLOPT)
            cells = GridUtility.cell_records(CellGroup(shape), state, rows, cols)   # NORMAL cell, full size
            out = self.m_engine.build_photomosaic(cells, step.shape[1], step.shape[0], step, rows, cols,
                                                  backgroundColour, step_index=index,
                                                  want_output=index == len(self.m_bestFits) - 1)
        return out

    def getBestFits(self):
        return [s.copy() for s in self.m_bestFits]

    def getBestScores(self):
        """bestVariant per cell (NaN where the grid has no value); not in the
        reference interface, used by the parity tests."""
        return [s.copy() for s in self.m_bestScores]

    def timing(self):
        return self.m_engine.timing()

    def timingLog(self):
        """The last generateBestFits' times under the ids the reference's TimingLogger records for its device path
        (CUDAPhotomosaicGenerator.cpp:45-358,376-506; Other/TimingLogger.h:209-326), in ms.  The engine's own phases
        are finer and differently cut (mm_timing); this is the coarse view a maintainer comparing logs wants:
        cudaMalloc = device allocations since the engine was made, Preprocess = the u8 -> f32 / Lab kernels,
        cudaMemcpy = cell records to the device + the cell gather (the reference uploads every cell, :176-229),
        DiffReduce = seeds + filter + its audit + fp64 re-score (the reference's difference and reduction kernels, :231-259),
        Flatten = candidate selection (:262-268), Repeats + FindLowest = the resolve (:271-292)."""
        t, st = self.m_engine.timing(), self.m_engine.stage_timing()
        return {"generateBestFits": t["total_ms"], "Preprocess": st["preprocess_ms"], "cudaMalloc": st["alloc_ms"] + st["pinned_ms"],
                "cudaMemcpy": t["stage_ms"], "DiffReduce": t["seed_ms"] + t["score_ms"] + t["audit_ms"] + t["rescore_ms"],
                "Flatten": t["select_ms"], "Repeats+FindLowest": t["resolve_ms"]}
