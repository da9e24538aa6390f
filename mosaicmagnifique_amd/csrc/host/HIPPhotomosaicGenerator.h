// HIPPhotomosaicGenerator.h -- C++ host side above the C ABI (include/mosaic_hip.h).
//
// Mirrors the reference's generator interface for the best-fit path: the setters,
// generateBestFits(), getBestFits(), getMaxProgress(), cancel() and the progress
// signal of PhotomosaicGeneratorBase (src/Photomosaic/PhotomosaicGeneratorBase.h:36-81),
// constructed like CUDAPhotomosaicGenerator(device) (CUDA/CUDAPhotomosaicGenerator.h:31),
// with the same types where they are plain C++ (GridUtility::MosaicBestFit is the
// reference's std::vector<std::vector<std::vector<std::optional<size_t>>>>,
// Grid/GridUtility.h:29-31).  cv::Mat becomes `Image` (f32, HWC, 3 channels, already
// colour-converted as preprocessMainImage / preprocessLibraryImages would) and
// CellShape / CellGroup keep only what the path reads.  No Qt, no OpenCV: those are
// not in this image; INTEGRATION.md shows the same class written against them.
//
// Host-side arithmetic here is integer geometry only (GridUtility.cpp:25-132,
// PhotomosaicGeneratorBase.cpp:299-326); every colour difference, sum and argmin
// happens behind the C ABI.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/mosaic_hip.h"

namespace ColourDifference {
enum class Type { RGB_EUCLIDEAN = 0, CIE76 = 1, CIEDE2000 = 2, MAX = 3 };   // ColourDifference.h:13-19
}

namespace ColourScheme {
enum class Type { NONE = 0, COMPLEMENTARY, TRIADIC, COMPOUND, TETRADIC, ANALAGOUS, MAX };   // ColourScheme.h:13-22
// Hue rotations (degrees) of the variants that follow the original image, ColourScheme.cpp:44-177.
std::vector<float> rotations(Type type);                                                  // throws std::invalid_argument on MAX
}

namespace GridUtility {
using CellBestFit = std::optional<size_t>;
using StepBestFit = std::vector<std::vector<CellBestFit>>;
using MosaicBestFit = std::vector<StepBestFit>;
constexpr int PAD_GRID = 2;                                                  // GridUtility.h:33
struct Rect { int x, y, width, height; };
struct Point { int x, y; };
struct FlipState { bool horizontal = false, vertical = false; };
}  // namespace GridUtility

// f32 HWC, 3 channels (already colour-converted), and u8 BGR HWC as cv::imread delivers it.
struct Image {
    int rows = 0, cols = 0;
    std::vector<float> data;
};
struct Image8 {
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
};

// Other/ImageUtility.{h,cpp} as far as the grid needs it.  OpenCV is not available here:
// INTER_AREA and BGR2GRAY restate the library's published algorithms (see
// mosaicmagnifique_amd/imageutil.py for the references); the tests hold this file, the
// Python host and the CPU checker to the same values.
namespace ImageUtility {
constexpr double MAX_ENTROPY = 8.0;                                           // ImageUtility.h:69
// cv::resize(..., INTER_AREA) for a shrink of a u8 image with cn interleaved channels.
std::vector<uint8_t> resizeArea(const std::vector<uint8_t>& src, int rows, int cols, int cn, int outRows, int outCols);
// calculateEntropy (ImageUtility.cpp:189-242): bgr [rows][cols][3]; mask (may be null) with row stride maskStride.
double calculateEntropy(const uint8_t* bgr, int rows, int cols, const uint8_t* mask, int maskStride);
}  // namespace ImageUtility

// The `.mil` container the reference feeds to setLibrary (ImageLibrary/ImageLibrary.cpp:117-236), read without Qt / OpenCV:
// big-endian QDataStream -- u32 magic 0xADBE2480, u32 version (4..6), u32 image size, u32 count, then per image the pixels
// (v6: a PNG in a QByteArray; v4 / v5: raw type / rows / cols + bytes, CustomQDataStream.h:22-87) and its name as a QString.
// Images come back in file order as u8 BGR [count][size][size][3] -- what setLibrary takes (the reference's shuffle of
// pre-v5 files on load, :207-217, is not reproduced).
struct ImageLibraryFile {
    int imageSize = 0;
    std::vector<std::string> names;
    std::vector<uint8_t> bgr;
    size_t count() const { return names.size(); }
    static ImageLibraryFile loadFromFile(const std::string& path);
};

// The part of CellShape (CellShape/CellShape.h:25-113) the path reads.
class CellShape {
public:
    CellShape() = default;
    CellShape(int size, std::vector<uint8_t> mask);     // thresholded > 127 like setCellMask
    const std::vector<uint8_t>& getCellMask(bool flippedHorizontal, bool flippedVertical) const;
    int getSize() const { return m_size; }
    bool empty() const { return m_size == 0; }
    // CellShape::resized (CellShape.cpp:281-312): mask resized + re-thresholded, spacings
    // floored (never < 1), offsets floored, flip flags copied.
    CellShape resized(int size) const;
    // CellShape::loadFromFile (CellShape.cpp:373-434) without Qt / OpenCV: big-endian QDataStream -- u32 magic
    // 0x87AECFB1, u32 version (7 or 8), QString name (UTF-16BE), the mask (v8: PNG in a QByteArray; v7: raw
    // type / rows / cols + bytes, CustomQDataStream.h:22-87), 6 x i32 spacings / offsets, 4 x bool flips.
    // The mask is kept as stored (not re-thresholded), as the reference does.  Throws std::runtime_error.
    static CellShape loadFromFile(const std::string& path);
    std::string name;
    int rowSpacing = 0, colSpacing = 0, alternateRowSpacing = 0, alternateColSpacing = 0;
    int alternateRowOffset = 0, alternateColOffset = 0;
    bool alternateColFlipHorizontal = false, alternateColFlipVertical = false;
    bool alternateRowFlipHorizontal = false, alternateRowFlipVertical = false;

private:
    int m_size = 0;
    std::vector<uint8_t> m_masks[4];      // index flipH + 2*flipV
};

// CellGroup (CellShape/CellGroup.h:26-64, CellGroup.cpp): normal + detail cell shape per size
// step.  The engine covers detail levels and steps whose cell sizes divide evenly.
class CellGroup {
public:
    void setCellShape(const CellShape& shape);
    void setDetail(int detailPercent = 100, bool reset = false);      // CellGroup.cpp:65-87
    void setSizeSteps(size_t steps, bool reset = false);              // CellGroup.cpp:96-128
    const CellShape& getCell(size_t sizeStep, bool detail = false) const { return (detail ? m_detailCells : m_cells).at(sizeStep); }
    int getCellSize(size_t sizeStep, bool detail = false) const { return getCell(sizeStep, detail).getSize(); }
    double getDetail() const { return m_detail; }
    size_t getSizeSteps() const { return m_sizeSteps; }

private:
    std::vector<CellShape> m_cells{1}, m_detailCells{1};
    double m_detail = 1.0;
    size_t m_sizeSteps = 0;
};

namespace GridUtility {
Point calculateGridSize(const CellShape& cellShape, int imageWidth, int imageHeight, int pad);
Rect getRectAt(const CellShape& cellShape, int x, int y);
FlipState getFlipStateAt(const CellShape& cellShape, int x, int y);
// The local bound of a cell at the detail size (PhotomosaicGeneratorBase.cpp:318-326).
Rect detailBound(const Rect& local, double detail, int size);
// The active bounds of a size step, kept as their UNION: all findCellState asks of them is whether a cell rect shares
// positive area with any bound (GridGenerator.cpp:124-140), and the reference's GridBounds::mergeBounds
// (Grid/GridBounds.cpp:36-107) never changes that union -- so nothing is merged: coordinate compression + a summed-area
// table of the covered boxes, four look-ups per query.
class BoundsUnion {
public:
    explicit BoundsUnion(const std::vector<Rect>& bounds);
    bool meets(const Rect& cellRect) const;
private:
    std::vector<int> m_xs, m_ys;          // distinct edge coordinates
    std::vector<int> m_area;              // [(ys - 1) + 1][(xs - 1) + 1] summed-area table of covered elementary boxes
};
// GridGenerator::getGridState (Grid/GridGenerator.cpp:29-193).  mainImage (u8 BGR, may be null)
// drives the entropy subdivision over size steps; without it every cell meeting the image is valid.
MosaicBestFit getGridState(const CellGroup& cells, const Image8* mainImage, int imageRows, int imageCols);
inline MosaicBestFit getGridState(const CellGroup& cells, int imageRows, int imageCols) { return getGridState(cells, nullptr, imageRows, imageCols); }
}  // namespace GridUtility

class HIPPhotomosaicGenerator {
public:
    explicit HIPPhotomosaicGenerator(int device = 0);
    // Several GPUs from ONE process (the reference's generator is single-device,
    // CUDA/CUDAPhotomosaicGenerator.h:31): the library's mm_group -- one engine per entry of `devices`,
    // library / masks / target replicated, the valid cells of a step cut into contiguous shards, ONE RCCL
    // all-gather of the candidate lists, the same verifying resolve on every device (mm_group_generate); a
    // device id listed more than once gathers through host memory (RCCL refuses it).  The grid equals
    // the single-device one.
    bool usesRccl() const { return m_group && mm_group_uses_rccl(m_group) != 0; }
    explicit HIPPhotomosaicGenerator(const std::vector<int>& devices);
    ~HIPPhotomosaicGenerator();
    size_t deviceCount() const { return m_engines.size(); }
    // mm_set_option on every engine (tuning / tests).
    bool setEngineOption(const char* name, int value);
    HIPPhotomosaicGenerator(const HIPPhotomosaicGenerator&) = delete;
    HIPPhotomosaicGenerator& operator=(const HIPPhotomosaicGenerator&) = delete;

    void setMainImage(const Image& img);                       // already preprocessed (f32 BGR / Lab)
    void setLibrary(const std::vector<Image>& lib);            // already preprocessed, at the detail cell size
    // u8 BGR as the reference's setters receive them: resized (library) and colour-converted on the device.
    void setMainImage(const Image8& img);
    void setLibrary(const std::vector<Image8>& lib);
    // The library as u8 BGR (what the reference's setLibrary receives before preprocessing);
    // only buildPhotomosaic needs it.
    void setLibraryBGR(const std::vector<std::vector<uint8_t>>& libBgr) { m_libBgr = libBgr; m_staleBgr = true; }
    void setColourDifference(ColourDifference::Type type);      // throws std::invalid_argument on MAX
    // PhotomosaicGeneratorBase::setColourScheme (PhotomosaicGeneratorBase.cpp:64-69).  getCellAt
    // builds the V variant cells as copies of ONE cv::Mat header (:310), so they share a buffer and
    // every cell the best-fit scan sees holds the LAST variant (:311-312): the reference's result
    // with a scheme equals its result with the last variant as the main image.  That variant is
    // what gets staged (u8 main image required); mainImageVariants() returns the whole list.
    void setColourScheme(ColourScheme::Type type = ColourScheme::Type::NONE);
    std::vector<Image8> mainImageVariants();                      // m_colourSchemeFunc(m_img), :231-232; empty on error
    void setCellGroup(const CellGroup& cellGroup);
    const CellGroup& getCellGroup() const { return m_cells; }
    void setGridState(const GridUtility::MosaicBestFit& gridState);
    void setRepeat(int repeatRange = 0, int repeatAddition = 0);

    // Returns true if successful; false on cancel or device error (lastError()).
    bool generateBestFits();
    GridUtility::MosaicBestFit getBestFits() const { return m_bestFits; }
    // PhotomosaicGeneratorBase::buildPhotomosaic (PhotomosaicGeneratorBase.cpp:110-207) on the
    // device: BGRA u8 [rows][cols][4]; empty on error (lastError()).
    std::vector<uint8_t> buildPhotomosaic(const uint8_t backgroundColour[4]);
    int getMaxProgress() const;
    void cancel() { m_wasCanceled = 1; }
    std::function<void(int)> progress;                            // the progress(int) signal
    const std::string& lastError() const { return m_lastError; }
    mm_timing timing() const;
    // The last generateBestFits' times (ms) under the ids the reference's TimingLogger records for its device path
    // (CUDAPhotomosaicGenerator.cpp:45-358,376-506): generateBestFits, Preprocess, cudaMalloc, cudaMemcpy, DiffReduce,
    // Flatten, Repeats+FindLowest -- the coarse view of mm_timing / mm_stage_timing a maintainer comparing logs wants.
    std::vector<std::pair<std::string, double>> timingLog() const;

private:
    mm_engine* m_engine = nullptr;          // m_engines[0]: the one that resolves and paints
    std::vector<mm_engine*> m_engines;
    mm_group* m_group = nullptr;            // several devices: owns m_engines
    bool generateStepSharded(const std::vector<mm_cell>& cells, int gridW, int gridH, std::vector<int64_t>& bestFit);
    Image m_img;
    std::vector<Image> m_lib;
    Image8 m_img8;
    std::vector<Image8> m_lib8;
    int m_rows = 0, m_cols = 0;
    size_t m_libCount = 0;
    bool stageInputs();
    bool stageInputsOn();           // ... onto m_engine
    std::vector<std::vector<uint8_t>> m_libBgr;
    bool m_staleInputs = true, m_staleBgr = true;
    std::vector<mm_cell> cellRecords(size_t step, bool painting) const;
    ColourDifference::Type m_colourDiffType = ColourDifference::Type::RGB_EUCLIDEAN;
    ColourScheme::Type m_colourSchemeType = ColourScheme::Type::NONE;
    CellGroup m_cells;
    GridUtility::MosaicBestFit m_bestFits;
    int m_repeatRange = 0, m_repeatAddition = 0;
    volatile int m_wasCanceled = 0;
    std::string m_lastError;
};
