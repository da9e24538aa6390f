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
#include <vector>

#include "../../../include/mosaic_hip.h"

namespace ColourDifference {
enum class Type { RGB_EUCLIDEAN = 0, CIE76 = 1, CIEDE2000 = 2, MAX = 3 };   // ColourDifference.h:13-19
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

// f32 HWC, 3 channels.
struct Image {
    int rows = 0, cols = 0;
    std::vector<float> data;
};

// The part of CellShape (CellShape/CellShape.h:25-113) the path reads.
class CellShape {
public:
    CellShape() = default;
    CellShape(int size, std::vector<uint8_t> mask);     // thresholded > 127 like setCellMask
    const std::vector<uint8_t>& getCellMask(bool flippedHorizontal, bool flippedVertical) const;
    int getSize() const { return m_size; }
    int rowSpacing = 0, colSpacing = 0, alternateRowSpacing = 0, alternateColSpacing = 0;
    int alternateRowOffset = 0, alternateColOffset = 0;
    bool alternateColFlipHorizontal = false, alternateColFlipVertical = false;
    bool alternateRowFlipHorizontal = false, alternateRowFlipVertical = false;

private:
    int m_size = 0;
    std::vector<uint8_t> m_masks[4];      // index flipH + 2*flipV
};

// CellGroup (CellShape/CellGroup.h:26-64) at detail 100 %, size steps 0.
class CellGroup {
public:
    void setCellShape(const CellShape& shape) { m_cell = shape; }
    const CellShape& getCell(size_t sizeStep, bool detail = false) const { (void)sizeStep; (void)detail; return m_cell; }
    double getDetail() const { return 1.0; }
    size_t getSizeSteps() const { return 0; }

private:
    CellShape m_cell;
};

namespace GridUtility {
Point calculateGridSize(const CellShape& cellShape, int imageWidth, int imageHeight, int pad);
Rect getRectAt(const CellShape& cellShape, int x, int y);
FlipState getFlipStateAt(const CellShape& cellShape, int x, int y);
// GridGenerator::getGridState for size steps 0 (Grid/GridGenerator.cpp:29-110).
MosaicBestFit getGridState(const CellGroup& cells, int imageRows, int imageCols);
}  // namespace GridUtility

class HIPPhotomosaicGenerator {
public:
    explicit HIPPhotomosaicGenerator(int device = 0);
    ~HIPPhotomosaicGenerator();
    HIPPhotomosaicGenerator(const HIPPhotomosaicGenerator&) = delete;
    HIPPhotomosaicGenerator& operator=(const HIPPhotomosaicGenerator&) = delete;

    void setMainImage(const Image& img);
    void setLibrary(const std::vector<Image>& lib);
    // The library as u8 BGR (what the reference's setLibrary receives before preprocessing);
    // only buildPhotomosaic needs it.
    void setLibraryBGR(const std::vector<std::vector<uint8_t>>& libBgr) { m_libBgr = libBgr; m_staleBgr = true; }
    void setColourDifference(ColourDifference::Type type);      // throws std::invalid_argument on MAX
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

private:
    mm_engine* m_engine = nullptr;
    Image m_img;
    std::vector<Image> m_lib;
    std::vector<std::vector<uint8_t>> m_libBgr;
    bool m_staleInputs = true, m_staleBgr = true;
    std::vector<mm_cell> cellRecords(bool onlyWithValue) const;
    ColourDifference::Type m_colourDiffType = ColourDifference::Type::RGB_EUCLIDEAN;
    CellGroup m_cells;
    GridUtility::MosaicBestFit m_bestFits;
    int m_repeatRange = 0, m_repeatAddition = 0;
    volatile int m_wasCanceled = 0;
    std::string m_lastError;
};
